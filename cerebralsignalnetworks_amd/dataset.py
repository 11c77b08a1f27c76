"""EEG dataset with the call-site contract of the reference's ``EEGDataset``
(/root/reference/utils/PerilsEEGDataset.py:12-30, :541-623; utils/EEGDataset.py), kept
resident on the GPU.

Reference behaviour kept: stored EEG is ``[C, T_raw]`` per item (ConvertToPth.py:170-201);
``__getitem__(i)`` returns ``(eeg[T,C] float32, label, image, i, image_features)`` with
``eeg = stored.float().t()[time_low:time_high, :]`` (:549,:567); ``label`` is the dict
``{"ClassId","ClassName","imagenetClassId"}`` in ``inference_mode`` (:597-606);
``getLabelbyIndex``, ``class_id_to_str``, ``class_str_to_id``, ``extract_features``,
``transformEEGDataLSTMByList`` (incl. its batch-local label lookup, :336-338, behind
``compat_label_bug``).

What differs, on purpose: all segments live in ONE device tensor
``eeg_all[N,C,T]`` (already time-sliced), teacher embeddings in ``features_all[N,D]``; the hot
loop indexes them on the device instead of decoding a JPEG per item per step
(PerilsEEGDataset.py:608-611).  Images are only touched by ``extract_features``.
"""
import os

import numpy as np
import torch
from torch.utils.data import Dataset


def synthetic_eeg(n, channels=128, samples=500, fs=1000.0, freq=40.0, amp=0.5, seed=43):
    """N(0,1) + 0.5 sin(2 pi 40 t) -- utils/GenerateRandomEEGNoise.py:4-19 / PerilsEEGDataset.py:140-148."""
    g = torch.Generator().manual_seed(seed)
    t = torch.arange(samples, dtype=torch.float32) / fs
    return torch.randn(n, channels, samples, generator=g) + amp * torch.sin(2 * np.pi * freq * t)


def clustered_eeg(n, n_classes=40, C=128, T=500, seed=101, snr=0.2):
    """Seeded class-clustered raw EEG [n, C, T] f32 + labels: class template (low-pass noise) * snr + N(0,1).
    The input set of the retrieval acceptance check (bench.py, tests/test_gpu_fullsize.py; the CPU reference's
    neighbour lists for it are the fixture tests/golden/ref_retrieval_cfg2.npz)."""
    rng = np.random.default_rng(seed)
    labels = rng.integers(0, n_classes, n)
    tpl = rng.standard_normal((n_classes, C, T))
    k = np.hanning(15)
    k /= k.sum()
    tpl = np.apply_along_axis(lambda v: np.convolve(v, k, mode="same"), -1, tpl)
    tpl /= tpl.std(axis=-1, keepdims=True)
    x = snr * tpl[labels] + rng.standard_normal((n, C, T))
    return x.astype(np.float32), labels


class EEGDataset(Dataset):
    def __init__(self, eeg_signals_path=None, eeg_splits_path=None, subset='train', subject=1, exclude_subjects=(),
                 filter_channels=(), time_low=20, time_high=480, model_type="cnn",
                 imagesRoot="./data/images/imageNet_images", apply_norm_with_stds_and_means=False,
                 apply_channel_wise_norm=False, preprocessin_fn=None, inference_mode=True, onehotencode_label=False,
                 synthetic=0, synthetic_channels=128, synthetic_samples=500, n_classes=40, feature_dim=384, seed=43,
                 device=None, compat_label_bug=False, compat_stale_index=False, flavour="perils", **_ignored):
        assert subset in ('train', 'val', 'test')
        assert flavour in ("perils", "spampinato")       # utils/PerilsEEGDataset.py vs utils/EEGDataset.py
        self.time_low, self.time_high = time_low, time_high
        self.imagesRoot, self.preprocessin_fn = imagesRoot, preprocessin_fn
        self.inference_mode, self.onehotencode_label = inference_mode, onehotencode_label
        self.apply_norm_with_stds_and_means = apply_norm_with_stds_and_means
        self.compat_label_bug = compat_label_bug
        self.device = device or (torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available()
                                 else torch.device("cpu"))
        self.image_features_extracted = False
        self.class_id_to_str, self.class_str_to_id, self.class_labels_names = {}, {}, {}

        if synthetic:
            raw = synthetic_eeg(synthetic, synthetic_channels, synthetic_samples, seed=seed)      # [N,C,T_raw]
            self.labels = torch.randint(0, n_classes, (synthetic,), generator=torch.Generator().manual_seed(seed + 2))
            self.images = [f"n{int(c):08d}_{i}" for i, c in enumerate(self.labels)]
            for c in range(n_classes):
                wnid = f"n{c:08d}"
                self.class_labels_names[wnid] = {"ClassId": c, "ClassName": f"class_{c}", "imagenetClassId": str(c)}
                self.class_id_to_str[c] = f"class_{c}"
                self.class_str_to_id[f"class_{c}"] = c
            self.subjects = torch.ones(synthetic, dtype=torch.long)
            feats = torch.randn(synthetic, feature_dim, generator=torch.Generator().manual_seed(seed + 1))
            self.features_all = feats.to(self.device)
            self.image_features_extracted = True
            self.mean = torch.stack([r.mean() for r in raw]).mean()
            self.std = torch.stack([r.std() for r in raw]).mean()
        else:
            loaded = torch.load(eeg_signals_path, weights_only=True)       # ConvertToPth.py:170-201 format
            items = loaded["dataset"]
            if flavour == "spampinato":
                # utils/EEGDataset.py:52-53,99-128: split file -> subset indexes; subject != 0 keeps that subject,
                # subject == 0 keeps every subject not in exclude_subjects; per-channel (eeg - means) / stddevs
                splits = torch.load(eeg_splits_path, weights_only=True)
                keep = [int(i) for i in splits["splits"][0][subset]]
                if subject != 0:
                    keep = [i for i in keep if int(items[i]["subject"]) == subject]
                else:
                    keep = [i for i in keep if int(items[i]["subject"]) not in exclude_subjects]
                items = [items[i] for i in keep]
            raw = torch.stack([it["eeg"].float() for it in items])         # [N,C,T_raw]
            self.labels = torch.tensor([int(it["label"]) for it in items])
            self.subjects = torch.tensor([int(it.get("subject", 0)) for it in items])
            self.class_labels = loaded["labels"]
            image_names = loaded["images"]
            self.images = [image_names[int(it["image"])] for it in items]
            self._read_label_file(set(n.split("_")[0] for n in image_names))
            if flavour == "spampinato":
                if apply_norm_with_stds_and_means:                          # EEGDataset.py:104-105 (at load)
                    means, stds = loaded["means"], loaded["stddevs"]
                    means = means[0] if isinstance(means, (list, tuple)) else means
                    stds = stds[0] if isinstance(stds, (list, tuple)) else stds
                    raw = (raw - torch.as_tensor(means).reshape(1, -1, 1).float()) \
                        / torch.as_tensor(stds).reshape(1, -1, 1).float()
            else:
                # PerilsEEGDataset.py:90-103: scalars = mean over records of the record mean / std (torch std, ddof 1),
                # taken in the stored dtype; applied per item in __getitem__ (:572-573)
                self.mean = torch.stack([it["eeg"].mean() for it in items]).mean().float()
                self.std = torch.stack([it["eeg"].std() for it in items]).mean().float()
            self.features_all = None
        self.flavour = flavour
        self.filter_channels = list(filter_channels)
        self.apply_channel_wise_norm = apply_channel_wise_norm
        if len(self.filter_channels) > 0:
            raw = raw[:, self.filter_channels, :]
        self.eeg_all = raw[:, :, time_low:time_high].contiguous().to(self.device)      # [N,C,T]
        self.labels_dev = self.labels.to(self.device)
        self.size = self.eeg_all.shape[0]
        if apply_channel_wise_norm and flavour == "perils" and not synthetic:
            self.transformEEGDataToChannelWiseNorm(compat_stale_index=compat_stale_index,      # :132-134
                                                   stored_float32=items[0]["eeg"].dtype == torch.float32)

    def _read_label_file(self, wanted):
        path = f"{self.imagesRoot}/labels.txt"                              # PerilsEEGDataset.py:76-88
        if not os.path.exists(path):
            for idx, wnid in enumerate(self.class_labels):
                self.class_labels_names[wnid] = {"ClassId": idx, "ClassName": wnid, "imagenetClassId": str(idx)}
                self.class_id_to_str[idx], self.class_str_to_id[wnid] = wnid, idx
            return
        with open(path) as f:
            for line in f:
                parts = line.strip().split(" ")
                if parts[0] in wanted:
                    idx = self.class_labels.index(parts[0])
                    self.class_labels_names[parts[0]] = {"ClassId": int(idx), "ClassName": parts[-1],
                                                         "imagenetClassId": parts[1]}
                    self.class_id_to_str[int(idx)] = parts[-1]
                    self.class_str_to_id[parts[-1]] = int(idx)

    def __len__(self):
        return self.size

    def getLabelbyIndex(self, index):
        wnid = self.images[int(index)].split("_")[0]
        return self.class_labels_names[wnid]

    def __getitem__(self, i):
        i = int(i)
        eeg = self.eeg_all[i].t()                                          # [T,C]
        if len(self.filter_channels) > 0:
            # :552-565: the selected columns, each optionally z-scored (numpy std, ddof 0), and -- as the reference's
            # final ``.t()`` leaves it -- channel-first [len(filter_channels), T]
            # (eeg_all already holds only the selected channels)
            if self.apply_channel_wise_norm:
                eeg = (eeg - eeg.mean(dim=0, keepdim=True)) / eeg.std(dim=0, unbiased=False, keepdim=True)
            eeg = eeg.t()
        if self.apply_norm_with_stds_and_means and self.flavour == "perils":
            eeg = (eeg - self.mean) / self.std                             # :572-573, scalars
        label = self.getLabelbyIndex(i)
        if not self.inference_mode:
            label = label["ClassId"]
            if self.onehotencode_label:
                onehot = torch.zeros(len(self.class_labels_names), dtype=torch.long)
                onehot[label] = 1
                label = onehot
        image = self._load_image(i)
        feats = self.features_all[i] if self.image_features_extracted else []
        return eeg, label, image, i, feats

    def _load_image(self, i):
        wnid = self.images[i].split("_")[0]
        path = f"{self.imagesRoot}/{wnid}/{self.images[i]}.JPEG"
        if self.preprocessin_fn is not None and os.path.exists(path):
            from PIL import Image
            return self.preprocessin_fn(Image.open(path).convert('RGB'))
        return torch.zeros(1)

    # ---- frozen-teacher precompute (PerilsEEGDataset.py:168-226) -----------------------------
    def set_features(self, features):
        self.features_all = torch.as_tensor(np.asarray(features), dtype=torch.float32).to(self.device)
        assert self.features_all.shape[0] == self.size
        self.image_features_extracted = True

    @torch.no_grad()
    def extract_features(self, model, data_loader=None, use_cuda=True, multiscale=False, replace_eeg=False,
                         batch_size=64):
        """Runs the frozen teacher once over all images; one all_gather after the local loop."""
        import torch.distributed as dist
        rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)
        idx = list(range(rank, self.size, world))
        outs = []
        for s in range(0, len(idx), batch_size):
            imgs = torch.stack([self._load_image(i) for i in idx[s:s + batch_size]]).to(self.device)
            outs.append(model(imgs).float())
        local = torch.cat(outs) if outs else torch.zeros(0, 1, device=self.device)
        if world > 1:
            pad = (self.size + world - 1) // world
            buf = torch.zeros(pad, local.shape[1], device=self.device)
            buf[: local.shape[0]] = local
            gathered = torch.empty(world * pad, local.shape[1], device=self.device)
            dist.all_gather_into_tensor(gathered, buf)
            feats = torch.empty(self.size, local.shape[1], device=self.device)
            for r in range(world):
                n_r = len(range(r, self.size, world))
                feats[r::world] = gathered[r * pad: r * pad + n_r]
        else:
            feats = local
        self.features_all = feats
        self.image_features_extracted = True

    @torch.no_grad()
    def transformEEGDataToChannelWiseNorm(self, compat_stale_index=False, stored_float32=False):
        """Class-wise, channel-wise normalisation of the resident segments (PerilsEEGDataset.py:464-507): per class
        and channel, (x - mean of the per-segment means) / (mean of the per-segment stds, ddof 0), statistics over
        the [time_low:time_high] window.  One pass on the device instead of the reference's N x C Python loop.

        compat_stale_index=True reproduces what the reference's code actually leaves behind: it indexes the
        channel-first record as ``eeg[:, ch]`` (:503-506: channel ch's statistics land on raw time sample ch), stores
        every result into entry N-1 (``self.subsetData[i]`` with a stale ``i``, :507) and, depending on the stored
        dtype, works on a copy (float64 records, what ConvertToPth.py writes) or in place (``stored_float32``:
        ``.float().cpu().numpy()`` aliases a float32 record).  The host walks the visiting order once to find which
        class statistics end up applied to which record; the arithmetic runs on the device."""
        x = self.eeg_all                                                # [N, C, T]
        N, C, T = x.shape
        labels = self.labels_dev.long()
        seg_mean = x.mean(dim=2)                                        # [N, C]
        seg_std = x.std(dim=2, unbiased=False)
        K = int(labels.max().item()) + 1
        count = torch.zeros(K, device=x.device).index_add_(0, labels, torch.ones(N, device=x.device)).clamp_(min=1)
        cls_mean = torch.zeros(K, C, device=x.device).index_add_(0, labels, seg_mean) / count[:, None]
        cls_std = torch.zeros(K, C, device=x.device).index_add_(0, labels, seg_std) / count[:, None]
        if not compat_stale_index:
            self.eeg_all = ((x - cls_mean[labels][:, :, None]) / cls_std[labels][:, :, None]).contiguous()
            return
        # which (source record, class statistics in order) does every entry end up holding?
        lab = self.labels.tolist()
        order = list(dict.fromkeys(lab))
        applied = [[] for _ in range(N)]         # per physical array: classes applied, in order
        source = list(range(N))                  # per physical array: the original record it started from
        aliased = [bool(stored_float32)] * N
        slot = list(range(N))
        for k in order:
            for i in (j for j in range(N) if lab[j] == k):
                phys = slot[i]
                if not aliased[phys]:
                    applied.append(list(applied[phys]))
                    source.append(source[phys])
                    aliased.append(True)
                    phys = len(applied) - 1
                applied[phys].append(k)
                slot[N - 1] = phys
        chs = torch.arange(C, device=x.device)
        cols = chs - max(0, self.time_low)
        ok = (cols >= 0) & (cols < T)
        cols, chs = cols[ok], chs[ok]
        out = x.clone()
        simple = [i for i in range(N) if slot[i] == i and applied[i] == [lab[i]]]
        if simple:                                # the common case of the in-place form: own class, once
            si = torch.tensor(simple, device=x.device)
            m, sd = cls_mean[labels[si]][:, chs], cls_std[labels[si]][:, chs]
            out[si[:, None], :, cols[None, :]] = ((x[si][:, :, cols] - m[:, None, :]) / sd[:, None, :]).permute(0, 2, 1)
        for i in range(N):
            if i in simple or (slot[i] == i and not applied[i]):
                continue
            rec = x[source[slot[i]]].clone()
            for k in applied[slot[i]]:
                rec[:, cols] = (rec[:, cols] - cls_mean[k][chs][None, :]) / cls_std[k][chs][None, :]
            out[i] = rec
        self.eeg_all = out

    @torch.no_grad()
    def transformEEGDataLSTMByList(self, model, data_loader):
        """Embeds a loader's EEG with ``model``; returns (list of np rows, list of label dicts).
        ``compat_label_bug`` reproduces the reference's batch-local label lookup (:336-338)."""
        image_features, image_labels = [], []
        for EEG, labels, image, index, img_feat in data_loader:
            feats = model(EEG.to(self.device))
            feats = feats[0] if isinstance(feats, tuple) else feats
            rows = feats.float().cpu().numpy()
            for j, feat in enumerate(rows):
                image_features.append(feat)
                image_labels.append(self.getLabelbyIndex(j if self.compat_label_bug else int(index[j])))
        return image_features, image_labels
