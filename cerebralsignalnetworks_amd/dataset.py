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


class EEGDataset(Dataset):
    def __init__(self, eeg_signals_path=None, eeg_splits_path=None, subset='train', subject=1, exclude_subjects=(),
                 filter_channels=(), time_low=20, time_high=480, model_type="cnn",
                 imagesRoot="./data/images/imageNet_images", apply_norm_with_stds_and_means=False,
                 apply_channel_wise_norm=False, preprocessin_fn=None, inference_mode=True, onehotencode_label=False,
                 synthetic=0, synthetic_channels=128, synthetic_samples=500, n_classes=40, feature_dim=384, seed=43,
                 device=None, compat_label_bug=False, **_ignored):
        assert subset in ('train', 'val', 'test')
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
            self.mean, self.std = raw.mean(), raw.std()
        else:
            loaded = torch.load(eeg_signals_path, weights_only=True)       # ConvertToPth.py:170-201 format
            items = loaded["dataset"]
            if eeg_splits_path:                                             # EEGDataset.py:52-69
                splits = torch.load(eeg_splits_path, weights_only=True)
                keep = [i for i in splits["splits"][0][subset] if i < len(items)]
                items = [items[i] for i in keep]
            if subject and any("subject" in it for it in items) and eeg_splits_path:
                items = [it for it in items if it.get("subject", subject) == subject
                         and it.get("subject") not in exclude_subjects]
            raw = torch.stack([it["eeg"].float() for it in items])         # [N,C,T_raw]
            self.labels = torch.tensor([int(it["label"]) for it in items])
            self.subjects = torch.tensor([int(it.get("subject", 0)) for it in items])
            self.class_labels = loaded["labels"]
            image_names = loaded["images"]
            self.images = [image_names[int(it["image"])] for it in items]
            self._read_label_file(set(n.split("_")[0] for n in image_names))
            if apply_norm_with_stds_and_means and "means" in loaded and "stddevs" in loaded:
                raw = (raw - torch.as_tensor(loaded["means"]).reshape(1, -1, 1).float()) \
                    / torch.as_tensor(loaded["stddevs"]).reshape(1, -1, 1).float()      # EEGDataset.py:104-105
            self.mean = torch.stack([r.mean() for r in raw]).mean()        # PerilsEEGDataset.py:90-103
            self.std = torch.stack([r.std() for r in raw]).mean()
            self.features_all = None
        if len(filter_channels) > 0:
            raw = raw[:, list(filter_channels), :]
        self.eeg_all = raw[:, :, time_low:time_high].contiguous().to(self.device)      # [N,C,T]
        self.labels_dev = self.labels.to(self.device)
        self.size = self.eeg_all.shape[0]

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
        if self.apply_norm_with_stds_and_means and not hasattr(self, "class_labels"):
            eeg = (eeg - self.mean) / self.std
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
    def transformEEGDataToChannelWiseNorm(self, compat_stale_index=False):
        """Class-wise, channel-wise normalisation of the resident segments (PerilsEEGDataset.py:464-507): per class
        and channel, (x - mean of the per-segment means) / (mean of the per-segment stds, ddof 0), statistics over
        the [time_low:time_high] window.  One pass on the device instead of the reference's N x C Python loop.

        compat_stale_index=True reproduces what the reference's code actually leaves behind (it stores every
        result into the LAST record -- ``self.subsetData[i]`` with a stale ``i``, :507 -- and indexes the
        channel-first record as ``eeg[:, ch]``, :503-506): only record N-1 changes."""
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
        seen = []
        for k in self.labels.tolist():                                   # classes in order of first appearance
            if k not in seen:
                seen.append(k)
        last_class = seen[-1]
        last_idx = int((self.labels == last_class).nonzero()[-1])
        rec = x[last_idx].clone()
        lo = max(0, self.time_low)
        chs = torch.arange(C, device=x.device)
        cols = chs - lo
        ok = (cols >= 0) & (cols < T)
        rec[:, cols[ok]] = (rec[:, cols[ok]] - cls_mean[last_class][chs[ok]][None, :]) / cls_std[last_class][chs[ok]][None, :]
        self.eeg_all[N - 1] = rec

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
