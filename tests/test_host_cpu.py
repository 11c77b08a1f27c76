"""CPU: host-side logic -- CLI surface, dataset tuple contract, DistributedSampler-style sharding,
flat-gradient all-reduce over a 2-process gloo group (the N>1 path of the trainer)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cli_flags_match_reference_surface():
    import LstmDistillFromDinoV2Train as train
    # flags of /root/reference/LstmDistillFromDinoV2Train.py:151-225
    ref = {"learning_rate": 0.001, "num_epochs": 100, "batch_size": 16, "log_dir": './logs/DinoV2LstmDistillv2sdsad/',
           "gallery_subject": 1, "query_subject": 1, "images_root": "./data/images/imageNet_images",
           "eeg_dataset_split": "./data/eeg/block_splits_by_image_all.pth", "mode": "train",
           "custom_model_weights": "", "search_gallery": "train", "query_gallery": "test", "topK": 5,
           "gallery_tranformation_type": "eeg2eeg", "query_tranformation_type": "eeg2eeg", "seed": 43,
           "num_workers": 4, "dist_url": "env://", "local_rank": 0}
    flags, unknown = train.build_parser().parse_known_args(["--not_a_flag", "3"])      # parse_known_args, :231
    assert unknown == ["--not_a_flag", "3"]
    for k, v in ref.items():
        assert getattr(flags, k) == v, k
    assert "eeg_dataset" in vars(flags) and "hyperprams" in vars(flags)
    import ast
    assert ast.literal_eval(flags.hyperprams)["temperature"] == 2


def test_dataset_tuple_contract_and_label_bug_switch():
    from cerebralsignalnetworks_amd.dataset import EEGDataset
    ds = EEGDataset(synthetic=12, time_low=20, time_high=480, device=torch.device("cpu"), feature_dim=8)
    eeg, label, image, i, feats = ds[3]
    assert eeg.shape == (460, 128) and eeg.dtype == torch.float32          # [T,C], window 20:480
    assert set(label) == {"ClassId", "ClassName", "imagenetClassId"} and i == 3 and feats.shape == (8,)
    assert ds.class_str_to_id[ds.class_id_to_str[label["ClassId"]]] == label["ClassId"]
    assert ds.getLabelbyIndex(3) == label and len(ds) == 12
    np.testing.assert_array_equal(eeg.numpy(), ds.eeg_all[3].t().numpy())

    class Identity(torch.nn.Module):
        def forward(self, x):
            return x.mean(dim=1)
    loader = [([ds[j][0] for j in (5, 7)], None, None, torch.tensor([5, 7]), None)]
    loader = [(torch.stack(loader[0][0]), None, None, loader[0][3], None)]
    f, lab = ds.transformEEGDataLSTMByList(Identity(), loader)
    assert [l["ClassId"] for l in lab] == [ds.getLabelbyIndex(5)["ClassId"], ds.getLabelbyIndex(7)["ClassId"]]
    ds.compat_label_bug = True                                            # reference quirk: batch-local index
    f, lab = ds.transformEEGDataLSTMByList(Identity(), loader)
    assert [l["ClassId"] for l in lab] == [ds.getLabelbyIndex(0)["ClassId"], ds.getLabelbyIndex(1)["ClassId"]]


def test_on_disk_format_round_trip(tmp_path):
    """ConvertToPth.py:170-201 layout: {"dataset":[{eeg[C,T],image,label,subject}], "labels", "images", ...}."""
    from cerebralsignalnetworks_amd.dataset import EEGDataset
    g = torch.Generator().manual_seed(0)
    items = [{"eeg": torch.randn(6, 50, generator=g), "image": k % 3, "label": k % 2, "subject": 1} for k in range(5)]
    blob = {"dataset": items, "labels": ["n01", "n02"], "images": ["n01_a", "n02_b", "n01_c"],
            "means": torch.zeros(6), "stddevs": torch.ones(6)}
    path = tmp_path / "eeg.pth"
    torch.save(blob, path)
    ds = EEGDataset(eeg_signals_path=str(path), imagesRoot=str(tmp_path), time_low=5, time_high=45,
                    device=torch.device("cpu"))
    assert ds.eeg_all.shape == (5, 6, 40)
    eeg, label, _, _, _ = ds[4]
    np.testing.assert_array_equal(eeg.numpy(), items[4]["eeg"].t()[5:45].numpy())
    assert label["ClassId"] == blob["labels"].index(blob["images"][items[4]["image"]].split("_")[0])


def test_shard_indices_follow_distributed_sampler():
    from torch.utils.data.distributed import DistributedSampler
    from cerebralsignalnetworks_amd.trainer import shard_indices
    n, world = 103, 4
    seen = []
    for r in range(world):
        mine = shard_indices(n, epoch=3, seed=43, rank=r, world=world).tolist()
        ref = list(DistributedSampler(range(n), num_replicas=world, rank=r, shuffle=True, seed=43))
        # DistributedSampler needs set_epoch; emulate
        s = DistributedSampler(range(n), num_replicas=world, rank=r, shuffle=True, seed=43)
        s.set_epoch(3)
        assert mine == list(s)
        seen += mine
        assert len(mine) == len(ref)
    assert set(seen) == set(range(n))
    assert shard_indices(10, 0, 0, 1, 2, shuffle=False).tolist() == [1, 3, 5, 7, 9]


def _dp_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from cerebralsignalnetworks_amd.trainer import FlatGrads, dist_info
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
    for p in model.parameters():
        dist.broadcast(p.data, src=0)
    fg = FlatGrads(model.parameters())
    x = torch.arange(24, dtype=torch.float32).reshape(4, 6) / 10 + rank        # each rank: its own shard
    fg.zero()
    model(x).pow(2).mean().backward()
    fg.all_reduce_mean()
    assert dist_info() == (rank, world)
    out[rank] = fg.flat.clone().numpy()
    dist.barrier()
    dist.destroy_process_group()


def test_flat_gradient_allreduce_two_process_gloo():
    world, port = 2, 29611
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_dp_worker, args=(world, port, out), nprocs=world, join=True)
    # single-process reference: mean of the per-shard gradients == gradient of the mean loss over both shards
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
    grads = []
    for r in range(world):
        model.zero_grad()
        x = torch.arange(24, dtype=torch.float32).reshape(4, 6) / 10 + r
        model(x).pow(2).mean().backward()
        grads.append(torch.cat([p.grad.reshape(-1) for p in model.parameters()]).numpy().copy())
    want = np.mean(grads, axis=0)
    for r in range(world):
        np.testing.assert_allclose(out[r], want, rtol=1e-6, atol=1e-7)
    np.testing.assert_array_equal(out[0], out[1])


def _overlap_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from cerebralsignalnetworks_amd.trainer import FlatGrads
    res = {}
    for mode in ("blocking", "overlapped", "overlapped_head_late"):
        torch.manual_seed(0)
        model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 4), torch.nn.Tanh(), torch.nn.Linear(4, 3))
        fg = FlatGrads(model.parameters())
        n0 = sum(p.numel() for p in model[0].parameters())
        n1 = n0 + sum(p.numel() for p in model[2].parameters())
        if mode != "blocking":
            fg.segments = [(0, n0), (n0, n1), (n1, fg.flat.numel())]     # "layer 0", "layer 1", "head"
        x = torch.arange(24, dtype=torch.float32).reshape(4, 6) / 10 + rank
        fg.zero()
        model(x).pow(2).mean().backward()
        if mode == "overlapped":
            fg.segment_ready(1, also=(2,))          # top layer + head in one message while "layer 0" is still computing
        elif mode == "overlapped_head_late":
            fg.segment_ready(1)                     # the head's gradients were not final yet: they go with the rest
        fg.all_reduce_mean()
        res[mode] = fg.flat.clone().numpy()
        assert fg._works == [] and (fg._pending is None or not any(fg._pending))
    out[rank] = res
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_overlapped_allreduce_is_bit_identical_to_blocking_two_process_gloo():
    """FlatGrads with readiness-ordered segments (the form the trainer's gradient-ready hook drives) against the single
    blocking collective: same bits on both ranks, whichever way the head's segment is grouped."""
    world, port = 2, 29651
    out = mp.Manager().dict()
    mp.spawn(_overlap_worker, args=(world, port, out), nprocs=world, join=True)
    for r in range(world):
        for mode in ("overlapped", "overlapped_head_late"):
            np.testing.assert_array_equal(out[r][mode], out[r]["blocking"])
    np.testing.assert_array_equal(out[0]["blocking"], out[1]["blocking"])
    assert np.abs(out[0]["blocking"]).sum() > 0


def test_classwise_channel_norm_matches_oracle_both_modes():
    """PerilsEEGDataset.transformEEGDataToChannelWiseNorm (:464-507): the intended normalisation and, behind a
    switch, the state the reference's stale-index / transposed-index code actually leaves."""
    import torch
    from cerebralsignalnetworks_amd.dataset import EEGDataset
    from oracle import eeg_filter
    for compat in (False, True):
        ds = EEGDataset(synthetic=37, synthetic_channels=24, synthetic_samples=90, n_classes=5, time_low=10,
                        time_high=80, seed=3, device=torch.device("cpu"))
        before = ds.eeg_all.clone().numpy()
        want = eeg_filter.classwise_channel_norm(before, ds.labels.numpy(), time_low=10, compat_stale_index=compat)
        ds.transformEEGDataToChannelWiseNorm(compat_stale_index=compat)
        got = ds.eeg_all.numpy()
        np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-5)
        if compat:                                   # only the last record changed
            np.testing.assert_array_equal(got[:-1], before[:-1])
            assert np.abs(got[-1] - before[-1]).max() > 1e-3
        else:                                        # every class/channel: mean of segment means 0, mean of stds 1
            for k in set(ds.labels.tolist()):
                sel = got[ds.labels.numpy() == k]
                np.testing.assert_allclose(sel.mean(axis=2).mean(axis=0), 0.0, atol=1e-5)
                np.testing.assert_allclose(sel.std(axis=2).mean(axis=0), 1.0, atol=1e-5)


def _eval_data(seed=7, ng=90, nq=33, d=16, ncls=6):
    rng = np.random.default_rng(seed)
    cents = rng.standard_normal((ncls, d)) * 2.0
    gl = rng.integers(0, ncls, ng)
    ql = rng.integers(0, ncls, nq)
    gal = (cents[gl] + rng.standard_normal((ng, d))).astype(np.float32)
    qry = (cents[ql] + rng.standard_normal((nq, d))).astype(np.float32)
    lab = lambda k: {"ClassId": int(k), "ClassName": f"class_{int(k)}", "imagenetClassId": str(int(k))}
    return gal, qry, [lab(k) for k in gl], [lab(k) for k in ql], ncls


class _DS:
    def __init__(self, ncls):
        self.class_id_to_str = {k: f"class_{k}" for k in range(ncls)}
        self.class_str_to_id = {f"class_{k}": k for k in range(ncls)}


def _eval_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from types import SimpleNamespace
    from cerebralsignalnetworks_amd import retrieval
    from oracle import retrieval as oracle_retrieval
    gal, qry, gl, ql, ncls = _eval_data()
    gs, qs = slice(rank, None, world), slice(rank * 17, (rank + 1) * 17 if rank + 1 < world else None)   # ragged shards
    search = lambda g, q, k: oracle_retrieval.l2_topk(g, q, k)                          # the checker stands in for the GPU
    r = retrieval.evaluate_distributed(SimpleNamespace(topK=5), gal[gs], qry[qs], gl[gs], ql[qs], _DS(ncls),
                                       search_fn=search)
    out[rank] = (r["Recall_Total"], r["Precision_Total"], r["top1"], r["I"].shape)
    dist.barrier()
    dist.destroy_process_group()


def test_distributed_retrieval_eval_two_process_gloo():
    """Sharded gallery + sharded queries give every rank the single-process Recall / Precision / top-1 (the search
    itself is the HIP kernel in production; here the oracle's brute force stands in, on CPU)."""
    from oracle import retrieval as oracle_retrieval
    world, port = 2, 29613
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_eval_worker, args=(world, port, out), nprocs=world, join=True)
    gal, qry, gl, ql, ncls = _eval_data()
    # single-process reference over the data in the order the ranks concatenate it
    g_order = np.concatenate([np.arange(len(gal))[r::world] for r in range(world)])
    q_order = np.concatenate([np.arange(len(qry))[r * 17:(r + 1) * 17 if r + 1 < world else None] for r in range(world)])
    ds = _DS(ncls)
    rec, prec = oracle_retrieval.evaluate(gal[g_order], qry[q_order], [gl[i] for i in g_order], [ql[i] for i in q_order],
                                          ds.class_id_to_str, topK=5)[:2]
    for r in range(world):
        assert abs(out[r][0] - rec) < 1e-9 and abs(out[r][1] - prec) < 1e-9, (out[r], rec, prec)
        assert out[r][3] == (len(qry), 5)
    assert out[0] == out[1]



# ---- Barlow-Twins loss over two ranks (net.py:33-42): the product's loss + DDP-mean against the reference's run ----
def _barlow_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from cerebralsignalnetworks_amd.losses import BarlowTwinsLoss
    g = np.load(os.path.join(ROOT, "tests", "golden", "ref_barlow_2rank.npz"))
    n = g["z1"].shape[0] // world
    a = torch.from_numpy(g["z1"][rank * n:(rank + 1) * n]).requires_grad_(True)
    b = torch.from_numpy(g["z2"][rank * n:(rank + 1) * n]).requires_grad_(True)
    crit = BarlowTwinsLoss(g["z1"].shape[1], g["z1"].shape[0]).double().train()      # batch_size = GLOBAL batch
    loss = crit(a, b)
    loss.backward()
    out[rank] = (loss.item(), a.grad.numpy().copy(), b.grad.numpy().copy())
    dist.barrier()
    dist.destroy_process_group()


def test_barlow_loss_two_process_gloo_matches_reference_run():
    """Fixture = the reference's BarlowTwins.forward executed on two gloo ranks (make_ref_goldens.py): the in-place
    all_reduce of c is invisible to autograd, so each rank's gradient goes through its own term only (and DDP then
    averages parameter gradients).  The product's all-reduce must have an identity backward to match."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "ref_barlow_2rank.npz"))
    world, port = 2, 29615
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_barlow_worker, args=(world, port, out), nprocs=world, join=True)
    for r in range(world):
        loss, g1, g2 = out[r]
        np.testing.assert_allclose(loss, g[f"loss_r{r}"], rtol=1e-12)
        np.testing.assert_allclose(g1, g[f"g1_r{r}"], atol=1e-13)
        np.testing.assert_allclose(g2, g[f"g2_r{r}"], atol=1e-13)


# ---- extract_features (PerilsEEGDataset.py:168-226): frozen teacher over all images, gathered on every rank -----------
class _TeacherStandIn(torch.nn.Module):
    """A frozen 'teacher' for the plumbing test: a fixed linear map of the image tensor (DINOv2 needs a network)."""

    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(5)
        self.w = torch.nn.Parameter(torch.randn(12, 7, generator=g), requires_grad=False)

    def forward(self, x):
        return x.reshape(x.shape[0], -1)[:, :12] @ self.w


def _make_image_dataset(tmp, n=11):
    from PIL import Image
    rng = np.random.default_rng(2)
    names = []
    for i in range(n):
        wnid = f"n0{i % 3}"
        os.makedirs(os.path.join(tmp, wnid), exist_ok=True)
        names.append(f"{wnid}_{i}")
        Image.fromarray(rng.integers(0, 255, (4, 4, 3), dtype=np.uint8)).save(os.path.join(tmp, wnid, f"{wnid}_{i}.JPEG"))
    items = [{"eeg": torch.randn(4, 30, generator=torch.Generator().manual_seed(i)), "image": i, "label": i % 3, "subject": 1}
             for i in range(n)]
    path = os.path.join(tmp, "eeg.pth")
    torch.save({"dataset": items, "labels": ["n00", "n01", "n02"], "images": names}, path)
    return path


def _to_tensor(img):
    return torch.from_numpy(np.asarray(img, dtype=np.float32).copy()).permute(2, 0, 1) / 255.0


def _extract_worker(rank, world, port, tmp, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from cerebralsignalnetworks_amd.dataset import EEGDataset
    ds = EEGDataset(eeg_signals_path=os.path.join(tmp, "eeg.pth"), imagesRoot=tmp, time_low=0, time_high=30,
                    device=torch.device("cpu"), preprocessin_fn=_to_tensor)
    ds.extract_features(_TeacherStandIn(), batch_size=4)
    out[rank] = ds.features_all.numpy().copy()
    dist.barrier()
    dist.destroy_process_group()


def test_extract_features_single_and_two_process_gloo(tmp_path):
    from cerebralsignalnetworks_amd.dataset import EEGDataset
    tmp = str(tmp_path)
    path = _make_image_dataset(tmp)
    ds = EEGDataset(eeg_signals_path=path, imagesRoot=tmp, time_low=0, time_high=30, device=torch.device("cpu"),
                    preprocessin_fn=_to_tensor)
    assert not ds.image_features_extracted and ds[0][4] == []
    teacher = _TeacherStandIn()
    ds.extract_features(teacher, batch_size=4)
    want = torch.stack([teacher(ds._load_image(i)[None])[0] for i in range(len(ds))]).numpy()
    np.testing.assert_allclose(ds.features_all.numpy(), want, atol=1e-6)
    assert ds.image_features_extracted and ds[3][4].shape == (7,)
    world, port = 2, 29617
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_extract_worker, args=(world, port, tmp, out), nprocs=world, join=True)
    for r in range(world):                        # every rank ends with the whole table, in dataset order
        np.testing.assert_allclose(out[r], want, atol=1e-6)


# ---- the Spampinato flavour (utils/EEGDataset.py:52-53,99-128): split file, subject filter, per-channel normalisation ---
def test_spampinato_split_subject_filter_and_dataset_level_norm(tmp_path):
    from utils.EEGDataset import EEGDataset
    g = torch.Generator().manual_seed(1)
    items = [{"eeg": torch.randn(5, 40, generator=g) * 3 + 1, "image": k % 4, "label": k % 2, "subject": 1 + k % 3}
             for k in range(12)]
    means, stds = torch.randn(5, 1, generator=g), torch.rand(5, 1, generator=g) + 0.5
    torch.save({"dataset": items, "labels": ["n01", "n02"], "images": ["n01_a", "n02_b", "n01_c", "n02_d"],
                "means": means, "stddevs": stds}, tmp_path / "eeg.pth")
    torch.save({"splits": [{"train": [0, 1, 2, 3, 4, 5, 6, 7], "val": [8, 9], "test": [10, 11]}]}, tmp_path / "splits.pth")
    kw = dict(eeg_signals_path=str(tmp_path / "eeg.pth"), eeg_splits_path=str(tmp_path / "splits.pth"),
              imagesRoot=str(tmp_path), time_low=5, time_high=35, device=torch.device("cpu"))
    ds = EEGDataset(subset="train", subject=2, **kw)                       # subject != 0: that subject only
    keep = [k for k in range(8) if items[k]["subject"] == 2]
    assert len(ds) == len(keep) and ds.subjects.tolist() == [2] * len(keep)
    np.testing.assert_array_equal(ds[0][0].numpy(), items[keep[0]]["eeg"].t()[5:35].numpy())
    ds = EEGDataset(subset="train", subject=0, exclude_subjects=[3], **kw)  # subject == 0: all but the excluded ones
    keep = [k for k in range(8) if items[k]["subject"] != 3]
    assert len(ds) == len(keep) and 3 not in ds.subjects.tolist()
    assert len(EEGDataset(subset="test", subject=0, **kw)) == 2
    ds = EEGDataset(subset="val", subject=0, apply_norm_with_stds_and_means=True, **kw)
    want = ((items[8]["eeg"] - means) / stds).t()[5:35]                     # (eeg - means) / stddevs at load, :104-105
    np.testing.assert_allclose(ds[0][0].numpy(), want.numpy(), atol=1e-6)
    assert ds[0][1]["ClassId"] == 0 and ds.getLabelbyIndex(1)["ClassId"] == 1


def test_bench_launcher_argv_env_and_rank_count_refusals():
    """`bench.py --gpus N`: the parent builds a torch.distributed.run job of N fresh rank processes (it never touches
    the GPU itself), a rank whose WORLD_SIZE disagrees with --gpus exits non-zero, and a host with fewer GPUs than
    asked for is refused instead of silently benching one device (reference launch pattern:
    EEG-BarlowNetworks/train.py:71,76-78 -- one worker per GPU over a loopback tcp rendezvous)."""
    import subprocess
    sys.path.insert(0, ROOT)
    import bench
    argv = ["--gpus", "4", "--steps", "7", "--warmup", "2"]
    cmd, env = bench.launcher_command(4, argv, port=29876)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29876"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == argv                       # the ranks get the caller's flags verbatim (incl. --gpus 4)
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and "WORLD_SIZE" not in env and "RANK" not in env
    assert bench.parse(argv).gpus == 4
    assert "CSN_NO_PERSIST" not in env or os.environ.get("CSN_NO_PERSIST")
    # one-device rehearsal: the ranks' launches must be the per-diagonal ones (a weight-stationary launch owns every CU)
    os.environ["CSN_SINGLE_DEVICE"] = "1"
    try:
        assert bench.launcher_command(2, ["--gpus", "2"], port=29877)[1]["CSN_NO_PERSIST"] == "1"
    finally:
        os.environ.pop("CSN_SINGLE_DEVICE")
    # no GPU in this container: asking for 2 must fail loudly, not print an n_gpus=1 line
    clean = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "CSN_SINGLE_DEVICE")}
    if torch.cuda.device_count() < 2:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=clean,
                           capture_output=True, text=True, timeout=300)
        assert r.returncode != 0 and r.stdout.strip() == "" and "refusing" in r.stderr
    # a rank launched with the wrong world size exits before touching any device
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=dict(clean, WORLD_SIZE="3", RANK="0"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and r.stdout.strip() == "" and "WORLD_SIZE=3" in r.stderr
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"], env=dict(clean, WORLD_SIZE="2", RANK="0"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and r.stdout.strip() == ""
