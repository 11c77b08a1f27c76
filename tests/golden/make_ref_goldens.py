"""Reference-pinned fixtures: made by EXECUTING the reference's own hot-path definitions (run once, here).

    python tests/golden/make_ref_goldens.py [small|cfg2|cfg4|retrieval|retrieval_cfg4|all]

``ref_lift.lift`` compiles the named class / function nodes straight out of the files under /root/reference
(which cannot be imported as modules: faiss / torchvision / cv2 / librosa / ``models.lstm`` are absent) and
runs them on seeded inputs.  Executed definitions (file:line):

  LSTMDistillRetreival.py:85-110       LSTMModel (last step -> fc), incl. its .view(B, C, T) reshape
  LSTMDistill.py:112-142               LSTMModel (every step -> fc -> class_pred, ReLU)
  LstmDistillFromDinoV2Train.py:16-25,36-43,107-140     HyperParams, CosineSimilarityLoss, FeatureDistributionLoss
  LstmDistillFromDinoV2TrainSpampinato.py:107-121,125-184   loss_fn_kd, FeatureDistributionLoss (KD form)
  LstmDistillFromDinoV2Eval.py:106-146                  FeatureDistributionLoss (soft-target form)
  LstmDistillation.py:66-99,101-159,161-172             DINOHead, DINOLoss, FeatureDistributionLoss (MSE-stat form)
  EEG-BarlowNetworks/net.py:6-9,33-42                   off_diagonal, BarlowTwins.forward (identity backbones; 1 and 2 gloo ranks)
  EEG-BarlowNetworks/barlow_utils.py:8-21               adjust_learning_rate
  utils/PerilsEEGDataset.py:454-461,464-507,541-623     normlizeEEG, transformEEGDataToChannelWiseNorm, __getitem__
  utils/EEGDataset.py:539-591                           __getitem__ (Spampinato dataset-level (x - means) / stddevs)
  utils/Utilities.py:411-428                            Utilities.remove_noise (zero-phase band-pass)
  utils/utils.py:594-630 (imported)                     MultiCropWrapper, cosine_scheduler, trunc_normal_

Not executable here, therefore still restated only: ``evaluate`` (faiss), ``extract_features`` /
``transformEEGDataLSTMByList`` (hard-coded ``.cuda()``).

Only arrays are written (inputs + what the reference computed); /root/reference is never read at test time.
The full-size fixtures keep every bias / fc gradient in full and, of the large weight gradients, a strided
sample, the Frobenius norm and two seeded random projections (so every element is still covered).
"""
import contextlib
import io
import os
import sys
import tempfile
import time
import types
import warnings

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from ref_lift import REF, base_namespace, lift   # noqa: E402

warnings.filterwarnings("ignore")


def _save(name, out):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **out)
    print(f"{name}: {os.path.getsize(path)} bytes, {len(out)} arrays")


def _quiet():
    return contextlib.redirect_stdout(io.StringIO())


@contextlib.contextmanager
def _default_dtype(dt):
    """The reference builds its zero initial state with torch.zeros(...): default dtype decides f32 / f64."""
    old = torch.get_default_dtype()
    torch.set_default_dtype(dt)
    try:
        yield
    finally:
        torch.set_default_dtype(old)


def _load_into(model, params, dtype):
    sd = {k: torch.from_numpy(np.asarray(v)).to(dtype) for k, v in params.items()}
    model.to(dtype).load_state_dict(sd)
    return model


def _ensure_pg():
    if not dist.is_initialized():
        f = tempfile.NamedTemporaryFile(delete=False)
        dist.init_process_group("gloo", init_method=f"file://{f.name}", rank=0, world_size=1)


# ------------------------------------------------------------------------------------------------
# encoders
# ------------------------------------------------------------------------------------------------
def ref_lstm_small():
    from oracle.lstm import init_params
    out = {}
    # (a) LSTMDistillRetreival.LSTMModel: x[B, timespan, channels] is *viewed* as [B, channels, timespan]
    ns = lift("LSTMDistillRetreival.py", ["LSTMModel"])
    cos = lift("LstmDistillFromDinoV2Train.py", ["CosineSimilarityLoss"])["CosineSimilarityLoss"]()
    B, TS, CH, H, L, D = 4, 128, 32, 64, 2, 48
    params = init_params(TS, H, L, D, None, seed=43)
    rng = np.random.default_rng(21)
    x = rng.standard_normal((B, TS, CH)).astype(np.float32)
    tgt = rng.standard_normal((B, D)).astype(np.float32)
    out.update(a_x=x, a_target=tgt, a_dims=np.array([B, TS, CH, H, L, D]), a_seed_params=np.array(43))
    for name, dt in (("f64", torch.float64), ("f32", torch.float32)):
        with _default_dtype(dt):
            m = _load_into(ns["LSTMModel"](TS, H, L, D), params, dt)
            feat = m(torch.from_numpy(x).to(dt))
            loss = cos(feat, torch.from_numpy(tgt).to(dt))
            loss.backward()
        out[f"a_feat_{name}"] = feat.detach().numpy()
        out[f"a_loss_{name}"] = np.array(loss.item())
        if name == "f64":
            for k, p in m.named_parameters():
                out[f"a_grad__{k}"] = p.grad.numpy()
    # (b) LSTMDistill.LSTMModel: fc + class_pred on every step, ReLU on the features
    ns = lift("LSTMDistill.py", ["LSTMModel"])
    B, TS, CH, H, L, D, NC = 3, 128, 20, 64, 3, 48, 40
    params = init_params(TS, H, L, D, NC, seed=44)
    x = rng.standard_normal((B, TS, CH)).astype(np.float32)
    wf = rng.standard_normal((B, CH, D))
    wc = rng.standard_normal((B, CH, NC))
    out.update(b_x=x, b_wf=wf, b_wc=wc, b_dims=np.array([B, TS, CH, H, L, D, NC]), b_seed_params=np.array(44))
    m = _load_into(ns["LSTMModel"](TS, H, L, D, NC), params, torch.float64)
    with _quiet(), _default_dtype(torch.float64):
        feat, cls = m(torch.from_numpy(x).double())
    loss = (feat * torch.from_numpy(wf)).sum() + (cls * torch.from_numpy(wc)).sum()
    loss.backward()
    out.update(b_feat=feat.detach().numpy(), b_cls=cls.detach().numpy(), b_loss=np.array(loss.item()))
    for k, p in m.named_parameters():
        out[f"b_grad__{k}"] = p.grad.numpy()
    _save("ref_lstm.npz", out)


def _sample_grad(name, g, out, rng_seed=5):
    g = np.asarray(g, np.float64)
    if g.ndim == 1 or g.size <= 400000:
        out[f"grad__{name}"] = g.astype(np.float32) if g.size > 4096 else g
        return
    r = np.random.default_rng(rng_seed)
    out[f"gsamp__{name}"] = g[::37, ::41].copy()
    out[f"gnorm__{name}"] = np.array(np.linalg.norm(g))
    out[f"gprojr__{name}"] = g @ r.standard_normal(g.shape[1])        # [rows]
    out[f"gprojl__{name}"] = r.standard_normal(g.shape[0]) @ g        # [cols]


def ref_lstm_full(tag, B, T, C, H, L, D, seed_x):
    """Full-size training step of the reference's LSTMModel + CosineSimilarityLoss (torch CPU, f64 and f32)."""
    from oracle.lstm import init_params
    ns = lift("LSTMDistillRetreival.py", ["LSTMModel"])
    cos = lift("LstmDistillFromDinoV2Train.py", ["CosineSimilarityLoss"])["CosineSimilarityLoss"]()
    params = init_params(C, H, L, D, None, seed=43)
    rng = np.random.default_rng(seed_x)
    x = rng.standard_normal((B, T, C)).astype(np.float32)        # the sequence the LSTM must see: [B, T, C]
    tgt = rng.standard_normal((B, D)).astype(np.float32)
    # the model *views* its input [B, timespan, channels] as [B, channels, timespan]: hand it the same memory
    x_in = torch.from_numpy(x).reshape(B, C, T)
    out = dict(dims=np.array([B, T, C, H, L, D]), seed_params=np.array(43), seed_x=np.array(seed_x))
    torch.set_num_threads(8)
    for name, dt in (("f32", torch.float32), ("f64", torch.float64)):
        t0 = time.time()
        with _default_dtype(dt):
            m = _load_into(ns["LSTMModel"](C, H, L, D), params, dt)
            feat = m(x_in.to(dt))
            loss = cos(feat, torch.from_numpy(tgt).to(dt))
            loss.backward()
        out[f"feat_{name}"] = feat.detach().numpy()
        out[f"loss_{name}"] = np.array(loss.item())
        if name == "f64":
            for k, p in m.named_parameters():
                _sample_grad(k, p.grad.numpy(), out)
        else:
            for k, p in m.named_parameters():
                out[f"gnorm_f32__{k}"] = np.array(float(p.grad.double().norm()))
        print(f"  {tag} {name}: loss {loss.item():.8f}  ({time.time() - t0:.1f} s)", flush=True)
    _save(f"ref_lstm_{tag}.npz", out)


# ------------------------------------------------------------------------------------------------
# losses
# ------------------------------------------------------------------------------------------------
def ref_losses():
    _ensure_pg()
    rng = np.random.default_rng(3)
    B, D, NC = 16, 384, 40
    s = rng.standard_normal((B, D)).astype(np.float32)
    t = rng.standard_normal((B, D)).astype(np.float32)
    cls = rng.standard_normal((B, NC)).astype(np.float32)
    tcls = rng.standard_normal((B, NC)).astype(np.float32)
    lab = rng.integers(0, NC, B)
    out = dict(student=s, teacher=t, cls=cls, tcls=tcls, labels=lab)
    S64 = lambda a: torch.from_numpy(a).double().requires_grad_(True)   # noqa: E731
    Lb = torch.from_numpy(lab)

    tr = lift("LstmDistillFromDinoV2Train.py", ["HyperParams", "CosineSimilarityLoss", "FeatureDistributionLoss"])
    sg = S64(s)
    loss = tr["CosineSimilarityLoss"]()(sg, torch.from_numpy(t).double())
    loss.backward()
    out.update(cosine_loss=np.array(loss.item()), cosine_grad=sg.grad.numpy())
    hp = tr["HyperParams"]
    fd = tr["FeatureDistributionLoss"](100, hp.warmup_teacher_temp, hp.teacher_temp, hp.warmup_teacher_temp_epochs)
    out["temp_schedule_100"] = np.asarray(fd.teacher_temp_schedule)
    for ep in (0, 25, 50, 99):
        sg, cg = S64(s), S64(cls)
        loss = fd(sg, torch.from_numpy(t).double(), ep, Lb, pred_label=cg)
        loss.backward()
        out[f"featdist_ep{ep}"] = np.array(loss.item())
        out[f"featdist_ep{ep}_gs"] = sg.grad.numpy()
        out[f"featdist_ep{ep}_gc"] = cg.grad.numpy()

    sp = lift("LstmDistillFromDinoV2TrainSpampinato.py", ["HyperParams", "loss_fn_kd", "FeatureDistributionLoss"])
    for alpha, temp in ((1.0, 2.0), (0.5, 4.0), (0.9, 20.0)):
        cg = S64(cls)
        loss = sp["loss_fn_kd"](cg, Lb, torch.from_numpy(tcls).double(), types.SimpleNamespace(alpha=alpha, temperature=temp))
        loss.backward()
        out[f"kd_a{alpha}_T{temp}"] = np.array(loss.item())
        out[f"kd_a{alpha}_T{temp}_g"] = cg.grad.numpy()
    hp = sp["HyperParams"]
    fd = sp["FeatureDistributionLoss"](100, hp.warmup_teacher_temp, hp.teacher_temp, hp.warmup_teacher_temp_epochs)
    for ep in (0, 25, 50):
        cg = S64(cls)
        loss = fd(cg, torch.from_numpy(tcls).double(), ep, Lb)
        loss.backward()
        out[f"featdist_spamp_ep{ep}"] = np.array(loss.item())
        out[f"featdist_spamp_ep{ep}_g"] = cg.grad.numpy()
    out["spamp_weights"] = np.array([hp.soft_target_loss_weight, hp.ce_loss_weight, hp.warmup_teacher_temp,
                                     hp.teacher_temp, hp.warmup_teacher_temp_epochs], dtype=np.float64)

    ev = lift("LstmDistillFromDinoV2Eval.py", ["HyperParams", "FeatureDistributionLoss"])
    hp = ev["HyperParams"]
    fd = ev["FeatureDistributionLoss"](100, hp.warmup_teacher_temp, hp.teacher_temp, hp.warmup_teacher_temp_epochs)
    for ep in (0, 50):
        sg = S64(s)
        loss = fd(sg, torch.from_numpy(t).double(), ep)
        loss.backward()
        out[f"featdist_eval_ep{ep}"] = np.array(loss.item())
        out[f"featdist_eval_ep{ep}_g"] = sg.grad.numpy()
    out["eval_temps"] = np.array([hp.warmup_teacher_temp, hp.teacher_temp, hp.warmup_teacher_temp_epochs], dtype=np.float64)

    # DINO pieces of LstmDistillation.py (trunc_normal_ comes from the vendored utils/utils.py, which imports here)
    import importlib.util
    sys.path.insert(0, REF)
    spec = importlib.util.spec_from_file_location("ref_utils_utils", os.path.join(REF, "utils", "utils.py"))
    ru = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ru)
    dn = lift("LstmDistillation.py", ["DINOHead", "DINOLoss", "FeatureDistributionLoss"],
              base_namespace(trunc_normal_=ru.trunc_normal_, utils=ru))
    sg = S64(s)
    loss = dn["FeatureDistributionLoss"]()(sg, torch.from_numpy(t).double())
    loss.backward()
    out.update(featdist_mse=np.array(loss.item()), featdist_mse_g=sg.grad.numpy())
    torch.manual_seed(5)
    head = dn["DINOHead"](16, 12, nlayers=3, hidden_dim=32, bottleneck_dim=8).double()
    hx = rng.standard_normal((6, 16))
    for k, v in head.state_dict().items():
        out["dinohead_sd__" + k] = v.numpy()
    out.update(dinohead_x=hx, dinohead_y=head(torch.from_numpy(hx)).detach().numpy())
    # DINOLoss: 6 student views x 5 samples against 2 teacher views x 5 samples, two successive steps (centre EMA)
    # (LstmDistillation.py:583-586 stacks the views: student [6, B, out], teacher [2, B, out])
    so = rng.standard_normal((2, 6, 5, 32))
    to = rng.standard_normal((2, 2, 5, 32))
    crit = dn["DINOLoss"](32, 6, 0.04, 0.07, 3, 10).double()
    out.update(dino_student=so, dino_teacher=to)
    for step in range(2):
        sg = S64(so[step])
        loss = crit(sg, torch.from_numpy(to[step]), step + 1)
        loss.backward()
        out[f"dino_loss{step}"] = np.array(loss.item())
        out[f"dino_grad{step}"] = sg.grad.numpy()
        out[f"dino_center{step}"] = crit.center.numpy().copy()
    out["cosine_scheduler"] = ru.cosine_scheduler(0.0005, 1e-6, 10, 7, warmup_epochs=2)
    out["cosine_scheduler_momentum"] = ru.cosine_scheduler(0.996, 1, 5, 11)

    # Barlow-Twins loss: the reference's forward with identity backbones / projector, one rank
    bl = lift("EEG-BarlowNetworks/net.py", ["off_diagonal", "BarlowTwins.forward"])
    z1 = rng.standard_normal((32, 96))
    z2 = z1 + 0.3 * rng.standard_normal((32, 96))
    me = types.SimpleNamespace(projector=nn.Identity(), backbone_image=nn.Identity(), backbone_eeg=nn.Identity(),
                               bn=nn.BatchNorm1d(96, affine=False).double().train(),
                               args=types.SimpleNamespace(batch_size=32, lambd=0.0051))
    a, b = S64(z1), S64(z2)
    loss = bl["BarlowTwins.forward"](me, a, b)
    loss.backward()
    out.update(barlow_z1=z1, barlow_z2=z2, barlow_loss=np.array(loss.item()), barlow_g1=a.grad.numpy(), barlow_g2=b.grad.numpy())
    c = rng.standard_normal((7, 7))
    out.update(offdiag_in=c, offdiag_out=bl["off_diagonal"](torch.from_numpy(c)).numpy())

    # LR schedule of the Barlow trainer: args / optimizer / loader are plain holders of the fields it reads
    al = lift("EEG-BarlowNetworks/barlow_utils.py", ["adjust_learning_rate"])
    args = types.SimpleNamespace(epochs=20, batch_size=512, learning_rate_weights=0.2, learning_rate_biases=0.0048)
    opt = types.SimpleNamespace(param_groups=[{}, {}])
    lrs = []
    for step in range(100):
        al["adjust_learning_rate"](args, opt, [None] * 5, step)
        lrs.append([opt.param_groups[0]["lr"], opt.param_groups[1]["lr"]])
    out["barlow_lr_table"] = np.array(lrs)
    _save("ref_losses.npz", out)


def _barlow_rank(rank, world, path, z1, z2, q):
    dist.init_process_group("gloo", init_method=f"file://{path}", rank=rank, world_size=world)
    bl = lift("EEG-BarlowNetworks/net.py", ["off_diagonal", "BarlowTwins.forward"])
    n = z1.shape[0] // world
    a = torch.from_numpy(z1[rank * n:(rank + 1) * n]).double().requires_grad_(True)
    b = torch.from_numpy(z2[rank * n:(rank + 1) * n]).double().requires_grad_(True)
    me = types.SimpleNamespace(projector=nn.Identity(), backbone_image=nn.Identity(), backbone_eeg=nn.Identity(),
                               bn=nn.BatchNorm1d(z1.shape[1], affine=False).double().train(),
                               args=types.SimpleNamespace(batch_size=z1.shape[0], lambd=0.0051))
    loss = bl["BarlowTwins.forward"](me, a, b)
    loss.backward()
    q.put((rank, loss.item(), a.grad.numpy(), b.grad.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def ref_barlow_2rank():
    """net.py:33-42 on two gloo ranks: per-rank BatchNorm statistics, c all-reduced in place (invisible to autograd),
    so each rank's gradient is d loss(global c) / d (its own z).  DDP's mean over ranks is applied in the test."""
    import torch.multiprocessing as mp
    rng = np.random.default_rng(9)
    z1 = rng.standard_normal((32, 48))
    z2 = z1 + 0.3 * rng.standard_normal((32, 48))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    f = tempfile.NamedTemporaryFile(delete=False)
    procs = [ctx.Process(target=_barlow_rank, args=(r, 2, f.name, z1, z2, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)])
    for p in procs:
        p.join()
    out = dict(z1=z1, z2=z2)
    for rank, loss, g1, g2 in res:
        out[f"loss_r{rank}"] = np.array(loss)
        out[f"g1_r{rank}"] = g1
        out[f"g2_r{rank}"] = g2
    _save("ref_barlow_2rank.npz", out)


# ------------------------------------------------------------------------------------------------
# preprocessing / dataset items
# ------------------------------------------------------------------------------------------------
def ref_preproc():
    from PIL import Image
    rng = np.random.default_rng(17)
    out = {}
    pe = lift("utils/PerilsEEGDataset.py", ["EEGDataset.normlizeEEG", "EEGDataset.__getitem__",
                                            "EEGDataset.transformEEGDataToChannelWiseNorm", "EEGDataset.__len__"],
              base_namespace(Image=Image))
    norm = pe["EEGDataset.normlizeEEG"]
    # normlizeEEG: numpy input -> ndarray.std() (ddof 0); torch input -> Tensor.std() (ddof 1)
    e = rng.standard_normal((60, 5)).astype(np.float32) * 3 + 1
    en = e.copy()
    for ch in range(5):
        en = norm(None, en, ch)
    et = torch.from_numpy(e.copy())
    for ch in range(5):
        et = norm(None, et, ch)
    out.update(norm_in=e, norm_np=en, norm_torch=et.numpy())

    # remove_noise (zero-phase butter-4 1-50 Hz, filtfilt per (sample, channel))
    ut = lift("utils/Utilities.py", ["Utilities.remove_noise"])
    xs = rng.standard_normal((2, 500, 8))
    out.update(filtfilt_x=xs, filtfilt_y=ut["Utilities.remove_noise"](None, xs, 1000))

    # __getitem__ of the Perils dataset: records [C, T_raw] -> eeg[T, C] window; channel subset + per-channel
    # z-score; scalar dataset-level normalisation.  `self` carries exactly the attributes the method reads.
    tmp = tempfile.mkdtemp()
    os.makedirs(f"{tmp}/n01", exist_ok=True)
    Image.fromarray(np.zeros((8, 8, 3), np.uint8)).save(f"{tmp}/n01/n01_7.JPEG")
    C, TR, N = 12, 96, 6
    raw = (rng.standard_normal((N, C, TR)) * 2 + 0.5).astype(np.float32)
    out["item_raw"] = raw

    def make_self(**kw):
        me = types.SimpleNamespace(
            subsetData=[{"eeg": torch.from_numpy(raw[i].copy()), "label": 0} for i in range(N)],
            Transform_EEG2Image_Shape=False, isDataTransformed=False, filter_channels=[], time_low=10, time_high=70,
            apply_channel_wise_norm=False, apply_norm_with_stds_and_means=False, data_augment_eeg=False,
            add_channel_dim_to_eeg=False, images=["n01_7"] * N, imagesRoot=tmp,
            class_labels_names={"n01": {"ClassId": 3, "ClassName": "cat", "imagenetClassId": "7"}},
            inference_mode=True, onehotencode_label=False, preprocessin_fn=lambda im: 0,
            image_features_extracted=False, image_features=[], mean=0, std=1)
        me.normlizeEEG = lambda EEG, ch_index, class_index=None: norm(me, EEG, ch_index, class_index)
        for k, v in kw.items():
            setattr(me, k, v)
        return me

    get = pe["EEGDataset.__getitem__"]
    out["item_plain"] = np.stack([get(make_self(), i)[0].numpy() for i in range(N)])
    me = make_self(filter_channels=[7, 2, 9], apply_channel_wise_norm=True)
    me.isDataTransformed = False
    out["item_subset_norm"] = np.stack([get(me, i)[0].numpy() for i in range(N)])     # [N, 3, 60] (channel-first: its .t())
    mean = float(np.mean([raw[i].mean() for i in range(N)]))
    std = float(np.mean([torch.from_numpy(raw[i]).std().item() for i in range(N)]))
    out["item_scalar_norm"] = np.stack([get(make_self(apply_norm_with_stds_and_means=True, mean=mean, std=std), i)[0].numpy()
                                        for i in range(N)])
    out["item_scalar_mean_std"] = np.array([mean, std])

    # transformEEGDataToChannelWiseNorm (:464-507).  What it leaves behind depends on the STORED dtype: a float64
    # record (what ConvertToPth.py:186 writes from mne data) is copied by ``.float()``, so only record N-1 -- the
    # stale loop index of :507 -- is ever overwritten; a float32 record is aliased by ``.float().cpu().numpy()``, so
    # every visited record is changed in place as well.
    class Holder:
        pass
    labels = [0, 1, 0, 2, 1, 0, 1]
    N2 = len(labels)
    raw2 = (rng.standard_normal((N2, C, TR)) * 2 + 0.5).astype(np.float32)
    out["cwn_raw"] = raw2
    out["cwn_labels"] = np.array(labels)
    for k in set(labels):
        os.makedirs(f"{tmp}/n0{k}", exist_ok=True)
        Image.fromarray(np.zeros((8, 8, 3), np.uint8)).save(f"{tmp}/n0{k}/n0{k}_7.JPEG")
    Holder.__len__ = lambda self: N2
    Holder.__getitem__ = lambda self, i: get(self, i)
    for tag, store in (("f64", torch.float64), ("f32", torch.float32)):
        ds = Holder()
        for k, v in vars(make_self(time_low=4, time_high=TR - 6)).items():
            setattr(ds, k, v)
        ds.subsetData = [{"eeg": torch.from_numpy(raw2[i].copy()).to(store), "label": labels[i]} for i in range(N2)]
        ds.images = [f"n0{k}_7" for k in labels]
        ds.class_labels_names = {f"n0{k}": {"ClassId": k, "ClassName": f"c{k}", "imagenetClassId": "7"} for k in set(labels)}
        with _quiet():
            pe["EEGDataset.transformEEGDataToChannelWiseNorm"](ds)
        out[f"cwn_after_{tag}"] = np.stack([np.asarray(ds.subsetData[i]["eeg"], dtype=np.float32) for i in range(N2)])
    out["cwn_window"] = np.array([4, TR - 6])

    # the Spampinato dataset file carries the same item code (utils/EEGDataset.py:539-567); its per-channel
    # (eeg - means) / stddevs happens in __init__ (:104-105), which needs torchvision -> restated only
    sp = lift("utils/EEGDataset.py", ["EEGDataset.__getitem__", "EEGDataset.normlizeEEG"], base_namespace(Image=Image))
    out["item_plain_spamp"] = np.stack([sp["EEGDataset.__getitem__"](make_self(), i)[0].numpy() for i in range(N)])
    _save("ref_preproc.npz", out)


def ref_dino_step():
    """One optimisation step's loss / gradients of the DINO self-distillation trainer (LstmDistillation.py:518-596):
    the reference's MultiCropWrapper (utils/utils.py:594-630), DINOHead and DINOLoss executed around a backbone.
    The backbone class itself (``models.lstm.Model``) is absent from the reference tree: a torch ``nn.LSTM`` ->
    last step module with the call-site contract stands in for it here (third-party call; the product's HIP LSTM is
    what the GPU test puts in its place)."""
    _ensure_pg()
    import importlib.util
    from oracle.lstm import init_params
    sys.path.insert(0, REF)
    spec = importlib.util.spec_from_file_location("ref_utils_utils", os.path.join(REF, "utils", "utils.py"))
    ru = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ru)
    dn = lift("LstmDistillation.py", ["DINOHead", "DINOLoss"], base_namespace(trunc_normal_=ru.trunc_normal_, utils=ru))
    B, C, H, L, OUT = 4, 16, 32, 2, 24

    class Backbone(nn.Module):
        def __init__(self):
            super().__init__()
            self.lstm = nn.LSTM(C, H, num_layers=L, batch_first=True)
            self.fc = nn.Linear(H, H)

        def forward(self, x):
            return self.fc(self.lstm(x)[0][:, -1, :])

    params = init_params(C, H, L, H, None, seed=47)
    rng = np.random.default_rng(23)
    views = [rng.standard_normal((B, 40, C)).astype(np.float32) for _ in range(2)] + \
            [rng.standard_normal((B, 24, C)).astype(np.float32) for _ in range(4)]
    torch.manual_seed(9)
    with _default_dtype(torch.float64):
        head_s = dn["DINOHead"](H, OUT, nlayers=3, hidden_dim=48, bottleneck_dim=16)
        head_t = dn["DINOHead"](H, OUT, nlayers=3, hidden_dim=48, bottleneck_dim=16)
        student = ru.MultiCropWrapper(_load_into(Backbone(), params, torch.float64), head_s).double()
        teacher = ru.MultiCropWrapper(_load_into(Backbone(), params, torch.float64), head_t).double()
        teacher.load_state_dict(student.state_dict())
        crit = dn["DINOLoss"](OUT, 6, 0.04, 0.07, 3, 10).double()
        vt = [torch.from_numpy(v).double() for v in views]
        with torch.no_grad():
            teacher_outputs = torch.stack([teacher(v) for v in vt[:2]], dim=0)
        student_outputs = torch.stack([student(v) for v in vt], dim=0)
        loss = crit(student_outputs, teacher_outputs, 0)
        loss.backward()
    out = dict(dims=np.array([B, C, H, L, OUT]), seed_params=np.array(47), loss=np.array(loss.item()),
               center=crit.center.numpy().copy(), student_out=student_outputs.detach().numpy())
    for i, v in enumerate(views):
        out[f"view{i}"] = v
    for k, v in student.state_dict().items():
        if k.startswith("head."):
            out["sd__" + k] = v.numpy()
    for k, p in student.named_parameters():
        if p.grad is not None:
            out["grad__" + k] = p.grad.numpy()
    _save("ref_dino_step.npz", out)


# ------------------------------------------------------------------------------------------------
# retrieval acceptance set (north star: bf16 top-1 within +-0.5 % of the CPU reference)
# ------------------------------------------------------------------------------------------------
def ref_retrieval(n_gallery=2048, n_query=512, tag="cfg2", T=500, H=768, seed=101):
    """The reference CPU path (scipy sosfilt + z-score -> the reference's LSTMModel, torch f32, eval) embeds a seeded
    clustered set at cfg2 (128 x 500, hidden 768) or cfg4 (the Spampinato shapes of BASELINE.json configs[3]: 128 x 440,
    hidden 1024, LstmDistillFromDinoV2TrainSpampinato.py:368; smaller set: the CPU run is the cost); stored: its top-5 neighbour lists / top-1 (exact L2 in f64), labels, a sample of the
    embeddings.  Inputs are regenerated from the seed at test time (cerebralsignalnetworks_amd.dataset.clustered_eeg)."""
    from oracle.lstm import init_params
    from oracle import cpu_path, eeg_filter, retrieval
    from cerebralsignalnetworks_amd.dataset import clustered_eeg
    C, L, D = 128, 2, 384
    ns = lift("LSTMDistillRetreival.py", ["LSTMModel"])
    params = init_params(C, H, L, D, None, seed=43)
    m = _load_into(ns["LSTMModel"](C, H, L, D), params, torch.float32).eval()
    n = n_gallery + n_query
    x, labels = clustered_eeg(n, T=T, seed=seed)
    sos = eeg_filter.design_bandpass_sos(1000, 3)
    torch.set_num_threads(8)
    embs = []
    t0 = time.time()
    with torch.no_grad():
        for i in range(0, n, 128):
            eeg = torch.from_numpy(cpu_path.preprocess_scipy(x[i:i + 128], sos))        # [b, T, C]
            embs.append(m(eeg.reshape(eeg.shape[0], C, T)).numpy())                       # (viewed back to [b, T, C] inside)
            print(f"  retrieval embed {i + 128}/{n}  {time.time() - t0:.0f} s", flush=True)
    emb = np.concatenate(embs)
    Dm, I = retrieval.l2_topk(emb[:n_gallery], emb[n_gallery:], 5)
    top1 = float((labels[:n_gallery][I[:, 0]] == labels[n_gallery:]).mean())
    print("  reference top-1:", top1)
    _save(f"ref_retrieval_{tag}.npz", dict(n_gallery=np.array(n_gallery), n_query=np.array(n_query), seed=np.array(seed),
                                         dims=np.array([C, T, H, L, D]),
                                         snr=np.array(0.2), labels=labels.astype(np.int16), top5=I.astype(np.int32),
                                         top5_dist=Dm, top1=np.array(top1), emb_sample=emb[::16].astype(np.float32)))


# ------------------------------------------------------------------------------------------------
# runtime helpers of utils/utils.py (meters, gradient clipping, parameter groups, accuracy)
# ------------------------------------------------------------------------------------------------
def ref_runtime():
    """Executes the reference's SmoothedValue / MetricLogger / clip_gradients / cancel_gradients_last_layer /
    get_params_groups / bool_flag / accuracy (utils/utils.py:132-149, 201-212, 224-284, 313-347, 506-513, 636-647)
    on seeded inputs.  Strings are stored as fixed-width unicode arrays (no pickle)."""
    import argparse
    import datetime
    import time
    from collections import defaultdict, deque
    ns = lift("utils/utils.py", ["SmoothedValue", "MetricLogger", "clip_gradients", "cancel_gradients_last_layer",
                                 "get_params_groups", "bool_flag", "accuracy", "is_dist_avail_and_initialized",
                                 "get_world_size", "get_rank", "is_main_process", "reduce_dict", "has_batchnorms"],
              base_namespace(defaultdict=defaultdict, deque=deque, time=time, datetime=datetime, argparse=argparse))
    out = {}
    rng = np.random.default_rng(77)
    series = rng.standard_normal(37).round(3)
    for win in (20, 4, 5):
        sv = ns["SmoothedValue"](window_size=win)
        stats = []
        for i, v in enumerate(series):
            sv.update(float(v), n=1 + (i % 3))
            stats.append([sv.median, sv.avg, sv.global_avg, sv.max, sv.value])
        out[f"smoothed_w{win}"] = np.array(stats, dtype=np.float64)
        out[f"smoothed_w{win}_str"] = np.array(str(sv))
    out["series"] = series
    ml = ns["MetricLogger"](delimiter="  ")
    for i, v in enumerate(series[:11]):
        ml.update(loss=torch.tensor(float(v)), lr=0.001 * (i + 1), step=i)
    out["logger_str"] = np.array(str(ml))
    out["logger_loss_global_avg"] = np.array(ml.loss.global_avg)
    # clip_gradients / cancel_gradients_last_layer / get_params_groups on a small seeded module tree
    torch.manual_seed(5)

    class Head(nn.Module):
        def __init__(self):
            super().__init__()
            self.mlp = nn.Linear(6, 5)
            self.norm = nn.LayerNorm(5)
            self.last_layer = nn.Linear(5, 3, bias=False)

        def forward(self, x):
            return self.last_layer(self.norm(self.mlp(x)))

    net = Head()
    x = torch.from_numpy(rng.standard_normal((4, 6)).astype(np.float32))
    (net(x) ** 2).sum().backward()
    names = [n for n, _ in net.named_parameters()]
    out["clip_names"] = np.array(names)
    for n, p in net.named_parameters():
        out[f"clip_param__{n}"] = p.detach().numpy().copy()
        out[f"clip_grad_before__{n}"] = p.grad.numpy().copy()
    out["clip_x"] = x.numpy()
    for clip in (0.3, 3.0):
        saved = [p.grad.clone() for p in net.parameters()]
        norms = ns["clip_gradients"](net, clip)
        out[f"clip{clip}_norms"] = np.array(norms)
        for n, p in net.named_parameters():
            out[f"clip{clip}_grad_after__{n}"] = p.grad.numpy().copy()
        for p, g in zip(net.parameters(), saved):
            p.grad = g
    groups = ns["get_params_groups"](net)
    idx = {id(p): n for n, p in net.named_parameters()}
    out["groups_regularized"] = np.array([idx[id(p)] for p in groups[0]["params"]])
    out["groups_not_regularized"] = np.array([idx[id(p)] for p in groups[1]["params"]])
    out["groups_wd1"] = np.array(groups[1]["weight_decay"])
    ns["cancel_gradients_last_layer"](0, net, 1)
    out["cancel_epoch0_none"] = np.array([n for n, p in net.named_parameters() if p.grad is None])
    logits = torch.from_numpy(rng.standard_normal((16, 10)).astype(np.float32))
    target = torch.from_numpy(rng.integers(0, 10, 16))
    out["acc_logits"], out["acc_target"] = logits.numpy(), target.numpy()
    out["acc_top1_5"] = np.array([float(a) for a in ns["accuracy"](logits, target, topk=(1, 5))])
    out["bool_flag_true"] = np.array([ns["bool_flag"](s) for s in ("on", "True", "1")])
    out["bool_flag_false"] = np.array([ns["bool_flag"](s) for s in ("off", "FALSE", "0")])
    _save("ref_runtime.npz", out)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "small"
    torch.manual_seed(0)
    if what in ("small", "all"):
        ref_lstm_small()
        ref_losses()
        ref_barlow_2rank()
        ref_preproc()
        ref_dino_step()
    if what in ("dino",):
        ref_dino_step()
    if what in ("runtime", "small", "all"):
        ref_runtime()
    if what in ("cfg2", "all"):
        ref_lstm_full("cfg2", 8, 500, 128, 768, 2, 384, seed_x=31)
    if what in ("cfg4", "all"):
        ref_lstm_full("cfg4", 8, 440, 128, 1024, 2, 384, seed_x=32)
    if what in ("retrieval", "all"):
        ref_retrieval()
    if what in ("retrieval_cfg4", "all"):
        ref_retrieval(n_gallery=1536, n_query=384, tag="cfg4", T=440, H=1024, seed=104)
