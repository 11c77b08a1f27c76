"""CPU: the oracle restatements -- and the product's host-side (torch) classes -- against fixtures obtained by
EXECUTING the reference's own definitions (tests/golden/make_ref_goldens.py, tests/golden/ref_lift.py: the class /
function nodes are compiled straight out of the files under /root/reference in the build container; only arrays
were committed).  This is what pins the oracle to the reference itself rather than to the third-party calls."""
import types

import numpy as np
import pytest
import torch

from oracle import eeg_filter, losses, lstm


# ---- encoders ------------------------------------------------------------------------------------------------------
def test_oracle_lstm_matches_reference_lstmmodel_last_step(golden):
    """LSTMDistillRetreival.py:85-110 incl. its .view(B, C, T) reshape + Train.py:36-43 loss, f64, every gradient."""
    g = golden("ref_lstm.npz")
    B, TS, CH, H, L, D = g["a_dims"]
    p = lstm.init_params(TS, H, L, D, None, seed=int(g["a_seed_params"]))
    feat, loss, grads = lstm.reference_lstm_model(g["a_x"], p, L, target=g["a_target"])
    np.testing.assert_allclose(feat, g["a_feat_f64"], atol=1e-12)
    np.testing.assert_allclose(loss, g["a_loss_f64"], atol=1e-13)
    for k, v in grads.items():
        np.testing.assert_allclose(v, g[f"a_grad__{k}"], atol=1e-12, err_msg=k)
    # the reference's own f32 run agrees with the f64 oracle to f32 rounding (the bar the GPU f32 path is held to)
    np.testing.assert_allclose(feat, g["a_feat_f32"], atol=2e-6)
    assert abs(loss - g["a_loss_f32"]) < 1e-6
    # and the reshape is NOT a transpose
    wrong = lstm.model_forward(np.transpose(g["a_x"], (0, 2, 1)), p, L)
    assert np.abs(wrong - g["a_feat_f64"]).max() > 1e-3


def test_oracle_lstm_matches_reference_lstmmodel_all_steps(golden):
    """LSTMDistill.py:112-142: fc on every step, class_pred on the un-rectified output, ReLU on the features."""
    g = golden("ref_lstm.npz")
    B, TS, CH, H, L, D, NC = g["b_dims"]
    p = lstm.init_params(TS, H, L, D, NC, seed=int(g["b_seed_params"]))
    feat, cls, grads = lstm.reference_lstm_model_all_steps(g["b_x"], p, L, dfeat=g["b_wf"], dcls=g["b_wc"])
    np.testing.assert_allclose(feat, g["b_feat"], atol=1e-12)
    np.testing.assert_allclose(cls, g["b_cls"], atol=1e-12)
    assert feat.min() == 0.0
    for k, v in grads.items():
        np.testing.assert_allclose(v, g[f"b_grad__{k}"], atol=1e-10, err_msg=k)


def _check_full(g, tol_feat=1e-9):
    B, T, C, H, L, D = g["dims"]
    p = lstm.init_params(C, H, L, D, None, seed=int(g["seed_params"]))
    rng = np.random.default_rng(int(g["seed_x"]))
    x = rng.standard_normal((B, T, C)).astype(np.float32)
    tgt = rng.standard_normal((B, D)).astype(np.float32)
    return p, x, tgt


@pytest.mark.parametrize("tag", ["cfg2", "cfg4"])
def test_oracle_full_size_forward_matches_reference(golden, tag):
    """Full-size (cfg2: T 500, H 768; cfg4: T 440, H 1024) forward + loss of the oracle on 2 of the 8 fixture
    segments against the reference's f64 run (the backward at this size is exercised by the GPU tests)."""
    g = golden(f"ref_lstm_{tag}.npz")
    p, x, tgt = _check_full(g)
    L = int(g["dims"][4])
    feat = lstm.model_forward(x[:2], p, L)
    np.testing.assert_allclose(feat, g["feat_f64"][:2], atol=1e-10)
    np.testing.assert_allclose(g["feat_f32"], g["feat_f64"], atol=5e-6)
    assert abs(float(g["loss_f32"]) - float(g["loss_f64"])) < 1e-6


# ---- losses --------------------------------------------------------------------------------------------------------
def test_oracle_losses_match_reference_classes(golden):
    g = golden("ref_losses.npz")
    s, t, cls, tcls, lab = g["student"], g["teacher"], g["cls"], g["tcls"], g["labels"]
    np.testing.assert_allclose(losses.cosine_similarity_loss(s, t), g["cosine_loss"], atol=1e-14)
    np.testing.assert_allclose(losses.cosine_similarity_loss_grad(s, t), g["cosine_grad"], atol=1e-14)
    sched = losses.teacher_temp_schedule(100)
    np.testing.assert_array_equal(sched, g["temp_schedule_100"])
    for ep in (0, 25, 50, 99):
        np.testing.assert_allclose(losses.feature_distribution_loss(s, t, sched[ep], lab, cls), g[f"featdist_ep{ep}"], rtol=1e-12)
    for alpha, temp in ((1.0, 2.0), (0.5, 4.0), (0.9, 20.0)):
        np.testing.assert_allclose(losses.loss_fn_kd(cls, lab, tcls, alpha, temp), g[f"kd_a{alpha}_T{temp}"], rtol=1e-12)
    sw, cw, wt, tt, we = g["spamp_weights"]
    sched = losses.teacher_temp_schedule(100, wt, tt, int(we))
    for ep in (0, 25, 50):
        np.testing.assert_allclose(losses.feature_distribution_loss_kd(cls, tcls, sched[ep], lab, sw, cw),
                                   g[f"featdist_spamp_ep{ep}"], rtol=1e-12)
    wt, tt, we = g["eval_temps"]
    sched = losses.teacher_temp_schedule(100, wt, tt, int(we))
    for ep in (0, 50):
        np.testing.assert_allclose(losses.feature_distribution_loss_soft(s, t, sched[ep]), g[f"featdist_eval_ep{ep}"], rtol=1e-12)
    np.testing.assert_allclose(losses.feature_distribution_loss_mse(s, t), g["featdist_mse"], rtol=1e-12)
    # Barlow: the reference's forward with identity backbones
    loss, c = losses.barlow_loss(g["barlow_z1"], g["barlow_z2"], 32)
    np.testing.assert_allclose(loss, g["barlow_loss"], rtol=1e-12)
    _, _, grads = losses.barlow_loss_sharded(g["barlow_z1"], g["barlow_z2"], 1)
    np.testing.assert_allclose(grads[0][0], g["barlow_g1"], atol=1e-13)
    np.testing.assert_allclose(grads[0][1], g["barlow_g2"], atol=1e-13)
    cm = g["offdiag_in"]
    np.testing.assert_allclose(losses.off_diagonal_sqsum(cm), (g["offdiag_out"] ** 2).sum(), rtol=1e-13)
    # LR schedule table of adjust_learning_rate (weights 0.2 / biases 0.0048)
    tab = np.array([[losses.barlow_lr(st, 20, 5, 512) * 0.2, losses.barlow_lr(st, 20, 5, 512) * 0.0048] for st in range(100)])
    np.testing.assert_allclose(tab, g["barlow_lr_table"], rtol=1e-14, atol=0)


def test_oracle_barlow_two_ranks_matches_reference(golden):
    """net.py:33-42 executed on two gloo ranks: per-rank BatchNorm, c all-reduced in place, per-rank gradients."""
    g = golden("ref_barlow_2rank.npz")
    loss, c, grads = losses.barlow_loss_sharded(g["z1"], g["z2"], 2)
    for r in range(2):
        np.testing.assert_allclose(loss, g[f"loss_r{r}"], rtol=1e-12)
        np.testing.assert_allclose(grads[r][0], g[f"g1_r{r}"], atol=1e-13)
        np.testing.assert_allclose(grads[r][1], g[f"g2_r{r}"], atol=1e-13)


def test_oracle_dino_pieces_match_reference(golden):
    g = golden("ref_losses.npz")
    center = np.zeros((1, 32))
    sched = np.concatenate((np.linspace(0.04, 0.07, 3), np.ones(7) * 0.07))
    for step in range(2):
        loss, center = losses.dino_loss(g["dino_student"][step], g["dino_teacher"][step], center, sched[step + 1])
        np.testing.assert_allclose(loss, g[f"dino_loss{step}"], rtol=1e-12)
        np.testing.assert_allclose(center, g[f"dino_center{step}"], atol=1e-14)
    assert center.shape == (1, 5, 32)            # the per-sample centre quirk
    sd = {k[len("dinohead_sd__"):]: g[k] for k in g.files if k.startswith("dinohead_sd__")}
    np.testing.assert_allclose(losses.dino_head(g["dinohead_x"], sd), g["dinohead_y"], atol=1e-13)
    np.testing.assert_allclose(losses.cosine_scheduler(0.0005, 1e-6, 10, 7, warmup_epochs=2), g["cosine_scheduler"], atol=1e-18)
    np.testing.assert_allclose(losses.cosine_scheduler(0.996, 1, 5, 11), g["cosine_scheduler_momentum"], atol=1e-15)


# ---- preprocessing / dataset items -------------------------------------------------------------------------------
def test_oracle_preprocessing_matches_reference_methods(golden):
    g = golden("ref_preproc.npz")
    e = g["norm_in"]
    # normlizeEEG on a numpy array: ndarray.std() -> ddof 0; on a tensor: Tensor.std() -> ddof 1
    np.testing.assert_allclose(eeg_filter.zscore_rows(e.T, ddof=0).T, g["norm_np"], atol=1e-6)
    np.testing.assert_allclose(eeg_filter.zscore_rows(e.T, ddof=1).T, g["norm_torch"], atol=1e-6)
    # Utilities.remove_noise (filtfilt); the (b, a) form of this band is ill-conditioned: see test_oracle.py
    np.testing.assert_allclose(eeg_filter.remove_noise(g["filtfilt_x"], 1000), g["filtfilt_y"], atol=2e-4)
    raw = g["item_raw"]
    plain = np.stack([eeg_filter.dataset_item(r, 10, 70) for r in raw])
    np.testing.assert_array_equal(plain, g["item_plain"])
    np.testing.assert_array_equal(plain, g["item_plain_spamp"])
    sub = np.stack([eeg_filter.dataset_item(r, 10, 70, filter_channels=[7, 2, 9], channel_wise_norm=True) for r in raw])
    assert sub.shape == (6, 3, 60)               # channel-first: the reference's final .t()
    np.testing.assert_allclose(sub, g["item_subset_norm"], atol=1e-6)
    mean, std = g["item_scalar_mean_std"]
    sc = np.stack([eeg_filter.dataset_item(r, 10, 70, mean=mean, std=std) for r in raw])
    np.testing.assert_allclose(sc, g["item_scalar_norm"], atol=1e-6)


@pytest.mark.parametrize("tag,f32", [("f64", False), ("f32", True)])
def test_classwise_norm_matches_reference_side_effects(golden, tag, f32):
    """transformEEGDataToChannelWiseNorm executed on float64- and float32-stored records: oracle and product."""
    from cerebralsignalnetworks_amd.dataset import EEGDataset
    g = golden("ref_preproc.npz")
    lo, hi = (int(v) for v in g["cwn_window"])
    raw, lab = g["cwn_raw"], g["cwn_labels"]
    want = g[f"cwn_after_{tag}"][:, :, lo:hi]
    got = eeg_filter.classwise_channel_norm(raw[:, :, lo:hi], lab, time_low=lo, compat_stale_index=True, stored_float32=f32)
    np.testing.assert_array_equal(got, want)
    ds = EEGDataset(synthetic=len(lab), synthetic_channels=raw.shape[1], synthetic_samples=raw.shape[2], time_low=lo,
                    time_high=hi, device=torch.device("cpu"))
    ds.eeg_all = torch.from_numpy(raw[:, :, lo:hi].copy())
    ds.labels = torch.from_numpy(lab)
    ds.labels_dev = ds.labels
    ds.transformEEGDataToChannelWiseNorm(compat_stale_index=True, stored_float32=f32)
    np.testing.assert_allclose(ds.eeg_all.numpy(), want, atol=1e-6)


def test_product_dataset_items_match_reference_getitem(golden, tmp_path):
    """The on-disk path of the product dataset (ConvertToPth layout) against __getitem__ as the reference runs it:
    plain window, channel subset + per-channel z-score (channel-first result), scalar dataset-level normalisation."""
    from cerebralsignalnetworks_amd.dataset import EEGDataset
    g = golden("ref_preproc.npz")
    raw = g["item_raw"]
    N = raw.shape[0]
    blob = {"dataset": [{"eeg": torch.from_numpy(raw[i]), "image": 0, "label": 0, "subject": 1} for i in range(N)],
            "labels": ["n01"], "images": ["n01_7"]}
    path = tmp_path / "eeg.pth"
    torch.save(blob, path)
    kw = dict(eeg_signals_path=str(path), imagesRoot=str(tmp_path), time_low=10, time_high=70, device=torch.device("cpu"))
    ds = EEGDataset(**kw)
    np.testing.assert_array_equal(np.stack([ds[i][0].numpy() for i in range(N)]), g["item_plain"])
    ds = EEGDataset(filter_channels=[7, 2, 9], **kw)
    ds.apply_channel_wise_norm = True             # per-item z-score only (the init-time class-wise pass is tested above)
    np.testing.assert_allclose(np.stack([ds[i][0].numpy() for i in range(N)]), g["item_subset_norm"], atol=1e-6)
    ds = EEGDataset(apply_norm_with_stds_and_means=True, **kw)
    np.testing.assert_allclose([float(ds.mean), float(ds.std)], g["item_scalar_mean_std"], rtol=1e-6)
    np.testing.assert_allclose(np.stack([ds[i][0].numpy() for i in range(N)]), g["item_scalar_norm"], atol=1e-6)


# ---- the product's host-side classes (torch ops, run anywhere) -----------------------------------------------------
def test_product_loss_classes_match_reference(golden):
    from cerebralsignalnetworks_amd import losses as pl
    g = golden("ref_losses.npz")
    T64 = lambda a: torch.from_numpy(a).double()                          # noqa: E731
    G64 = lambda a: torch.from_numpy(a).double().requires_grad_(True)     # noqa: E731
    lab = torch.from_numpy(g["labels"])
    hp = pl.HyperParams
    fd = pl.FeatureDistributionLoss(100, hp.warmup_teacher_temp, hp.teacher_temp, hp.warmup_teacher_temp_epochs)
    for ep in (0, 25, 50, 99):
        s, c = G64(g["student"]), G64(g["cls"])
        loss = fd(s, T64(g["teacher"]), ep, lab, pred_label=c)
        loss.backward()
        np.testing.assert_allclose(loss.item(), g[f"featdist_ep{ep}"], rtol=1e-12)
        np.testing.assert_allclose(s.grad.numpy(), g[f"featdist_ep{ep}_gs"], atol=1e-14)
        np.testing.assert_allclose(c.grad.numpy(), g[f"featdist_ep{ep}_gc"], atol=1e-14)
    for alpha, temp in ((1.0, 2.0), (0.5, 4.0), (0.9, 20.0)):
        c = G64(g["cls"])
        loss = pl.loss_fn_kd(c, lab, T64(g["tcls"]), types.SimpleNamespace(alpha=alpha, temperature=temp))
        loss.backward()
        np.testing.assert_allclose(loss.item(), g[f"kd_a{alpha}_T{temp}"], rtol=1e-12)
        np.testing.assert_allclose(c.grad.numpy(), g[f"kd_a{alpha}_T{temp}_g"], atol=1e-14)
    fk = pl.FeatureDistributionLossKD(100, **pl.FeatureDistributionLossKD.SCHEDULE)
    for ep in (0, 25, 50):
        c = G64(g["cls"])
        loss = fk(c, T64(g["tcls"]), ep, lab)
        loss.backward()
        np.testing.assert_allclose(loss.item(), g[f"featdist_spamp_ep{ep}"], rtol=1e-12)
        np.testing.assert_allclose(c.grad.numpy(), g[f"featdist_spamp_ep{ep}_g"], atol=1e-14)
    fs = pl.FeatureDistributionLossSoft(100, **pl.FeatureDistributionLossSoft.SCHEDULE)
    for ep in (0, 50):
        s = G64(g["student"])
        loss = fs(s, T64(g["teacher"]), ep)
        loss.backward()
        np.testing.assert_allclose(loss.item(), g[f"featdist_eval_ep{ep}"], rtol=1e-12)
        np.testing.assert_allclose(s.grad.numpy(), g[f"featdist_eval_ep{ep}_g"], atol=1e-14)
    s = G64(g["student"])
    loss = pl.FeatureDistributionLossMSE()(s, T64(g["teacher"]))
    loss.backward()
    np.testing.assert_allclose(loss.item(), g["featdist_mse"], rtol=1e-12)
    np.testing.assert_allclose(s.grad.numpy(), g["featdist_mse_g"], atol=1e-14)
    # Barlow on CPU tensors (torch reduction; the device path through the HIP kernel is a -m gpu test)
    a, b = G64(g["barlow_z1"]), G64(g["barlow_z2"])
    crit = pl.BarlowTwinsLoss(96, 32).double().train()
    loss = crit(a, b)
    loss.backward()
    np.testing.assert_allclose(loss.item(), g["barlow_loss"], rtol=1e-12)
    np.testing.assert_allclose(a.grad.numpy(), g["barlow_g1"], atol=1e-13)
    np.testing.assert_allclose(b.grad.numpy(), g["barlow_g2"], atol=1e-13)
    tab = np.array([[pl.barlow_learning_rate(st, 20, 5, 512) * 0.2, pl.barlow_learning_rate(st, 20, 5, 512) * 0.0048]
                    for st in range(100)])
    np.testing.assert_allclose(tab, g["barlow_lr_table"], rtol=1e-14, atol=0)


def test_product_dino_pieces_match_reference(golden):
    from cerebralsignalnetworks_amd.dino import DINOHead, DINOLoss, cosine_scheduler
    g = golden("ref_losses.npz")
    crit = DINOLoss(32, 6, 0.04, 0.07, 3, 10).double()
    for step in range(2):
        s = torch.from_numpy(g["dino_student"][step]).requires_grad_(True)
        loss = crit(s, torch.from_numpy(g["dino_teacher"][step]), step + 1)
        loss.backward()
        np.testing.assert_allclose(loss.item(), g[f"dino_loss{step}"], rtol=1e-12)
        np.testing.assert_allclose(s.grad.numpy(), g[f"dino_grad{step}"], atol=1e-14)
        np.testing.assert_allclose(crit.center.numpy(), g[f"dino_center{step}"], atol=1e-14)
    head = DINOHead(16, 12, nlayers=3, hidden_dim=32, bottleneck_dim=8).double()
    sd = {k[len("dinohead_sd__"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("dinohead_sd__")}
    head.load_state_dict(sd)                      # same module / parameter names as the reference's head
    np.testing.assert_allclose(head(torch.from_numpy(g["dinohead_x"])).detach().numpy(), g["dinohead_y"], atol=1e-13)
    np.testing.assert_allclose(cosine_scheduler(0.0005, 1e-6, 10, 7, warmup_epochs=2), g["cosine_scheduler"], atol=1e-18)
    np.testing.assert_allclose(cosine_scheduler(0.996, 1, 5, 11), g["cosine_scheduler_momentum"], atol=1e-15)
    # the paper's pairing (compat=False) against its definition, row-concatenated views
    rng = np.random.default_rng(1)
    so, to = rng.standard_normal((6 * 4, 16)), rng.standard_normal((2 * 4, 16))
    crit = DINOLoss(16, 6, 0.04, 0.07, 3, 10, compat=False).double()
    got = crit(torch.from_numpy(so), torch.from_numpy(to), 0).item()
    q = losses._softmax(to / 0.04).reshape(2, 4, 16)
    lp = losses._log_softmax(so / 0.1).reshape(6, 4, 16)
    want = np.mean([(-(q[i] * lp[v]).sum(-1)).mean() for i in range(2) for v in range(6) if v != i])
    assert abs(got - want) < 1e-12


def test_split_follows_random_split():
    """LstmDistillFromDinoV2Train.py:289-290: random_split(dataset, [0.8, 0.2], manual_seed(43)) -- floor of each
    fraction, remainder to the first split(s); N = 9, 14, 19 are sizes where round(0.8 N) gets it wrong."""
    from torch.utils.data import random_split
    from cerebralsignalnetworks_amd.trainer import split_indices
    for n, n_train in ((9, 8), (14, 12), (19, 16), (256, 205), (11965, 9572)):
        tr, va = split_indices(n, (0.8, 0.2), seed=43)
        ref_tr, ref_va = random_split(range(n), [0.8, 0.2], generator=torch.Generator().manual_seed(43))
        assert tr.tolist() == list(ref_tr.indices) and va.tolist() == list(ref_va.indices)
        assert len(tr) == n_train and len(tr) + len(va) == n and not set(tr.tolist()) & set(va.tolist())
        perm = torch.randperm(n, generator=torch.Generator().manual_seed(43))
        assert tr.tolist() == perm[:n_train].tolist() and va.tolist() == perm[n_train:].tolist()
