"""CPU: the oracle restatements against the committed golden vectors (scipy / torch /
importable reference pieces, see tests/golden/make_goldens.py)."""
import numpy as np

from oracle import eeg_filter, losses, lstm, retrieval


def test_filter_design_matches_reference_band_and_scipy(golden):
    g = golden("filter_design.npz")
    for fs in (1000, 2048):
        np.testing.assert_allclose(g[f"band_hz_fs{fs}"], [eeg_filter.LOW_CUTOFF_HZ, eeg_filter.HIGH_CUTOFF_HZ])
        np.testing.assert_allclose(g[f"band_norm_fs{fs}"], [0.1 / (fs / 2), 60.0 / (fs / 2)], rtol=1e-15)
        for order in eeg_filter.ORDERS:
            sos = eeg_filter.design_bandpass_sos(fs, order)
            assert sos.shape == (order, 6)
            np.testing.assert_allclose(sos, g[f"sos_fs{fs}_o{order}"], rtol=1e-12, atol=0)


def test_sosfilt_and_znorm_match_scipy(golden):
    g = golden("filter_apply.npz")
    x = g["x"]
    for order in (3, 4, 5):
        sos = eeg_filter.design_bandpass_sos(1000, order)
        y = eeg_filter.sosfilt_rows(sos, x)
        np.testing.assert_allclose(y, g[f"sosfilt_o{order}"], rtol=0, atol=1e-11)
        for ddof in (0, 1):
            z = eeg_filter.zscore_rows(y, ddof)
            np.testing.assert_allclose(z, g[f"znorm_o{order}_ddof{ddof}"], rtol=0, atol=1e-9)
    out = eeg_filter.eeg_bandpass_znorm(x, eeg_filter.design_bandpass_sos(1000, 3), ddof=0)
    assert out.shape == (2, 500, 16)
    np.testing.assert_allclose(out, np.transpose(g["znorm_o3_ddof0"], (0, 2, 1)), atol=1e-9)
    out_tm = eeg_filter.eeg_bandpass_znorm(x, eeg_filter.design_bandpass_sos(1000, 3), ddof=0, time_major=True)
    np.testing.assert_array_equal(out_tm, np.transpose(out, (1, 0, 2)))


def test_filtfilt_matches_scipy(golden):
    g = golden("filter_apply.npz")
    y = eeg_filter.remove_noise(g["filtfilt_x"], 1000)
    # the (b,a) form of this order-8 band-pass is ill-conditioned (max|pole| 0.9979): two
    # float64 evaluations that differ only in rounding order already disagree at ~5e-5.
    np.testing.assert_allclose(y, g["filtfilt_y"], rtol=0, atol=2e-4)


def test_synthetic_recipe():
    x = eeg_filter.synthetic_eeg(3)
    assert x.shape == (3, 128, 500) and x.dtype == np.float32
    # 40 Hz component of amplitude 0.5 is present in the channel mean
    m = x.mean(axis=(0, 1))
    t = np.arange(500) / 1000.0
    amp = 2 * np.abs((m * np.exp(-2j * np.pi * 40 * t)).mean())
    assert abs(amp - 0.5) < 0.05


def _params(g):
    return {k[len("param__"):]: g[k] for k in g.files if k.startswith("param__")}


def test_lstm_forward_backward_match_torch(golden):
    g = golden("lstm_small.npz")
    B, T, C, H, L, D, NC = g["dims"]
    p = _params(g)
    (feat, cls), saved = lstm.model_forward(g["x"], p, L, include_top=True, return_saved=True)
    np.testing.assert_allclose(feat, g["feat_f64"], atol=1e-12)
    np.testing.assert_allclose(cls, g["cls_f64"], atol=1e-12)
    np.testing.assert_allclose(saved["y"], g["yall_f64"], atol=1e-12)
    loss = losses.cosine_similarity_loss(feat, g["target"])
    np.testing.assert_allclose(loss, g["loss_f64"], atol=1e-13)
    dfeat = losses.cosine_similarity_loss_grad(feat, g["target"])
    _, grads = lstm.model_backward(dfeat, p, saved, L)
    for k, v in grads.items():
        key = f"grad_f64__{k}"
        if key in g.files:
            np.testing.assert_allclose(v, g[key], atol=1e-12, err_msg=k)
    # fp32 torch agrees with the f64 oracle to fp32 rounding: the tolerance the GPU f32 path is held to
    np.testing.assert_allclose(feat, g["feat_f32"], atol=2e-6)
    assert abs(loss - g["loss_f32"]) < 1e-6


def test_lstm_full_size_forward_matches_torch(golden):
    g = golden("lstm_full_fwd.npz")
    B, T, C, H, L, D = g["dims"]
    p = lstm.init_params(C, H, L, D, None, seed=int(g["seed_params"]))
    x = np.random.default_rng(int(g["seed_x"])).standard_normal((B, T, C)).astype(np.float32)
    feat, saved = lstm.model_forward(x, p, L, return_saved=True)
    np.testing.assert_allclose(saved["last"], g["ylast"], atol=5e-6)
    np.testing.assert_allclose(feat, g["feat"], atol=5e-6)


def test_losses_match_torch(golden):
    g = golden("losses.npz")
    s, t, cls, lab = g["student"], g["teacher"], g["cls"], g["labels"]
    np.testing.assert_allclose(losses.cosine_similarity_loss(s, t), g["cosine_loss"], atol=1e-14)
    np.testing.assert_allclose(losses.cosine_similarity_loss_grad(s, t), g["cosine_grad"], atol=1e-14)
    sched = losses.teacher_temp_schedule(100)
    np.testing.assert_allclose(sched, g["temp_schedule_100"], atol=0)
    for ep in (0, 25, 50):
        v = losses.feature_distribution_loss(s, t, sched[ep], lab, cls)
        np.testing.assert_allclose(v, g[f"featdist_ep{ep}"], rtol=1e-12)
    for alpha, temp in ((1.0, 2.0), (0.5, 4.0)):
        v = losses.loss_fn_kd(cls, lab, cls[::-1], alpha, temp)
        np.testing.assert_allclose(v, g[f"kd_a{alpha}_T{temp}"], rtol=1e-12)
    loss, c = losses.barlow_loss(g["barlow_z1"], g["barlow_z2"], 32)
    np.testing.assert_allclose(c, g["barlow_c"], atol=1e-12)
    np.testing.assert_allclose(losses.off_diagonal_sqsum(c), g["barlow_off"], rtol=1e-12)
    np.testing.assert_allclose(loss, g["barlow_loss"], rtol=1e-12)


def test_lars_matches_reference_optimizer(golden):
    g = golden("losses.npz")
    w, b = g["lars_w0"], g["lars_b0"]
    mw, mb = np.zeros_like(w), np.zeros_like(b)
    for it in range(2):
        w, mw = losses.lars_step(w, g["lars_gw"], mw, 0.2, 1e-3, weight_decay_filter=True, lars_adaptation_filter=True)
        b, mb = losses.lars_step(b, g["lars_gb"], mb, 0.2, 1e-3, weight_decay_filter=True, lars_adaptation_filter=True)
        np.testing.assert_allclose(w, g[f"lars_w{it + 1}"], atol=1e-14)
        np.testing.assert_allclose(b, g[f"lars_b{it + 1}"], atol=1e-14)


def test_barlow_lr_schedule_shape():
    lrs = [losses.barlow_lr(s, 20, 5, 512) for s in range(100)]
    assert lrs[0] == 0.0 and abs(lrs[50] - 2.0) < 1e-12          # end of 10-epoch warm-up = batch/256
    assert all(a >= b for a, b in zip(lrs[50:], lrs[51:]))        # cosine decay afterwards
    assert lrs[-1] > 2.0 * 0.001


def test_retrieval_known_answers():
    gal = np.array([[0, 0], [1, 0], [0, 2], [3, 3], [1, 0]], np.float32)   # 1 and 4 tie
    qry = np.array([[0.9, 0.1], [0, 1.9]], np.float32)
    D, I = retrieval.l2_topk(gal, qry, 3)
    assert I.tolist() == [[1, 4, 0], [2, 0, 1]]
    np.testing.assert_allclose(D[0], [0.02, 0.02, 0.82], atol=1e-6)
    names = {0: "cat", 1: "dog"}
    glab = [dict(ClassId=c, ClassName=names[c]) for c in (0, 1, 0, 1, 1)]
    qlab = [dict(ClassId=1, ClassName="dog"), dict(ClassId=1, ClassName="dog")]
    rec, prec, per, top1 = retrieval.evaluate_from_indices(I, glab, qlab, names, 3)
    # query0 top3 = dog,dog,cat -> hit, 2 instances; query1 top3 = cat,cat,dog -> hit, 1 instance
    assert per["dog"]["TP"] == 2 and per["dog"]["classIntanceRetrival"] == 3
    assert rec == 100.0 and prec == round(300 / 6, 2) and top1 == 0.5


def test_product_lars_and_lr_schedule_match_oracle(golden):
    """Host-side mirrors (pure torch, no kernels): LARS and the Barlow LR schedule."""
    import torch
    from cerebralsignalnetworks_amd.losses import LARS, barlow_learning_rate
    g = golden("losses.npz")
    w = torch.from_numpy(g["lars_w0"].copy()).requires_grad_(True)
    b = torch.from_numpy(g["lars_b0"].copy()).requires_grad_(True)
    opt = LARS([w, b], lr=0.2, weight_decay=1e-3, weight_decay_filter=True, lars_adaptation_filter=True)
    for it in range(2):
        w.grad, b.grad = torch.from_numpy(g["lars_gw"]).clone(), torch.from_numpy(g["lars_gb"]).clone()
        opt.step()
        np.testing.assert_allclose(w.detach().numpy(), g[f"lars_w{it + 1}"], atol=1e-14)
        np.testing.assert_allclose(b.detach().numpy(), g[f"lars_b{it + 1}"], atol=1e-14)
    for s in (0, 7, 50, 73, 99):
        assert barlow_learning_rate(s, 20, 5, 512) == losses.barlow_lr(s, 20, 5, 512)


def test_dino_loss_and_schedulers_cpu(golden):
    """DINO pieces are pure torch (run anywhere): loss value against a numpy restatement of
    LstmDistillation.py:118-159 incl. its chunk quirk, cosine_scheduler against the reference's own output."""
    import torch
    from cerebralsignalnetworks_amd.dino import DINOLoss, cosine_scheduler, temporal_crops
    g = golden("losses.npz")
    if "cosine_scheduler" in g.files:
        np.testing.assert_allclose(cosine_scheduler(0.0005, 1e-6, 10, 7, warmup_epochs=2), g["cosine_scheduler"], atol=0)
    rng = np.random.default_rng(0)
    so = rng.standard_normal((6, 5, 32))
    to = rng.standard_normal((2, 5, 32))
    crit = DINOLoss(32, 6, 0.04, 0.07, 3, 10)
    loss = crit(torch.from_numpy(so), torch.from_numpy(to), 1).item()
    temp = np.linspace(0.04, 0.07, 3)[1]
    q = losses._softmax(to / temp)                                  # centre is zero at the first step
    want = np.mean([(-(q * losses._log_softmax(so[v:v + 1] / 0.1)).sum(-1)).mean() for v in range(1, 6)])
    assert abs(loss - want) < 1e-12
    assert tuple(crit.center.shape) == (1, 5, 32)                   # the reference's per-sample centre quirk
    np.testing.assert_allclose(crit.center.numpy(), 0.1 * to.sum(0, keepdims=True) / 2, atol=1e-12)
    eeg = torch.arange(2 * 495 * 3, dtype=torch.float32).reshape(2, 495, 3)
    gv, lv = temporal_crops(eeg, rng=np.random.RandomState(1))
    assert [v.shape[1] for v in gv] == [300, 300] and [v.shape[1] for v in lv] == [200] * 4
