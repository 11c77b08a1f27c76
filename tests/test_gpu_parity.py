"""GPU parity tests: every HIP entry point, called through the C ABI, against the CPU oracle on
the same seeded inputs and against the committed golden vectors.  Run with ``-m gpu``.

Tolerances (written here, per the north star): float32 path -- distill loss within 1e-4 of the
oracle, gradients within 1e-4 relative to their scale; retrieval indices bit-exact.  The bf16
MFMA path is characterised separately with looser bounds (bf16 operands carry 8 significant bits).
"""
import os

import numpy as np
import pytest
import torch

from cerebralsignalnetworks_amd import cabi, Model, LSTMModel, CosineSimilarityLoss, EEGFilters, BarlowTwinsLoss
from cerebralsignalnetworks_amd import retrieval as hip_retrieval
from oracle import eeg_filter, losses, lstm, retrieval

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def dev_t(a, cuda, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
    return t.to(dtype) if dtype is not None else t


def bf16_round(a):
    return torch.from_numpy(np.asarray(a, np.float32)).to(torch.bfloat16).float().numpy()


# ----------------------------------------------------------------------------------------------
# K1 + K2
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("order", [3, 4, 5])
@pytest.mark.parametrize("ddof", [0, 1])
def test_filter_znorm_matches_golden(cuda, golden, order, ddof):
    g = golden("filter_apply.npz")
    x = g["x"]                                                     # [2,16,500]
    sos = eeg_filter.design_bandpass_sos(1000, order)
    y = cabi.eeg_bandpass_znorm(dev_t(x, cuda), sos, ddof=ddof).cpu().numpy()      # [B,T,C]
    want = np.transpose(g[f"znorm_o{order}_ddof{ddof}"], (0, 2, 1))
    assert y.shape == want.shape
    # float64 state on the device: only the final float32 rounding separates us from scipy
    np.testing.assert_allclose(y, want, rtol=0, atol=2e-6)


@pytest.mark.parametrize("B,C,T", [(3, 128, 500), (2, 96, 460), (1, 5, 461), (2, 128, 440), (1, 1, 2),
                                   # scan kernel edges: last tile / last wave partly empty (60 rows), T = 512 (no leading zeros),
                                   # T = 36 (14 of 16 chunk lanes all zeros), one float4 per row, 33 rows (second tile = 1 row)
                                   (3, 20, 512), (5, 12, 36), (2, 8, 8), (1, 132, 260),
                                   # row-walking fallback: T > 512, T % 4 != 0 with C % 4 == 0
                                   (2, 8, 600), (2, 8, 130)])
def test_filter_shapes_layouts_dtypes(cuda, B, C, T):
    x = eeg_filter.synthetic_eeg(B, C, T, seed=B * 1000 + T)
    sos = eeg_filter.design_bandpass_sos(1000, 3)
    want = eeg_filter.eeg_bandpass_znorm(x, sos, ddof=0)            # [B,T,C] f64
    xt = dev_t(x, cuda)
    y = cabi.eeg_bandpass_znorm(xt, sos).cpu().numpy()
    np.testing.assert_allclose(y, want, atol=5e-6)
    y_tm = cabi.eeg_bandpass_znorm(xt, sos, time_major=True).cpu().numpy()
    np.testing.assert_array_equal(y_tm, np.transpose(y, (1, 0, 2)))
    y_bf = cabi.eeg_bandpass_znorm(xt, sos, out_dtype=torch.bfloat16).float().cpu().numpy()
    np.testing.assert_array_equal(y_bf, bf16_round(y))
    # no filter sections: pure per-channel z-score (normlizeEEG)
    z = cabi.eeg_bandpass_znorm(xt, None, ddof=1).cpu().numpy()
    np.testing.assert_allclose(z, np.transpose(eeg_filter.zscore_rows(x, 1), (0, 2, 1)), atol=5e-6)


def test_filter_is_linear_and_idempotent_in_znorm_at_full_size(cuda):
    """Size-independent properties at the benchmark shape (B=256 is too slow for the python oracle):
    z-scored output has mean 0 / std 1 per row, and scaling the input leaves it unchanged."""
    x = torch.from_numpy(eeg_filter.synthetic_eeg(256, 128, 500, seed=5)).to(cuda)
    sos = eeg_filter.design_bandpass_sos(1000, 3)
    y = cabi.eeg_bandpass_znorm(x, sos)                            # [B,T,C]
    m = y.double().mean(dim=1)
    s = y.double().std(dim=1, unbiased=False)
    assert m.abs().max().item() < 1e-6 and (s - 1).abs().max().item() < 1e-6
    y2 = cabi.eeg_bandpass_znorm(x * 3.5 , sos)
    assert (y - y2).abs().max().item() < 1e-5
    # spot-check 4 segments against the oracle
    want = eeg_filter.eeg_bandpass_znorm(x[:4].cpu().numpy(), sos)
    np.testing.assert_allclose(y[:4].cpu().numpy(), want, atol=5e-6)


# ----------------------------------------------------------------------------------------------
# GEMM building blocks
# ----------------------------------------------------------------------------------------------
def test_mfma_layout_identity_times_asymmetric(cuda):
    """A = I with an asymmetric B catches a transposed C write or a swapped fragment map."""
    n = 128
    a = np.eye(n, dtype=np.float32)
    b = (np.arange(n)[:, None] * 3 + np.arange(n)[None, :] * 0.25).astype(np.float32) / 64.0   # Bt[N,K]
    for dt in (torch.float32, torch.bfloat16):
        c = cabi.gemm_nt(dev_t(a, cuda, dt), dev_t(b, cuda, dt)).cpu().numpy()
        want = a @ (bf16_round(b) if dt == torch.bfloat16 else b).T
        np.testing.assert_allclose(c, want, atol=1e-6)
        c2 = cabi.gemm_tn(dev_t(a, cuda, dt), dev_t(b, cuda, dt)).cpu().numpy()      # A^T B = B
        np.testing.assert_allclose(c2, bf16_round(b) if dt == torch.bfloat16 else b, atol=1e-6)


@pytest.mark.parametrize("M,N,K", [(256, 384, 128), (300, 260, 72), (128, 128, 64), (64, 3072, 768), (70, 52, 40),
                                   (1024, 3072, 768), (520, 1028, 320), (520, 1536, 320),    # 256 x 128 LDS-DMA ring kernel
                                   (512, 768, 3072)])                       # 128 x 128 LDS-DMA kernel
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
def test_gemm_nt(cuda, M, N, K, dt):
    rng = np.random.default_rng(M + N + K)
    a = rng.standard_normal((M, K)).astype(np.float32)
    b = rng.standard_normal((N, K)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    if dt == torch.bfloat16:
        a, b = bf16_round(a), bf16_round(b)
    want = a.astype(np.float64) @ b.astype(np.float64).T + bias
    c = cabi.gemm_nt(dev_t(a, cuda, dt), dev_t(b, cuda, dt), dev_t(bias, cuda)).cpu().numpy()
    np.testing.assert_allclose(c, want, atol=2e-4 * np.sqrt(K))
    cb = cabi.gemm_nt(dev_t(a, cuda, dt), dev_t(b, cuda, dt), dev_t(bias, cuda), out_dtype=torch.bfloat16)
    np.testing.assert_allclose(cb.float().cpu().numpy(), want, rtol=1e-2, atol=1e-1)
    acc = dev_t(np.ones((M, N), np.float32), cuda)
    cabi.gemm_nt(dev_t(a, cuda, dt), dev_t(b, cuda, dt), None, out=acc, accumulate=True)
    np.testing.assert_allclose(acc.cpu().numpy(), want - bias + 1.0, atol=2e-4 * np.sqrt(K))


@pytest.mark.parametrize("M,N,K", [(256, 128, 512), (384, 96, 1000), (3072, 768, 2048), (72, 40, 333)])
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
def test_gemm_tn(cuda, M, N, K, dt):
    rng = np.random.default_rng(M * 7 + N + K)
    a = rng.standard_normal((K, M)).astype(np.float32)
    b = rng.standard_normal((K, N)).astype(np.float32)
    if dt == torch.bfloat16:
        a, b = bf16_round(a), bf16_round(b)
    want = a.astype(np.float64).T @ b.astype(np.float64)
    c = cabi.gemm_tn(dev_t(a, cuda, dt), dev_t(b, cuda, dt)).cpu().numpy()
    np.testing.assert_allclose(c, want, atol=2e-4 * np.sqrt(K))
    if dt == torch.bfloat16:
        # the transposed-LDS-read kernel and the scalar-read kernel must agree bit for bit
        os.environ["CSN_TN_NO_TR"] = "1"
        try:
            c2 = cabi.gemm_tn(dev_t(a, cuda, dt), dev_t(b, cuda, dt)).cpu().numpy()
        finally:
            del os.environ["CSN_TN_NO_TR"]
        np.testing.assert_array_equal(c, c2)
        c_run2 = cabi.gemm_tn(dev_t(a, cuda, dt), dev_t(b, cuda, dt)).cpu().numpy()
        np.testing.assert_array_equal(c, c_run2)          # split-K combine is order-fixed


@pytest.mark.parametrize("M,N,K,stages", [(512, 256, 8192, "5"), (3072, 768, 16000, "5"), (264, 520, 8256, "4"),
                                            (768, 256, 12800, "3"), (3072, 768, 32000, "4"), (512, 768, 8192 + 64, "4"),
                                            (3072, 768, 16000, "4n")])
def test_gemm_tn_256_tile_pipeline(cuda, M, N, K, stages):
    """The 256 x 256 LDS-DMA weight-gradient kernel against float64 and against the register-staged 128 x 128
    kernel: "4" = the default, two wave groups one barrier interval apart on a ring of 5 stages; "3" / "5" / "4n"
    (CSN_TN_NO_STAGGER) = the single-phase rings."""
    if stages.endswith("n"):
        stages = stages[:-1]
        os.environ["CSN_TN_NO_STAGGER"] = "1"
    rng = np.random.default_rng(M + N + K)
    a = bf16_round(rng.standard_normal((K, M)).astype(np.float32))
    b = bf16_round(rng.standard_normal((K, N)).astype(np.float32))
    want = a.astype(np.float64).T @ b.astype(np.float64)
    os.environ["CSN_TN_STAGES"] = stages
    try:
        c = cabi.gemm_tn(dev_t(a, cuda, torch.bfloat16), dev_t(b, cuda, torch.bfloat16)).cpu().numpy()
        c2 = cabi.gemm_tn(dev_t(a, cuda, torch.bfloat16), dev_t(b, cuda, torch.bfloat16)).cpu().numpy()
    finally:
        del os.environ["CSN_TN_STAGES"]
        os.environ.pop("CSN_TN_NO_STAGGER", None)
    np.testing.assert_allclose(c, want, atol=2e-4 * np.sqrt(K))
    np.testing.assert_array_equal(c, c2)
    os.environ["CSN_GEMM_NO_256"] = "1"
    try:
        c3 = cabi.gemm_tn(dev_t(a, cuda, torch.bfloat16), dev_t(b, cuda, torch.bfloat16)).cpu().numpy()
    finally:
        del os.environ["CSN_GEMM_NO_256"]
    np.testing.assert_allclose(c, c3, atol=1e-5 * np.sqrt(K))     # different K split: same products, other sum order


def test_gemm_tn_staggered_ring_race_screen(cuda):
    """The default weight-gradient kernel orders its LDS-DMA stages by counted waits and barriers only (two wave groups
    one barrier interval apart, ring of 5): a misplaced wait shows up as RARE wrong tiles that come and go with the
    memory load.  Screen: the same products, summed in the same order, as the single-phase ring -- bit for bit, 25
    times per shape, with other work (a second stream hammering HBM) running beside it."""
    rng = np.random.default_rng(11)
    side = torch.cuda.Stream()
    junk = torch.empty(64 << 20, dtype=torch.float32, device=cuda)
    for (M, N, K) in ((3072, 768, 32768), (512, 256, 8192 + 192), (768, 768, 20480)):
        a = dev_t(bf16_round(rng.standard_normal((K, M)).astype(np.float32)), cuda, torch.bfloat16)
        b = dev_t(bf16_round(rng.standard_normal((K, N)).astype(np.float32)), cuda, torch.bfloat16)
        os.environ["CSN_TN_NO_STAGGER"] = "1"
        try:
            want = cabi.gemm_tn(a, b)
        finally:
            del os.environ["CSN_TN_NO_STAGGER"]
        for rep in range(25):
            with torch.cuda.stream(side):
                junk.add_(1.0)
            got = cabi.gemm_tn(a, b)
            assert torch.equal(got, want), (M, N, K, rep, float((got - want).abs().max()))
        torch.cuda.synchronize()


# ----------------------------------------------------------------------------------------------
# K3: cell steps, full sequence
# ----------------------------------------------------------------------------------------------
def _sig(x):
    return 1 / (1 + np.exp(-x))


@pytest.mark.parametrize("B,H", [(4, 64), (70, 96), (256, 768),
                                 # K-split cell kernels (H % 128 == 0 in f32, % 256 in bf16) with ragged row tiles
                                 (70, 256), (33, 1024), (130, 512)])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_cell_forward_backward_step(cuda, B, H, dt):
    rng = np.random.default_rng(B + H)
    k = 1 / np.sqrt(H)
    w = rng.uniform(-k, k, (4 * H, H)).astype(np.float32)
    h0 = rng.uniform(-1, 1, (B, H)).astype(np.float32)
    c0 = rng.standard_normal((B, H)).astype(np.float32)
    xp = rng.standard_normal((B, 4 * H)).astype(np.float32)
    if dt == torch.bfloat16:
        w, h0 = bf16_round(w), bf16_round(h0)
    a = xp.astype(np.float64) + h0.astype(np.float64) @ w.astype(np.float64).T
    i, f, g, o = _sig(a[:, :H]), _sig(a[:, H:2 * H]), np.tanh(a[:, 2 * H:3 * H]), _sig(a[:, 3 * H:])
    c1 = f * c0 + i * g
    h1 = o * np.tanh(c1)
    hd, cd, gd = cabi.lstm_cell_forward(dev_t(h0, cuda, dt), dev_t(w, cuda, dt), dev_t(xp, cuda), dev_t(c0, cuda))
    tol = 2e-6 if dt == torch.float32 else 1e-2
    np.testing.assert_allclose(cd.cpu().numpy(), c1, atol=2e-6 if dt == torch.float32 else 2e-5)
    np.testing.assert_allclose(hd.float().cpu().numpy(), h1, atol=tol)
    np.testing.assert_allclose(gd.float().cpu().numpy(), np.concatenate([i, f, g, o], 1), atol=tol)
    # zero initial state: null h_prev / c_prev
    hz, cz, _ = cabi.lstm_cell_forward(None, dev_t(w, cuda, dt), dev_t(xp, cuda), None)
    a0 = xp.astype(np.float64)
    cz_ref = _sig(a0[:, :H]) * np.tanh(a0[:, 2 * H:3 * H])
    np.testing.assert_allclose(cz.cpu().numpy(), cz_ref, atol=2e-6)

    # backward step against the oracle formulas
    dgn = (rng.standard_normal((B, 4 * H)) * 0.1).astype(np.float32)
    dy = rng.standard_normal((B, H)).astype(np.float32)
    dcn = rng.standard_normal((B, H)).astype(np.float32)
    gates = np.concatenate([i, f, g, o], 1).astype(np.float32)
    if dt == torch.bfloat16:
        dgn, gates = bf16_round(dgn), bf16_round(gates)
    gi, gf, gg, go = (gates[:, j * H:(j + 1) * H].astype(np.float64) for j in range(4))
    c1f = c1.astype(np.float32).astype(np.float64)
    dh = dy + dgn.astype(np.float64) @ w.astype(np.float64)
    tc = np.tanh(c1f)
    dc = dh * go * (1 - tc * tc) + dcn
    want = np.concatenate([dc * gg * gi * (1 - gi), dc * c0 * gf * (1 - gf), dc * gi * (1 - gg * gg),
                           dh * tc * go * (1 - go)], 1)
    wt = np.ascontiguousarray(w.T)
    dcar = dev_t(dcn, cuda)
    out = cabi.lstm_cell_backward(dev_t(dgn, cuda, dt), dev_t(wt, cuda, dt), dev_t(dy, cuda), dev_t(gates, cuda, dt),
                                  dev_t(c1.astype(np.float32), cuda), dev_t(c0, cuda), dcar)
    np.testing.assert_allclose(out.float().cpu().numpy(), want, atol=1e-5 if dt == torch.float32 else 2e-2)
    np.testing.assert_allclose(dcar.cpu().numpy(), dc * gf, atol=1e-5)


def _model_from_params(p, C, H, L, D, NC, dtype, cuda):
    m = Model(input_size=C, lstm_size=H, lstm_layers=L, output_size=D, include_top=bool(NC), n_classes=NC or 40,
              compute_dtype=dtype)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in p.items()})
    return m.to(cuda)


def test_lstm_model_matches_torch_golden_f32(cuda, golden):
    """Full path on the small golden: outputs, loss (<= 1e-4) and every gradient vs torch.nn.LSTM."""
    g = golden("lstm_small.npz")
    B, T, C, H, L, D, NC = (int(v) for v in g["dims"])
    p = {k[len("param__"):]: g[k] for k in g.files if k.startswith("param__")}
    m = _model_from_params(p, C, H, L, D, NC, torch.float32, cuda)
    feat, cls = m(dev_t(g["x"], cuda))
    loss = CosineSimilarityLoss()(feat, dev_t(g["target"], cuda))
    loss.backward()
    np.testing.assert_allclose(feat.detach().cpu().numpy(), g["feat_f64"], atol=5e-6)
    np.testing.assert_allclose(cls.detach().cpu().numpy(), g["cls_f64"], atol=5e-6)
    assert abs(loss.item() - float(g["loss_f64"])) < 1e-4
    assert abs(loss.item() - float(g["loss_f32"])) < 1e-5
    for name, par in m.named_parameters():
        key = f"grad_f64__{name}"
        if key in g.files:
            want = g[key]
            scale = max(1e-3, np.abs(want).max())
            np.testing.assert_allclose(par.grad.cpu().numpy(), want, atol=1e-4 * scale, err_msg=name)


def test_lstm_all_steps_and_input_grad_f32(cuda):
    rng = np.random.default_rng(0)
    B, T, C, H, L = 3, 20, 24, 32, 3
    p = lstm.init_params(C, H, L, 8, None, seed=9)
    lp = {k[len("lstm."):]: v for k, v in p.items() if k.startswith("lstm.")}
    x = rng.standard_normal((B, T, C)).astype(np.float32)
    dy = rng.standard_normal((B, T, H)).astype(np.float32)
    y, saved = lstm.lstm_forward(x, lp, L, return_saved=True)
    dx_ref, g_ref = lstm.lstm_backward(dy, lp, saved, L)
    m = _model_from_params(p, C, H, L, 8, None, torch.float32, cuda)
    xt = dev_t(x, cuda).requires_grad_(True)
    y_all, y_last = m.lstm(xt, want_all=True)
    np.testing.assert_allclose(y_all.detach().cpu().numpy(), y, atol=5e-6)
    np.testing.assert_allclose(y_last.detach().cpu().numpy(), y[:, -1], atol=5e-6)
    (y_all * dev_t(dy, cuda)).sum().backward()
    np.testing.assert_allclose(xt.grad.cpu().numpy(), dx_ref, atol=2e-5)
    for k, v in g_ref.items():
        got = getattr(m.lstm, k).grad.cpu().numpy()
        np.testing.assert_allclose(got, v, atol=1e-4 * max(1.0, np.abs(v).max()), err_msg=k)


def test_lstm_full_size_forward_f32_and_bf16(cuda, golden):
    g = golden("lstm_full_fwd.npz")
    B, T, C, H, L, D = (int(v) for v in g["dims"])
    p = lstm.init_params(C, H, L, D, None, seed=int(g["seed_params"]))
    x = np.random.default_rng(int(g["seed_x"])).standard_normal((B, T, C)).astype(np.float32)
    with torch.no_grad():
        m = _model_from_params(p, C, H, L, D, None, torch.float32, cuda)
        feat = m(dev_t(x, cuda)).cpu().numpy()
        np.testing.assert_allclose(feat, g["feat"], atol=2e-5)
        mb = _model_from_params(p, C, H, L, D, None, torch.bfloat16, cuda)
        featb = mb(dev_t(x, cuda)).cpu().numpy()
    # bf16 operands over 500 recurrent steps x 2 layers: characterised, not bit-tight
    err = np.abs(featb - g["feat"]).max()
    assert err < 3e-2, err
    cos = (featb * g["feat"]).sum(1) / np.linalg.norm(featb, axis=1) / np.linalg.norm(g["feat"], axis=1)
    assert cos.min() > 0.999


def test_bf16_training_step_tracks_f32(cuda):
    rng = np.random.default_rng(4)
    B, T, C, H, L, D = 16, 48, 32, 64, 2, 32
    p = lstm.init_params(C, H, L, D, None, seed=3)
    x = rng.standard_normal((B, T, C)).astype(np.float32)
    tgt = rng.standard_normal((B, D)).astype(np.float32)
    res = {}
    for dt in (torch.float32, torch.bfloat16):
        m = _model_from_params(p, C, H, L, D, None, dt, cuda)
        loss = CosineSimilarityLoss()(m(dev_t(x, cuda)), dev_t(tgt, cuda))
        loss.backward()
        res[dt] = (loss.item(), {n: q.grad.cpu().numpy() for n, q in m.named_parameters()})
    feat, saved = lstm.model_forward(x, p, L, return_saved=True)
    assert abs(res[torch.float32][0] - losses.cosine_similarity_loss(feat, tgt)) < 1e-5
    assert abs(res[torch.bfloat16][0] - res[torch.float32][0]) < 5e-3
    for n, gf in res[torch.float32][1].items():
        gb = res[torch.bfloat16][1][n]
        rel = np.linalg.norm(gb - gf) / max(1e-12, np.linalg.norm(gf))
        assert rel < 5e-2, (n, rel)


def test_reference_view_quirk_lstmmodel(cuda):
    """LSTMDistillRetreival.LSTMModel: x.view(B, C, T) is a reshape, sequence runs over channels."""
    rng = np.random.default_rng(1)
    B, T, C, H = 2, 32, 6, 32
    m = LSTMModel(input_size=T, hidden_size=H, n_layers=2, out_features=8, compute_dtype=torch.float32).to(cuda)
    x = rng.standard_normal((B, T, C)).astype(np.float32)
    p = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    want = lstm.model_forward(x.reshape(B, C, T), p, 2)
    with torch.no_grad():
        got = m(dev_t(x, cuda)).cpu().numpy()
    np.testing.assert_allclose(got, want, atol=5e-6)


# ----------------------------------------------------------------------------------------------
# K5 / K7 / K8
# ----------------------------------------------------------------------------------------------
def test_cosine_loss_and_grad(cuda, golden):
    g = golden("losses.npz")
    s = dev_t(g["student"], cuda).requires_grad_(True)
    loss = CosineSimilarityLoss()(s, dev_t(g["teacher"], cuda))
    loss.backward()
    assert abs(loss.item() - float(g["cosine_loss"])) < 1e-6
    np.testing.assert_allclose(s.grad.cpu().numpy(), g["cosine_grad"], atol=1e-8)


def test_barlow_reduction(cuda, golden):
    g = golden("losses.npz")
    out = cabi.barlow_offdiag_sqsum(dev_t(g["barlow_c"].astype(np.float32), cuda)).cpu().numpy()
    np.testing.assert_allclose(out, [float(g["barlow_on"]), float(g["barlow_off"])], rtol=1e-5)
    z1 = dev_t(g["barlow_z1"].astype(np.float32), cuda).requires_grad_(True)
    z2 = dev_t(g["barlow_z2"].astype(np.float32), cuda)
    crit = BarlowTwinsLoss(96, 32).to(cuda)
    loss = crit(z1, z2)
    loss.backward()
    assert abs(loss.item() - float(g["barlow_loss"])) < 1e-3 * float(g["barlow_loss"])
    # gradient against torch autograd of the reference formula
    z1r = dev_t(g["barlow_z1"].astype(np.float32), cuda).requires_grad_(True)
    bn = torch.nn.BatchNorm1d(96, affine=False).to(cuda)
    c = bn(z1r).T @ bn(z2) / 32
    ref = torch.diagonal(c).add(-1).pow(2).sum() + 0.0051 * (c.pow(2).sum() - torch.diagonal(c).pow(2).sum())
    ref.backward()
    np.testing.assert_allclose(z1.grad.cpu().numpy(), z1r.grad.cpu().numpy(), atol=1e-5)


@pytest.mark.parametrize("Ng,Nq,D,k", [(200, 50, 384, 5), (1000, 257, 384, 5), (37, 3, 20, 37), (5, 2, 2, 3)])
def test_l2_topk_indices_bit_exact(cuda, Ng, Nq, D, k):
    rng = np.random.default_rng(Ng + Nq)
    gal = rng.standard_normal((Ng, D)).astype(np.float32)
    qry = rng.standard_normal((Nq, D)).astype(np.float32)
    gal[Ng // 2] = gal[0]                       # an exact duplicate: tie must go to the lower index
    qry[0] = gal[0]
    Dref, Iref = retrieval.l2_topk(gal, qry, k)
    Dg, Ig = cabi.l2_topk(dev_t(gal, cuda), dev_t(qry, cuda), k)
    np.testing.assert_array_equal(Ig.cpu().numpy(), Iref)
    np.testing.assert_allclose(Dg.cpu().numpy(), Dref, rtol=1e-6, atol=1e-6)


def test_evaluate_matches_oracle_bookkeeping(cuda):
    rng = np.random.default_rng(8)
    names = {i: f"class{i % 37}" for i in range(40)}            # two ids share a name on purpose
    glab = [dict(ClassId=int(c), ClassName=names[int(c)]) for c in rng.integers(0, 40, 200)]
    qlab = [dict(ClassId=int(c), ClassName=names[int(c)]) for c in rng.integers(0, 40, 50)]
    gal = rng.standard_normal((200, 384)).astype(np.float32)
    qry = rng.standard_normal((50, 384)).astype(np.float32)

    class DS:
        class_id_to_str = names
        class_str_to_id = {v: k for k, v in names.items()}

    class FL:
        topK = 5
    r = hip_retrieval.evaluate_full(FL, list(gal), list(qry), glab, qlab, DS)
    rec, prec, per, top1, D, I = retrieval.evaluate(gal, qry, glab, qlab, names, 5)
    np.testing.assert_array_equal(r["I"], I)
    assert r["Recall_Total"] == rec and r["Precision_Total"] == prec and r["top1"] == top1
    for k2, v in per.items():
        for f in ("TP", "classIntanceRetrival", "TotalRetrival", "TotalClass", "Recall", "Precision"):
            assert r["class_scores"][k2][f] == v[f]


# ----------------------------------------------------------------------------------------------
# end to end: filter -> LSTM -> loss -> backward, one training step, vs the oracle pipeline
# ----------------------------------------------------------------------------------------------
def test_end_to_end_step_loss_within_1e4(cuda):
    from cerebralsignalnetworks_amd.trainer import DistillTrainer
    B, C, T, H, L, D = 8, 128, 500, 96, 2, 384
    x = eeg_filter.synthetic_eeg(B, C, T, seed=43)
    tgt = np.random.default_rng(44).standard_normal((B, D)).astype(np.float32)
    p = lstm.init_params(C, H, L, D, None, seed=43)
    m = _model_from_params(p, C, H, L, D, None, torch.float32, cuda)
    filt = EEGFilters(1000, order=3)
    tr = DistillTrainer(m, filt.sos, loss="cosine", lr=1e-3, optimizer="rmsprop")
    loss = tr.train_step(dev_t(x, cuda), dev_t(tgt, cuda))
    eeg_ref = eeg_filter.eeg_bandpass_znorm(x, filt.sos)
    feat, saved = lstm.model_forward(eeg_ref, p, L, return_saved=True)
    loss_ref = losses.cosine_similarity_loss(feat, tgt)
    assert abs(loss.item() - loss_ref) < 1e-4, (loss.item(), loss_ref)
    # the RMSprop update moved the weights in the direction the oracle gradient predicts
    _, grads = lstm.model_backward(losses.cosine_similarity_loss_grad(feat, tgt), p, saved, L)
    w_new = m.lstm.weight_hh_l1.detach().cpu().numpy()
    step = w_new - p["lstm.weight_hh_l1"]
    gref = grads["lstm.weight_hh_l1"]
    big = np.abs(gref) > 1e-3 * np.abs(gref).max()
    assert (np.sign(step[big]) == -np.sign(gref[big])).mean() > 0.999


# ----------------------------------------------------------------------------------------------
# bf16 fast path ("il": fragment-major, gate-interleaved, layer wavefront, side-stream GEMMs)
# ----------------------------------------------------------------------------------------------
def _run_lstm(p, x, dy_all, dy_last, C, H, L, dtype, cuda, env=None, want_dx=True, info=None):
    env = env or {}
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        m = _model_from_params(p, C, H, L, 8, None, dtype, cuda)
        xt = dev_t(x, cuda).requires_grad_(want_dx)
        y_all, y_last = m.lstm(xt, want_all=True)
        loss = (y_all * dev_t(dy_all, cuda)).sum() + (y_last * dev_t(dy_last, cuda)).sum()
        loss.backward()
        torch.cuda.synchronize()
        for plan in m.lstm.all_plans():
            assert plan.status() == 0, "an in-kernel hand-off of the weight-stationary forward timed out"
        out = dict(y_all=y_all.detach().cpu().numpy())
        if want_dx:
            out["dx"] = xt.grad.cpu().numpy()
        if info is not None:
            info["dgates_copies"] = [plan.dgates_copies() for plan in m.lstm.all_plans()]
        for n, q in m.lstm.named_parameters():
            out[n] = q.grad.cpu().numpy()
        return out
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("B,T,C,H,L", [(1, 1, 8, 128, 1), (2, 2, 8, 128, 2), (3, 3, 16, 256, 2), (1, 5, 128, 768, 2),
                                       (65, 7, 128, 768, 2), (64, 1, 128, 768, 2), (256, 2, 128, 768, 2),
                                       (7, 3, 128, 1024, 2), (64, 4, 32, 512, 3), (129, 5, 16, 384, 4),
                                       (32, 20, 128, 768, 2), (16, 33, 16, 128, 2), (48, 31, 32, 256, 4)])
def test_short_sequences_and_tiny_batches(cuda, B, T, C, H, L):
    """Edge shapes at the DEFAULT chunk length (32): sequences shorter than one chunk, shorter than the hand-off ring
    (T < 4), one row, one step, a second chunk of one step.  With T <= chunk and two or more layers the backward's
    chunk diagonals (layers two launches apart) include one WITHOUT any layer in range, which must still run the
    input-gradient GEMM the launch before left for it (a status-1 error before round 3's last fix)."""
    rng = np.random.default_rng(B * T + H)
    p = lstm.init_params(C, H, L, 8, None, seed=5)
    lp = {k[len("lstm."):]: v for k, v in p.items() if k.startswith("lstm.")}
    x = rng.standard_normal((B, T, C)).astype(np.float32)
    dy_all = (rng.standard_normal((B, T, H)) * 0.1).astype(np.float32)
    dy_last = rng.standard_normal((B, H)).astype(np.float32)
    y, saved = lstm.lstm_forward(x, lp, L, return_saved=True)
    dy = dy_all.astype(np.float64).copy()
    dy[:, -1] += dy_last
    dx_ref, g_ref = lstm.lstm_backward(dy, lp, saved, L)
    for dt, tol in ((torch.bfloat16, 1.5e-2), (torch.float32, 1e-5)):
        out = _run_lstm(p, x, dy_all, dy_last, C, H, L, dt, cuda)
        assert all(np.isfinite(v).all() for v in out.values())
        assert _rel(out["y_all"], y) < tol and _rel(out["dx"], dx_ref) < tol, (dt, _rel(out["y_all"], y), _rel(out["dx"], dx_ref))
        for n, gref in g_ref.items():
            assert _rel(out[n], gref) < tol, (dt, n, _rel(out[n], gref))


def _assert_same_bits(a, b, what):
    """Exact equality with a diagnosis of WHERE the arrays differ (a stale hand-off shows up as a block of rows /
    units at one timestep, a summation-order difference as scattered single ulps)."""
    if np.array_equal(a, b):
        return
    bad = np.argwhere(a != b)
    axes = [sorted(set(bad[:, i].tolist()))[:12] for i in range(bad.shape[1])]
    raise AssertionError(f"{what}: {len(bad)} of {a.size} elements differ, max |diff| {np.abs(a - b).max():.3g}; "
                         f"indices per axis (first 12): {axes}")


def _rel(a, b):
    return np.linalg.norm(np.asarray(a, np.float64) - b) / max(1e-12, np.linalg.norm(b))


@pytest.mark.parametrize("B,T,C,H,L,chunk", [(70, 75, 24, 128, 2, "32"), (33, 40, 16, 128, 3, "8"),
                                             (64, 20, 32, 256, 1, "32"), (5, 9, 8, 384, 2, "4"),
                                             (6, 44, 128, 1024, 2, "16"),      # cfg4 width (Spampinato split: H=1024)
                                             (40, 24, 128, 128, 4, "8"),       # 4 layers (TrainSpampinato.py:368)
                                             (130, 33, 16, 512, 2, "32"),
                                             # the benchmark width: kernels <6,6> forward / <2,24> backward / K2xN2 body of
                                             # lstm_fwd_ns.hip, one M-tile and all four (8 hand-off groups = 8 XCDs)
                                             (64, 40, 128, 768, 2, "8"), (256, 24, 128, 768, 2, "8"),
                                             # T*B >= 8192 rows: 256 x 256 weight-gradient kernel + fused bias sums
                                             (64, 130, 32, 256, 2, "32")])
def test_fast_path_matches_oracle_and_v1(cuda, B, T, C, H, L, chunk):
    rng = np.random.default_rng(B * T + H)
    p = lstm.init_params(C, H, L, 8, None, seed=5)
    lp = {k[len("lstm."):]: v for k, v in p.items() if k.startswith("lstm.")}
    x = rng.standard_normal((B, T, C)).astype(np.float32)
    dy_all = (rng.standard_normal((B, T, H)) * 0.1).astype(np.float32)
    dy_last = rng.standard_normal((B, H)).astype(np.float32)
    y, saved = lstm.lstm_forward(x, lp, L, return_saved=True)
    dy = dy_all.astype(np.float64).copy()
    dy[:, -1] += dy_last
    dx_ref, g_ref = lstm.lstm_backward(dy, lp, saved, L)

    run = lambda **env: _run_lstm(p, x, dy_all, dy_last, C, H, L, torch.bfloat16, cuda,
                                  {"CSN_LSTM_CHUNK": chunk, **env})
    fast = run()
    slow = _run_lstm(p, x, dy_all, dy_last, C, H, L, torch.bfloat16, cuda, {"CSN_CELL_V1": "1"})
    # the schedules / hand-off forms of the default path compute the same bits: single stream instead of side
    # streams, the placement-independent hand-off instead of the L2-local one
    # (nobeside: the input-gradient GEMMs as kernels of their own between two backward launches, layers one chunk
    # apart, instead of on the idle workgroups of the next launch with the layers two chunks apart)
    # (flags: the hand-off of round 1 -- drained stores, one flag word per workgroup and step -- instead of the data
    # polls on a ring of 4 sentinel-armed slabs; nohint: the data polls without hint words in the backward too (the
    # forward's default), so that EVERY step loads before its operands are there and goes through the re-read / redo
    # paths; fwd hint: the forward with hint words)
    for name, other in (("serial", run(CSN_NO_SIDE_STREAM="1")), ("anyplace", run(CSN_NO_XCD_LOCAL="1")),
                        ("nobeside", run(CSN_NO_BESIDE="1")),
                        ("flags", run(CSN_FWD_FLAGS="1", CSN_BWD_FLAGS="1")), ("nohint", run(CSN_DPOLL_NO_HINT="1")),
                        ("fwd hint", run(CSN_FWD_HINT="1")),
                        ("nohint anyplace", run(CSN_DPOLL_NO_HINT="1", CSN_NO_XCD_LOCAL="1"))):
        for k in fast:
            _assert_same_bits(fast[k], other[k], f"{name}: {k}")
    # with every workgroup walking K in the same order (the default rotates the walk per workgroup, which only
    # reorders the f32 summation), the weight-stationary kernels compute exactly the bits of the per-diagonal
    # launches, forward and backward
    # (and with layer 0's input projection as a separate GEMM instead of fused into its kernel, which moves the
    # x W_ih^T products into the recurrent sum)
    # (the K-split weight-stationary forward: the default N-split kernel keeps each complete K sum in one
    # accumulator instead of four K-quarter partials reduced through LDS, a different -- equally valid -- order)
    exact = dict(CSN_NO_ROTATE="1", CSN_NO_FUSE_X="1", CSN_FWD_KSPLIT="1")
    norot = run(**exact)
    # the N-split forward kernel (lstm_fwd_ns.hip; the default at H = 1024, behind CSN_FWD_NSPLIT at H <= 512): one launch
    # per layer on its own stream gives the bits of the grouped launch, and it agrees with the K-split kernel to bf16
    # rounding.  (Its losing variants -- K2 x N2 body at H = 768, half-pipelined, GEMM carried inside, wave-specialised
    # forward -- live in `make experiments` and tests/test_gpu_experiments.py, outside the product library.)
    if H != 768:
        ns = run(CSN_FWD_NSPLIT="1")
        _assert_same_bits(ns["y_all"], run(CSN_FWD_NSPLIT="1", CSN_PERSIST_STREAMS="1")["y_all"], "ns streams: y_all")
        for k in ns:
            assert _rel(ns[k], fast[k]) < 1e-2, (k, _rel(ns[k], fast[k]))
    # (streams: one forward launch per layer on its own stream instead of the grouped launch, per-diagonal backward)
    for name, other in (("diag", run(CSN_NO_PERSIST="1", **exact)),
                        ("diag_bwd", run(CSN_NO_PERSIST_BWD="1", **exact)),
                        ("streams", run(CSN_PERSIST_STREAMS="1", **exact))):
        for k in norot:
            _assert_same_bits(norot[k], other[k], f"{name}: {k}")
    for k in fast:
        assert _rel(fast[k], norot[k]) < 1e-2, (k, _rel(fast[k], norot[k]))
    assert np.abs(fast["y_all"] - y).max() < 3e-2
    assert _rel(fast["dx"], dx_ref) < 4e-2
    for k, v in g_ref.items():
        assert _rel(fast[k], v) < 4e-2, (k, _rel(fast[k], v))
        # same bf16 arithmetic, different layouts / activations: much closer to each other than to f64
        assert _rel(fast[k], slow[k]) < 2e-2, (k, _rel(fast[k], slow[k]))
    assert np.abs(fast["y_all"] - slow["y_all"]).max() < 2e-2


def test_cli_spampinato_trainer_surface(cuda, tmp_path):
    """LstmDistillFromDinoV2TrainSpampinato.py (BASELINE.json configs[3]): AdamW 1e-4, Model(128, 128, layers=4,
    include_top=False), loss_fn_kd with alpha / T from --hyperprams, the two checkpoint names, weights-only resume when
    the file exists (reference :368-378, :467-475), 128 x 440 synthetic segments."""
    import LstmDistillFromDinoV2TrainSpampinato as spamp
    import LstmDistillFromDinoV2Train as loop
    fl = loop.SPAMPINATO
    d = loop.build_parser(fl).parse_args([])
    assert (d.learning_rate, d.num_epochs, d.hidden_size, d.lstm_layers, d.loss, d.optimizer) == (1e-4, 200, 128, 4, "kd", "adamw")
    assert "'alpha': 0" in d.hyperprams and d.eeg_dataset.endswith("spampinato/eeg_signals_raw_with_mean_std.pth")
    args = ["--synthetic", "128", "--batch_size", "16", "--num_epochs", "11", "--log_dir", str(tmp_path),
            "--hyperprams", "{'alpha': 0.3, 'temperature': 2}", "--learning_rate", "0.002"]
    hist = spamp.main(args)
    assert len(hist) == 11 and all(np.isfinite(hist)) and hist[-1] < hist[0]           # KD + CE on 40 classes goes down
    first = os.path.join(str(tmp_path), "lstm_dinov2_epoch_5_best_loss.pth")           # first validated best (epoch 5)
    assert os.path.exists(first)
    sd = torch.load(first, weights_only=True)
    assert sd["lstm.weight_hh_l3"].shape == (512, 128) and sd["fc.weight"].shape == (384, 128) and "class_pred.weight" not in sd
    # a later improvement goes to the second name (on random labels the validation loss need not improve at epoch 10)
    assert fl.checkpoint_path("d", 10, 11, False) == "d/lstm_dinov2_epochs_11_best_loss.pth"
    assert fl.checkpoint_path("d", 5, 11, True) == "d/lstm_dinov2_epoch_5_best_loss.pth"
    assert loop.PERILS.checkpoint_path("d", 5, 11, True) == loop.PERILS.checkpoint_path("d", 9, 11, False) == "d/lstm_dinov2_best_loss.pth"
    assert not os.path.exists(os.path.join(str(tmp_path), "lstm_dinov2_best_loss.pth"))
    later = first
    # resume: the file exists -> loaded (strict); the first epoch then starts from the trained loss, not from scratch
    hist2 = spamp.main(args[:-2] + ["--learning_rate", "1e-5", "--num_epochs", "1", "--custom_model_weights", later,
                                    "--log_dir", str(tmp_path / "resume")])
    assert hist2[0] < 0.5 * (hist[0] + hist[-1])
    # a path that does not exist is ignored, as in the reference (os.path.exists guard)
    hist3 = spamp.main(args[:-2] + ["--num_epochs", "1", "--custom_model_weights", str(tmp_path / "nope.pth"),
                                    "--log_dir", str(tmp_path / "fresh")])
    assert abs(hist3[0] - hist[0]) < 0.25 * hist[0]
    # first step of this configuration against the oracle's loss_fn_kd (same weights, same batch)
    from cerebralsignalnetworks_amd.dataset import EEGDataset
    from cerebralsignalnetworks_amd.trainer import DistillTrainer
    torch.manual_seed(43)
    ds = EEGDataset(synthetic=64, synthetic_samples=440, time_low=0, time_high=440, seed=43, device=cuda)
    m = Model(input_size=128, lstm_size=128, lstm_layers=4, output_size=384, include_top=False,
              compute_dtype=torch.float32).to(cuda)
    params = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    kd = loop._KdParams()
    kd.alpha, kd.temperature = 0.3, 2
    sos = EEGFilters(1000, order=3).sos
    tr = DistillTrainer(m, sos, loss="kd", lr=1e-4, optimizer="adamw", kd_params=kd)
    assert isinstance(tr.opt, torch.optim.AdamW)
    idx = torch.arange(16, device=cuda)
    loss = float(tr.train_step(ds.eeg_all[idx], ds.features_all[idx], ds.labels_dev[idx]))
    eeg = eeg_filter.eeg_bandpass_znorm(ds.eeg_all[idx].cpu().numpy(), sos)
    feat = lstm.model_forward(eeg, params, 4)
    want = losses.loss_fn_kd(feat, ds.labels_dev[idx].cpu().numpy(), ds.features_all[idx].cpu().numpy(), 0.3, 2)
    assert abs(loss - want) < 1e-4, (loss, want)


def test_handoff_forms_agree_in_poisoned_workspaces(cuda):
    """The variant-equality tests above re-run one problem in workspaces the caching allocator recycles, so a consumer
    that loads BEFORE its producer stored finds the previous run's value of the same element -- the right bits, by
    accident.  tools/handoff_poison.py fills the memory every run's workspace is carved from with bf16 NaNs first (data
    to the sentinel proof, poison to the arithmetic) and requires the default, flag, no-hint and placement-independent
    hand-offs to reproduce the clean default run bit for bit at the H = 512 (partial last M-tile), cfg2, H = 256 and cfg4
    widths.  (This is the test that exposed the compiler's broken spill code in the H = 512 forward: DESIGN.md 3.5.)"""
    import json
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "handoff_poison.py"), "2"],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    bad = {k: v for k, v in res.items() if v["differs"] or v["nonfinite"] or v["status"]}
    assert not bad, bad


@pytest.mark.parametrize("M,N,K", [(8192, 3072, 768), (300, 24960, 128), (513, 12480, 192), (4096, 3072, 64),
                                   (8192, 4096, 1024), (500, 32768, 128), (2048, 8192, 320)])      # (the last three: 256 x 256 tiles)
def test_gemm_nt_wide_tile_kernel(cuda, M, N, K):
    """csn_gemm_nt's 256 x 192 / 256 x 256 tile kernel (N % 192 or % 256 == 0, whole rounds of tiles over the CUs): same k order as the 256 x 128 kernel,
    so the float32 output must be BIT-equal to it (CSN_GEMM_NO_192 selects the old kernel); rows that do not fill the last
    tile, bias, accumulate and bfloat16 output against float64."""
    g = torch.Generator(device=cuda).manual_seed(M + N + K)
    a = torch.randn(M, K, device=cuda, generator=g).to(torch.bfloat16)
    b = torch.randn(N, K, device=cuda, generator=g).to(torch.bfloat16)
    bias = torch.randn(N, device=cuda, generator=g)
    ref = a.double() @ b.double().t() + bias.double()
    out = cabi.gemm_nt(a, b, bias)
    assert float((out.double() - ref).norm() / ref.norm()) < 2e-6
    os.environ["CSN_GEMM_NO_192"] = "1"
    try:
        old = cabi.gemm_nt(a, b, bias)
    finally:
        del os.environ["CSN_GEMM_NO_192"]
    assert torch.equal(out, old)
    o16 = cabi.gemm_nt(a, b, bias, out_dtype=torch.bfloat16)
    assert float((o16.double() - ref).norm() / ref.norm()) < 4e-3
    acc = out.clone()
    cabi.gemm_nt(a, b, None, out=acc, accumulate=True)
    assert float((acc.double() - (2 * ref - bias.double())).norm() / ref.norm()) < 4e-6


def test_entry_points_on_random_odd_shapes(cuda):
    """tools/fuzz_entry_points.py: every stateless entry point of the C ABI (band-pass + z-score, filtfilt, both GEMMs in
    both dtypes, cosine loss, RMSprop step, Barlow reduction, L2 top-k) on seeded random shapes that are NOT multiples of
    any tile -- one row, one channel, 139 channels, K = 1, prime sizes -- against numpy / scipy / torch in float64."""
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_entry_points.py"), "10"],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert out.stdout.count("\nok ") + out.stdout.startswith("ok ") >= 100, out.stdout[-2000:]


def test_bench_two_ranks_rehearsal_on_one_device(cuda):
    """`python bench.py --gpus 2` end to end -- launcher, rank processes, rank 0's fixture check before the timed region,
    all-reduced steps, the collective status verdict, the data-parallel proof in the line -- rehearsed on ONE device
    (CSN_SINGLE_DEVICE: both ranks on GPU 0, gloo instead of RCCL, per-step launches instead of the weight-stationary
    kernels, which need all of a GPU's CUs to themselves).  Every collective must be reached by every rank: a status
    check that all-reduced its verdict inside rank 0's solo fixture check broke exactly this run once."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, CSN_SINGLE_DEVICE="1", CSN_DIST_BACKEND="gloo", CSN_AR_OVERLAP="1")     # (the launcher adds CSN_NO_PERSIST itself)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "CSN_NO_PERSIST"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    dp = res["data_parallel"]
    assert res["n_gpus"] == 2 and dp["ranks_seen"] == 2 and dp["param_checksum_equal_on_all_ranks"] and dp["single_device_rehearsal"]
    assert res["parity"]["f32_within_1e-4"] and res["config"]["global_batch"] == 512
    # a rehearsal line carries no throughput claim; the gradient all-reduce ran bucketed and overlapped (two messages)
    assert res["value"] is None and res["rehearsal_value"] > 0
    assert dp["allreduce_overlapped"] and len(dp["allreduce_segments_bytes"]) == 3 and dp["allreduce_exposed_ms_per_step"] > 0
    assert sum(dp["allreduce_segments_bytes"]) == dp["allreduce_bytes"]


def test_train_cli_two_ranks_rehearsal_on_one_device(cuda, tmp_path):
    """The training CLI under `torch.distributed.run` with two ranks (rehearsal switches as in the bench test: one device,
    gloo, per-step launches): sharded batches, the flat-buffer gradient all-reduce every step, the collective status
    verdict per epoch, validation + rank-0 evaluation and checkpoint, barrier + teardown -- no rank may be left in a
    collective the other never reaches (exit code 0 within the timeout)."""
    import subprocess
    import sys
    env = dict(os.environ, CSN_SINGLE_DEVICE="1", CSN_DIST_BACKEND="gloo", CSN_NO_PERSIST="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29613", os.path.join(ROOT, "LstmDistillFromDinoV2Train.py"),
                          "--synthetic", "250", "--batch_size", "16", "--num_epochs", "3", "--validation_frequency", "1",
                          "--log_dir", str(tmp_path), "--hidden_size", "128", "--lstm_layers", "2", "--loss", "cosine"],
                         env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    assert "EPOCH 2 train_loss" in out.stdout and "val_loss" in out.stdout
    assert os.path.exists(os.path.join(str(tmp_path), "lstm_dinov2_best_loss.pth"))


def test_long_sequences_beyond_4gib_offsets(cuda):
    """tests/diag/long_sequence.py at T = 1100: at B = 256, H = 1024 the float32 input projection is 4 MB per step, and the
    N-split forward kernel addressed it with a 32-bit byte offset from step 0 -- garbage from step 1024 on (relative
    difference 1.0), silently.  Its buffer resources are based per launch now; the K-split kernels leave sequences whose
    slabs reach 4 GiB to the per-diagonal launches.  Outputs per step and the weight gradients against those launches."""
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "diag", "long_sequence.py"), "1100"],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]


def test_rccl_backend_one_rank(cuda):
    """What one GPU can show of the RCCL path: every collective the bench and the trainer issue (dtypes, reduce ops) on a
    one-rank "nccl" communicator, and six data-parallel training steps at the benchmark shape with the flat-buffer
    all-reduce on the device between the weight-stationary launches (tests/diag/rccl_ops_one_rank.py,
    nccl_trainer_one_rank.py; each in a process of its own)."""
    import subprocess
    import sys
    for script in ("rccl_ops_one_rank.py", "nccl_trainer_one_rank.py"):
        env = dict(os.environ)
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k, None)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "diag", script)], env=env,
                             capture_output=True, text=True, timeout=600)
        assert out.returncode == 0 and " ok" in out.stdout, (script, out.stdout[-1500:], out.stderr[-3000:])


def test_lstm_usage_patterns(cuda):
    """tests/diag/usage_patterns.py: inference plans (no_grad), a non-default stream, two forwards of different batch sizes
    awaiting one backward, repeated calls bit-equal, batches of 512 / 1024 rows (off the weight-stationary path) -- both
    dtypes, against the float64 oracle."""
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "diag", "usage_patterns.py")],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]


def test_lstm_plans_on_random_shapes(cuda):
    """tests/diag/fuzz_lstm.py: 40 seeded random plans -- both dtypes, 1-4 layers, batch / length / channel counts that are not
    multiples of any tile, hidden sizes on and off the weight-stationary list, chunk lengths 1..64 (T <= chunk and T >> chunk),
    with and without per-step output gradients and dx -- against the float64 oracle (bf16 2e-2, f32 2e-5 of the norm)."""
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "diag", "fuzz_lstm.py"), "40", "12"],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]


def test_ring_slots_never_serve_a_stale_step(cuda):
    """The hand-off by data reuses four slab addresses per layer, so a consumer must never be served the PREVIOUS occupant
    of a slot -- data, which the sentinel proof cannot tell from the right step.  The debug library `make tags`
    (-DCSN_SLAB_TAGS) tags every piece with bit 2 of its step and checks every piece consumed (tools/tags_check.py, run in
    a process of its own: one process loads one library): clean under load in the default, no-hint, placement-independent
    and per-layer-stream forms at the cfg2 / H = 512 / cfg4 widths, and -- fault injection -- the detector fires when the
    re-arm stores are compiled out of the protocol."""
    import json
    import subprocess
    import sys
    lib = os.path.join(ROOT, "cerebralsignalnetworks_amd", "lib", "libcsn_hip_tags.so")
    if not os.path.exists(lib):
        pytest.skip("debug library not built (make -C cerebralsignalnetworks_amd/csrc tags)")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "tags_check.py"), "4"],
                         env=dict(os.environ, CSN_LIB_PATH=lib), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    print("tags check:", res)
    assert all(v == 0 for v in res["clean"].values()), res
    assert all(v & 4 for v in res["injected"].values()), res         # the detector does fire when the protocol is broken


# ----------------------------------------------------------------------------------------------
# CLI drop-ins (BASELINE.json configs[0] shape: 256 synthetic segments, batch 16, 1 epoch)
# ----------------------------------------------------------------------------------------------
def test_cli_train_then_eval_cfg1(cuda, tmp_path):
    import LstmDistillFromDinoV2Train as train
    import LstmDistillFromDinoV2Eval as evaluate_cli
    args = ["--synthetic", "256", "--batch_size", "16", "--num_epochs", "1", "--log_dir", str(tmp_path),
            "--hidden_size", "128", "--lstm_layers", "2", "--loss", "cosine", "--dtype", "f32"]
    hist = train.main(args)
    assert len(hist) == 1 and np.isfinite(hist[0]) and 0.5 < hist[0] < 1.5      # cosine loss vs random targets ~ 1
    ckpt = os.path.join(str(tmp_path), "lstm_dinov2_best_loss.pth")
    sd = torch.load(ckpt, weights_only=True)
    assert "lstm.weight_hh_l1" in sd and "fc.weight" in sd
    r = evaluate_cli.main(args + ["--custom_model_weights", ckpt, "--topK", "5"])
    assert 0.0 <= r["Recall_Total"] <= 100.0 and r["I"].shape == (51, 5)
    for suffix in (".pth", ".txt", "_.csv"):
        assert os.path.exists(os.path.join(str(tmp_path), f"Theperils_sub_1_Scores{suffix}"))
    # first optimisation step of the same configuration against the oracle (distill loss within 1e-4)
    from cerebralsignalnetworks_amd.dataset import EEGDataset
    from cerebralsignalnetworks_amd.trainer import DistillTrainer, shard_indices
    torch.manual_seed(43)
    ds = EEGDataset(synthetic=256, time_low=0, time_high=500, seed=43, device=cuda)
    m = Model(input_size=128, lstm_size=128, lstm_layers=2, output_size=384, include_top=False,
              compute_dtype=torch.float32).to(cuda)
    p = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    filt = EEGFilters(1000, order=3)
    tr = DistillTrainer(m, filt.sos, loss="cosine")
    perm = torch.randperm(256, generator=torch.Generator().manual_seed(43))[:205]
    b = perm[shard_indices(205, 0, 43, 0, 1)][:16].to(cuda)
    loss = tr.train_step(ds.eeg_all[b], ds.features_all[b])
    x = ds.eeg_all[b].cpu().numpy()
    feat = lstm.model_forward(eeg_filter.eeg_bandpass_znorm(x, filt.sos), p, 2)
    assert abs(loss.item() - losses.cosine_similarity_loss(feat, ds.features_all[b].cpu().numpy())) < 1e-4


@pytest.mark.parametrize("B,T,C,H,L,dtype", [(8, 40, 16, 128, 2, torch.bfloat16), (5, 33, 12, 64, 3, torch.float32),
                                             (256, 70, 128, 768, 2, torch.bfloat16), (64, 40, 128, 1024, 2, torch.bfloat16)])
def test_direct_gradients_and_ready_hook(cuda, B, T, C, H, L, dtype):
    """The trainer's LSTM writes its gradients straight into the flat buffer (no temporaries, no accumulation pass) and
    announces every layer through csn_lstm_plan_set_grad_callback, top layer first, on every path (generic cells,
    per-diagonal launches, weight-stationary): same bits as autograd's accumulation of the library's outputs."""
    from cerebralsignalnetworks_amd.trainer import DistillTrainer
    rng = np.random.default_rng(B + T)
    x = dev_t(rng.standard_normal((B, C, T)).astype(np.float32), cuda)
    tgt = dev_t(rng.standard_normal((B, 24)).astype(np.float32), cuda)
    torch.manual_seed(3)
    m_ref = Model(input_size=C, lstm_size=H, lstm_layers=L, output_size=24, include_top=False, compute_dtype=dtype).to(cuda)
    m = Model(input_size=C, lstm_size=H, lstm_layers=L, output_size=24, include_top=False, compute_dtype=dtype).to(cuda)
    m.load_state_dict(m_ref.state_dict())
    CosineSimilarityLoss()(m_ref(x.transpose(1, 2).contiguous()), tgt).backward()
    want = torch.cat([p.grad.reshape(-1) for p in m_ref.parameters()])
    tr = DistillTrainer(m, None, loss="cosine", lr=1e-3, preprocess=False)
    assert m.lstm.direct_grads and m.lstm.grad_ready_hook is None          # one rank: nothing to overlap
    calls = []
    m.lstm.grad_ready_hook = calls.append
    tr.train_step(x, tgt)
    assert calls == list(range(L - 1, -1, -1)), calls
    assert torch.equal(tr.grads.flat, want)
    tr.train_step(x, tgt)                                                   # a second step: the hook is re-armed per backward
    assert calls == 2 * list(range(L - 1, -1, -1))
    # without the trainer's contract (direct_grads off) the same module accumulates as any autograd function does
    m.lstm.direct_grads = False
    m.lstm.grad_ready_hook = None
    tr.check_device_status()


def test_transform_eeg_data_lstm_by_list_runs_the_hip_model(cuda):
    """a12 (utils/PerilsEEGDataset.py:323-341): ``dataset.transformEEGDataLSTMByList(model, loader)`` with the HIP
    ``Model`` over loaders of the dataset, as LstmDistillFromDinoV2Train.py:381-387 calls it: the rows against the
    oracle's embeddings of the same items, the labels in both modes -- by dataset index, and (``compat_label_bug``) by
    the position inside the batch, which is what the reference's ``getLabelbyIndex(idx)`` at :338 looks up."""
    from torch.utils.data import DataLoader, Subset
    from cerebralsignalnetworks_amd.dataset import EEGDataset
    from cerebralsignalnetworks_amd.trainer import split_indices
    torch.manual_seed(7)
    N, C, T, H, D, bs = 90, 16, 48, 64, 24, 16
    ds = EEGDataset(synthetic=N, synthetic_channels=C, synthetic_samples=T, time_low=0, time_high=T, seed=11, device=cuda,
                    feature_dim=D)
    m = Model(input_size=C, lstm_size=H, lstm_layers=2, output_size=D, include_top=False,
              compute_dtype=torch.float32).to(cuda).eval()
    p = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    tr, te = split_indices(N, (0.8, 0.2), seed=43)
    for ix in (tr, te):
        loader = DataLoader(Subset(ds, ix.tolist()), batch_size=bs, shuffle=False)
        want = lstm.model_forward(ds.eeg_all[ix.to(cuda)].transpose(1, 2).double().cpu().numpy(), p, 2)    # items are eeg[T, C]
        for compat in (False, True):
            ds.compat_label_bug = compat
            feats, labs = ds.transformEEGDataLSTMByList(model=m, data_loader=loader)
            assert len(feats) == len(labs) == len(ix) and feats[0].shape == (D,)
            np.testing.assert_allclose(np.stack(feats), want, atol=2e-5)
            pos = [j % bs for j in range(len(ix))]                       # position of item j inside its batch
            expect = [ds.getLabelbyIndex(pos[j] if compat else int(ix[j])) for j in range(len(ix))]
            assert labs == expect
    # include_top=True: the tuple output's features are taken (the product does not iterate over the tuple)
    ds.compat_label_bug = False
    m2 = Model(input_size=C, lstm_size=H, lstm_layers=1, output_size=D, include_top=True,
               compute_dtype=torch.float32).to(cuda).eval()
    feats, labs = ds.transformEEGDataLSTMByList(model=m2, data_loader=DataLoader(Subset(ds, te.tolist()), batch_size=bs))
    assert len(feats) == len(te) and feats[0].shape == (D,)
    # bf16 path through the same call: the default compute dtype of the CLIs
    m3 = Model(input_size=C, lstm_size=128, lstm_layers=2, output_size=D, include_top=False).to(cuda).eval()
    p3 = {k: v.detach().cpu().numpy() for k, v in m3.state_dict().items()}
    feats, _ = ds.transformEEGDataLSTMByList(model=m3, data_loader=DataLoader(Subset(ds, te.tolist()), batch_size=bs))
    want = lstm.model_forward(ds.eeg_all[te.to(cuda)].transpose(1, 2).double().cpu().numpy(), p3, 2)
    np.testing.assert_allclose(np.stack(feats), want, atol=3e-2)
    from cerebralsignalnetworks_amd.trainer import check_device_status
    for mod in (m, m2, m3):
        check_device_status(mod)


def test_cli_compat_label_bug_changes_what_the_user_sees(cuda, tmp_path, capsys):
    """--compat_label_bug routes the retrieval legs of both CLIs through transformEEGDataLSTMByList (the reference's
    :381-387 / Eval :324-334): same embeddings, batch-local labels -> different Recall / Precision than the true labels."""
    import LstmDistillFromDinoV2Train as train
    import LstmDistillFromDinoV2Eval as evaluate_cli
    base = ["--synthetic", "192", "--batch_size", "16", "--num_epochs", "2", "--validation_frequency", "1", "--log_dir",
            str(tmp_path), "--hidden_size", "128", "--lstm_layers", "1", "--loss", "cosine", "--dtype", "f32", "--topK", "5"]
    seen = {}
    for tag, extra in (("true", []), ("compat", ["--compat_label_bug"])):
        train.main(base + extra)
        out = capsys.readouterr().out
        line = [l for l in out.splitlines() if l.startswith("Overall Recall")][-1]
        vline = [l for l in out.splitlines() if "val_loss:" in l][-1]
        seen[tag] = (line, float(vline.split("val_loss:")[1].split()[0]))
    assert seen["true"][0] != seen["compat"][0]                    # the flag is visible in the printed metrics
    assert abs(seen["true"][1] - seen["compat"][1]) < 1e-6         # ... and touches nothing else (same seed, same training)
    ckpt = os.path.join(str(tmp_path), "lstm_dinov2_best_loss.pth")
    r0 = evaluate_cli.main(base + ["--custom_model_weights", ckpt])
    r1 = evaluate_cli.main(base + ["--custom_model_weights", ckpt, "--compat_label_bug"])
    np.testing.assert_array_equal(r0["I"], r1["I"])                # same embeddings, same neighbours
    assert (r0["Recall_Total"], r0["Precision_Total"]) != (r1["Recall_Total"], r1["Precision_Total"])


def test_featdist_and_kd_losses_train_on_gpu(cuda):
    """The reference's active losses (FeatureDistributionLoss, loss_fn_kd) through the HIP model."""
    from cerebralsignalnetworks_amd.trainer import DistillTrainer
    from cerebralsignalnetworks_amd.losses import FeatureDistributionLoss, loss_fn_kd
    rng = np.random.default_rng(2)
    B, C, T, H, D = 8, 16, 40, 64, 24
    x = rng.standard_normal((B, C, T)).astype(np.float32)
    tgt = rng.standard_normal((B, D)).astype(np.float32)
    lab = rng.integers(0, 40, B)
    m = Model(input_size=C, lstm_size=H, lstm_layers=1, output_size=D, include_top=True,
              compute_dtype=torch.float32).to(cuda)
    p = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    tr = DistillTrainer(m, None, loss="featdist", nepochs=100)
    loss = tr.train_step(dev_t(x, cuda), dev_t(tgt, cuda), dev_t(lab, cuda), epoch=25)
    (feat, cls) = lstm.model_forward(eeg_filter.eeg_bandpass_znorm(x, np.zeros((0, 6))), p, 1, include_top=True)
    want = losses.feature_distribution_loss(feat, tgt, losses.teacher_temp_schedule(100)[25], lab, cls)
    assert abs(loss.item() - want) < 1e-4

    class KD:
        alpha, temperature = 0.5, 4.0
    m2 = Model(input_size=C, lstm_size=H, lstm_layers=1, output_size=40, include_top=False,
               compute_dtype=torch.float32).to(cuda)
    p2 = {k: v.detach().cpu().numpy() for k, v in m2.state_dict().items()}
    teacher = rng.standard_normal((B, 40)).astype(np.float32)
    tr2 = DistillTrainer(m2, None, loss="kd", kd_params=KD)
    loss2 = tr2.train_step(dev_t(x, cuda), dev_t(teacher, cuda), dev_t(lab, cuda))
    feat2 = lstm.model_forward(eeg_filter.eeg_bandpass_znorm(x, np.zeros((0, 6))), p2, 1)
    assert abs(loss2.item() - losses.loss_fn_kd(feat2, lab, teacher, 0.5, 4.0)) < 1e-4


def test_barlow_training_step_on_lstm_embeddings(cuda):
    """BASELINE.json configs[4] shape of the loss: Barlow-Twins on LSTM embeddings, HIP off-diagonal reduction,
    LARS step -- loss vs the oracle restatement of net.py:33-42."""
    from cerebralsignalnetworks_amd.trainer import DistillTrainer
    rng = np.random.default_rng(6)
    B, C, T, H, D = 32, 16, 24, 64, 48
    x = rng.standard_normal((B, C, T)).astype(np.float32)
    tgt = rng.standard_normal((B, D)).astype(np.float32)
    m = Model(input_size=C, lstm_size=H, lstm_layers=1, output_size=D, include_top=False,
              compute_dtype=torch.float32).to(cuda)
    p = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    tr = DistillTrainer(m, None, loss="barlow", optimizer="lars", lr=0.1)
    loss = tr.train_step(dev_t(x, cuda), dev_t(tgt, cuda))
    feat = lstm.model_forward(eeg_filter.eeg_bandpass_znorm(x, np.zeros((0, 6))), p, 1)
    want, _ = losses.barlow_loss(feat, tgt, B)
    assert abs(loss.item() - want) < 1e-3 * want
    moved = (m.fc.weight.detach().cpu().numpy() != p["fc.weight"]).mean()
    assert moved > 0.99


def test_zero_phase_filtfilt_matches_scipy_golden(cuda, golden):
    """Utilities.remove_noise (butter-4, 1-50 Hz, filtfilt) on [S,T,C].  The (b,a) polynomial form scipy
    evaluates is ill-conditioned (two float64 evaluations differ at 5e-5, see test_oracle); the device
    evaluates the same filter as a float64 biquad cascade, so agreement is held to 2e-4."""
    from cerebralsignalnetworks_amd import remove_noise
    g = golden("filter_apply.npz")
    x = g["filtfilt_x"].astype(np.float32)                       # [2,500,8]
    y = remove_noise(dev_t(x, cuda), 1000).cpu().numpy()
    np.testing.assert_allclose(y, g["filtfilt_y"], atol=2e-4)
    want = eeg_filter.remove_noise(x, 1000)                      # oracle on the float32-rounded input
    np.testing.assert_allclose(y, want, atol=2e-4)
    # linearity (size-independent property): filtfilt(2.5 x) == 2.5 filtfilt(x)
    y2 = remove_noise(dev_t(2.5 * x, cuda), 1000).cpu().numpy()
    np.testing.assert_allclose(y2, 2.5 * y, atol=1e-5)


def test_several_forwards_outstanding_before_backward(cuda):
    """Multi-crop pattern of LstmDistillation.py:570-585: the same module is run on several views (some of
    equal length) before one backward; every view keeps its own workspace until its backward has run."""
    rng = np.random.default_rng(12)
    B, C, H, L = 6, 16, 32, 2
    p = lstm.init_params(C, H, L, 8, None, seed=2)
    lp = {k[len("lstm."):]: v for k, v in p.items() if k.startswith("lstm.")}
    views = [rng.standard_normal((B, T, C)).astype(np.float32) for T in (30, 30, 20, 20, 20)]
    w = [rng.standard_normal((B, H)).astype(np.float32) for _ in views]
    m = _model_from_params(p, C, H, L, 8, None, torch.float32, cuda)
    total = sum((m.lstm(dev_t(v, cuda)) * dev_t(wi, cuda)).sum() for v, wi in zip(views, w))
    total.backward()
    want = {}
    for v, wi in zip(views, w):
        y, saved = lstm.lstm_forward(v, lp, L, return_saved=True)
        dy = np.zeros_like(y)
        dy[:, -1] = wi
        _, g = lstm.lstm_backward(dy, lp, saved, L)
        for k2, gv in g.items():
            want[k2] = want.get(k2, 0) + gv
    for k2, gv in want.items():
        got = getattr(m.lstm, k2).grad.cpu().numpy()
        np.testing.assert_allclose(got, gv, atol=1e-4 * max(1.0, np.abs(gv).max()), err_msg=k2)
    assert all(not pl.busy for pl in m.lstm.all_plans())


def test_dino_self_distillation_trainer_runs(cuda, tmp_path):
    """LstmDistillation.py drop-in: 2 epochs on synthetic 96-channel data, 6 temporal crops per step, EMA teacher,
    checkpoint dict with the reference's keys."""
    import LstmDistillation as dino_cli
    hist = dino_cli.main(["--synthetic", "80", "--batch_size_per_gpu", "16", "--epochs", "2", "--embed_dim", "128",
                          "--lstm_layers", "2", "--out_dim", "64", "--log_dir", str(tmp_path), "--warmup_epochs", "1",
                          "--warmup_teacher_temp_epochs", "1"])
    assert len(hist) == 2 and all(np.isfinite(h) for h in hist)
    ck = torch.load(os.path.join(str(tmp_path), "checkpoint.pth"), weights_only=False)
    assert {"student", "teacher", "optimizer", "epoch", "args", "dino_loss"} <= set(ck)
    assert any(k.startswith("backbone.lstm.weight_hh_l1") for k in ck["teacher"])
    # the Eval script's loader strips "backbone." (Eval.py:310-313)
    from LstmDistillFromDinoV2Eval import load_checkpoint_into
    m = Model(input_size=96, lstm_size=128, lstm_layers=2, output_size=128, include_top=False)
    torch.save({"teacher": ck["teacher"]}, os.path.join(str(tmp_path), "t.pth"))
    res = load_checkpoint_into(m, os.path.join(str(tmp_path), "t.pth"))
    assert not [k for k in res.missing_keys if k.startswith("lstm.")]


def test_handoffs_identical_under_uneven_load(cuda):
    """The in-kernel hand-offs of the weight-stationary kernels (flags + slabs through one XCD's L2) while a second
    stream streams 256 MB copies through HBM / L2 and takes CUs at random moments: every output and gradient must
    equal the quiet run bit for bit and no bounded wait may time out (tools/handoff_stress.py is the long form)."""
    from cerebralsignalnetworks_amd.lstm_model import HipLSTM
    B, T, C, H, L = 256, 96, 128, 768, 2
    torch.manual_seed(1)
    m = HipLSTM(C, H, L, compute_dtype=torch.bfloat16).to(cuda)
    x = torch.randn(B, T, C, device=cuda)
    dy = torch.randn(B, H, device=cuda)

    def run():
        for p in m.parameters():
            p.grad = None
        xt = x.clone().requires_grad_(True)
        y = m(xt)
        (y * dy).sum().backward()
        torch.cuda.synchronize()
        for plan in m.all_plans():
            assert plan.status() == 0
        return [y.detach().clone()] + [p.grad.clone() for p in m.parameters()] + [xt.grad.clone()]

    ref = run()
    side = torch.cuda.Stream()
    big = [torch.empty(64 << 20, dtype=torch.float32, device=cuda) for _ in range(3)]
    for rep in range(4):
        with torch.cuda.stream(side):
            for k in range(12 + 6 * rep):
                big[(k + 1) % 3].copy_(big[k % 3])
        out = run()
        side.synchronize()
        for a, b in zip(out, ref):
            assert torch.equal(a, b), f"repetition {rep}: outputs differ under load"



@pytest.mark.gpu
@pytest.mark.parametrize("B,T,C,H,L", [(256, 24, 128, 768, 2),      # the benchmark width, all four M-tiles (192 workgroups)
                                       (70, 37, 24, 128, 3),        # ragged last tile, 8 slices, three layers
                                       (130, 21, 16, 512, 2), (64, 18, 128, 1024, 2), (5, 9, 8, 384, 1), (33, 1, 8, 256, 2),
                                       (400, 6, 16, 768, 1)])      # 7 M-tiles at 48 slices: two launches per layer (5 + 2 tiles)
def test_f32_weight_stationary_recurrence(cuda, B, T, C, H, L):
    """The exact-float32 path's weight-stationary kernels (lstm_f32_persist.hip, one launch per layer) against the float64
    oracle AND against the per-step launches they replace (CSN_NO_PERSIST): the forward keeps the operand map, the k order and
    the order of the K-quarter partial sums of lstm_cell_fwd_ks_kernel, so its outputs are the same BITS; the backward groups
    K = 4H into four partial sums where the per-step kernel uses eight -- agreement to float32 rounding."""
    rng = np.random.default_rng(B * T + H)
    p = lstm.init_params(C, H, L, 8, None, seed=5)
    lp = {k[len("lstm."):]: v for k, v in p.items() if k.startswith("lstm.")}
    x = rng.standard_normal((B, T, C)).astype(np.float32)
    dy_all = (rng.standard_normal((B, T, H)) * 0.1).astype(np.float32)
    dy_last = rng.standard_normal((B, H)).astype(np.float32)
    y, saved = lstm.lstm_forward(x, lp, L, return_saved=True)
    dy = dy_all.astype(np.float64).copy()
    dy[:, -1] += dy_last
    dx_ref, g_ref = lstm.lstm_backward(dy, lp, saved, L)
    names = {}
    ws = _run_lstm(p, x, dy_all, dy_last, C, H, L, torch.float32, cuda, info=names)
    per_step = _run_lstm(p, x, dy_all, dy_last, C, H, L, torch.float32, cuda, {"CSN_NO_PERSIST": "1"})
    _assert_same_bits(ws["y_all"], per_step["y_all"], "weight-stationary vs per-step float32 forward: y_all")
    for k in ws:
        # (bias gradients: the weight-stationary backward sums dgates over the steps as it produces them -- per lane over t, then
        #  over the 16 rows of a row group, then the row groups in fixed order -- where the per-step path runs a column-sum pass)
        assert _rel(ws[k], per_step[k]) < (1e-5 if "bias" in k else 2e-6), (k, _rel(ws[k], per_step[k]))
    assert np.abs(ws["y_all"] - y).max() < 2e-5
    assert _rel(ws["dx"], dx_ref) < 1e-5
    for k, v in g_ref.items():
        assert _rel(ws[k], v) < 1e-5, (k, _rel(ws[k], v))
    # identical rows give identical bits whatever tile / row group they sit in (every row sees one summation order)
    if B >= 130:
        xx = x.copy()
        xx[129] = xx[0]
        xx[64] = xx[0]
        dya, dyl = dy_all.copy(), dy_last.copy()
        dya[129] = dya[64] = dya[0]
        dyl[129] = dyl[64] = dyl[0]
        rep = _run_lstm(p, xx, dya, dyl, C, H, L, torch.float32, cuda)
        for r in (64, 129):
            _assert_same_bits(rep["y_all"][r], rep["y_all"][0], f"row {r} vs row 0: y_all")
            _assert_same_bits(rep["dx"][r], rep["dx"][0], f"row {r} vs row 0: dx")


@pytest.mark.gpu
def test_f32_weight_stationary_path_is_the_one_that_runs(cuda):
    m = Model(input_size=16, lstm_size=128, lstm_layers=2, output_size=8, include_top=False, compute_dtype=torch.float32).to(cuda)
    x = torch.randn(4, 12, 16, device=cuda)
    m.lstm(x, want_all=True)
    torch.cuda.synchronize()
    plans = m.lstm.all_plans()
    assert plans and all(pl.kernel_names() == ("lstm_fwd_f32_persist_kernel", "lstm_bwd_f32_persist_kernel") and pl.path() == 4
                         for pl in plans)
    # inference plans (no saved gates, no gradient slots) give the bits of the training plan's forward
    y_train, _ = m.lstm(x, want_all=True)
    m.eval()
    with torch.no_grad():
        y_eval, _ = m.lstm(x, want_all=True)
    torch.cuda.synchronize()
    assert any(not pl.training for pl in m.lstm.all_plans()) and all(pl.status() == 0 for pl in m.lstm.all_plans())
    _assert_same_bits(y_eval.cpu().numpy(), y_train.detach().cpu().numpy(), "inference plan vs training plan: y_all")
    os.environ["CSN_NO_PERSIST"] = "1"
    try:
        m2 = Model(input_size=16, lstm_size=128, lstm_layers=2, output_size=8, include_top=False, compute_dtype=torch.float32).to(cuda)
        m2.lstm(x, want_all=True)
        torch.cuda.synchronize()
        assert all(pl.path() == 0 and pl.kernel_names()[0] == "lstm_cell_fwd_ks_kernel" for pl in m2.lstm.all_plans())
    finally:
        del os.environ["CSN_NO_PERSIST"]


@pytest.mark.gpu
@pytest.mark.parametrize("dt", ["bf16", "f32"])
def test_two_weight_stationary_plans_on_two_streams(cuda, dt):
    """tests/diag/two_streams.py: two models of the benchmark width, forward + backward enqueued on two streams so that their
    weight-stationary launches (every workgroup of a launch co-resident, one per CU) can meet on the device: no bounded wait
    times out and the gradients are the bits of the serial runs."""
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "diag", "two_streams.py"), "4", dt],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
