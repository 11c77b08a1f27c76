"""GPU acceptance tests at the benchmark sizes, against fixtures the REFERENCE's own classes produced
(tests/golden/make_ref_goldens.py: LSTMDistillRetreival.LSTMModel + CosineSimilarityLoss executed on torch CPU).

  * cfg2 (T 500, C 128, H 768, L 2, D 384) and cfg4 (T 440, H 1024) training step: batch 256 = 32 copies of the 8
    fixture segments, so every 64-row M-tile / XCD hand-off group of the weight-stationary kernels sees all 8
    segments.  The loss is a mean over the batch, so loss and every parameter gradient equal the 8-segment
    fixture's.  float32 path: features, loss (north star: 1e-4) and gradients to 1e-4 of their scale; bf16 MFMA
    path: bounds stated below.
  * retrieval acceptance (north star): top-1 of the bf16 path within +-0.5 % of the CPU reference on a seeded
    clustered set of 2048 gallery / 512 query segments, 40 classes, cfg2 model.
"""
import numpy as np
import pytest
import torch

from cerebralsignalnetworks_amd import cabi, Model, CosineSimilarityLoss, EEGFilters
from cerebralsignalnetworks_amd.dataset import clustered_eeg
from oracle import eeg_filter, losses, lstm

pytestmark = pytest.mark.gpu

COPIES = 32


def _model(p, C, H, L, D, dtype, cuda):
    m = Model(input_size=C, lstm_size=H, lstm_layers=L, output_size=D, include_top=False, compute_dtype=dtype)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in p.items()})
    return m.to(cuda)


def _status_ok(m):
    torch.cuda.synchronize()
    for plan in m.lstm.all_plans():
        assert plan.status() == 0, "an in-kernel hand-off of a weight-stationary kernel timed out"


def _grad_checks(g, name, got, rel_tol, what):
    """got: full gradient (numpy f64).  Fixture holds it in full, or as sample + norm + two random projections."""
    got = np.asarray(got, np.float64)
    if f"grad__{name}" in g.files:
        want = g[f"grad__{name}"].astype(np.float64)
        scale = max(1e-6, np.abs(want).max())
        assert np.abs(got - want).max() <= rel_tol * scale, (what, name, np.abs(got - want).max(), scale)
        return
    r = np.random.default_rng(5)
    want_s = g[f"gsamp__{name}"]
    scale = max(1e-6, np.abs(want_s).max())
    assert np.abs(got[::37, ::41] - want_s).max() <= rel_tol * scale, (what, name, "sample")
    pr = got @ r.standard_normal(got.shape[1])
    pl = r.standard_normal(got.shape[0]) @ got
    for mine, key in ((pr, "gprojr"), (pl, "gprojl")):
        want = g[f"{key}__{name}"]
        assert np.linalg.norm(mine - want) <= rel_tol * max(1e-6, np.linalg.norm(want)) * 4, (what, name, key)
    n = float(g[f"gnorm__{name}"])
    assert abs(np.linalg.norm(got) - n) <= rel_tol * n * 4, (what, name, "norm")


def _grad_rel_err(g, name, got):
    """One number per gradient: max |got - want| / max |want| where the fixture holds the gradient in full, else the
    largest of the relative errors of the strided sample, the two random projections and the norm."""
    got = np.asarray(got, np.float64)
    if f"grad__{name}" in g.files:
        want = g[f"grad__{name}"].astype(np.float64)
        return float(np.abs(got - want).max() / max(1e-6, np.abs(want).max()))
    r = np.random.default_rng(5)
    want_s = g[f"gsamp__{name}"]
    e = [np.abs(got[::37, ::41] - want_s).max() / max(1e-6, np.abs(want_s).max())]
    pr = got @ r.standard_normal(got.shape[1])
    pl = r.standard_normal(got.shape[0]) @ got
    for mine, key in ((pr, "gprojr"), (pl, "gprojl")):
        want = g[f"{key}__{name}"]
        e.append(np.linalg.norm(mine - want) / max(1e-6, np.linalg.norm(want)))
    n = float(g[f"gnorm__{name}"])
    e.append(abs(np.linalg.norm(got) - n) / n)
    return float(max(e))


def _write_report(name, obj):
    import json
    import os
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, name), "w") as fh:
            json.dump(obj, fh, indent=1, sort_keys=True)
    except OSError:
        pass


# bf16 path against the f64 fixture: 2 x the values measured on MI355X in round 3 (profiles/r03_parity_cfg*_bf16.json)
# measured: cfg2 feat 1.55e-4, loss 6.0e-7, worst gradient 0.57 %; cfg4 feat 1.03e-4, loss 2.4e-6, worst gradient 0.62 %
# (round 2's bounds were 3e-2 / 5e-3 / 6 %: 50 - 8000 x looser than what the kernels do)
# cfg4's loss bound (round 4): the scalar loss error of the bf16 path moves with the ORDER of the float32 accumulation over K
# -- 2.4e-6 (round 3), 5.0e-6 (layer-0 projection fused: its MFMAs now come first), 8.8e-6 (K walk rotated in steps of 4
# k-blocks for the LDS-DMA pieces) for the same kernels' arithmetic; features and gradients do not move (1.0e-4, 0.6 %).
# 2e-5 covers the orders seen with a factor 2 and is 5 x inside the north star's 1e-4.
BF16_BOUNDS = {"cfg2": {"feat": 3.2e-4, "loss": 1.3e-6, "grad": 1.2e-2},
               "cfg4": {"feat": 2.1e-4, "loss": 2.0e-5, "grad": 1.3e-2}}


@pytest.mark.parametrize("tag", ["cfg2", "cfg4"])
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_training_step_matches_reference_fixture(cuda, golden, tag, dtype):
    g = golden(f"ref_lstm_{tag}.npz")
    B8, T, C, H, L, D = (int(v) for v in g["dims"])
    p = lstm.init_params(C, H, L, D, None, seed=int(g["seed_params"]))
    rng = np.random.default_rng(int(g["seed_x"]))
    x8 = rng.standard_normal((B8, T, C)).astype(np.float32)
    tgt8 = rng.standard_normal((B8, D)).astype(np.float32)
    x = np.tile(x8, (COPIES, 1, 1))                     # row r = segment r % 8
    tgt = np.tile(tgt8, (COPIES, 1))
    dt = torch.float32 if dtype == "f32" else torch.bfloat16
    m = _model(p, C, H, L, D, dt, cuda)
    xt = torch.from_numpy(x).to(cuda).requires_grad_(True)
    feat = m(xt)
    loss = CosineSimilarityLoss()(feat, torch.from_numpy(tgt).to(cuda))
    loss.backward()
    _status_ok(m)
    f = feat.detach().cpu().numpy()
    dx = xt.grad.detach().cpu().numpy()
    # every copy of a segment -- in whatever M-tile / hand-off group it sits -- gives the same bits: features, and the
    # input gradient at every one of the T timesteps (a stale or half-written backward hand-off in any tile / step
    # shows up as a block of differing dx rows, not as 0.1 % of a norm)
    for c in range(1, COPIES):
        np.testing.assert_array_equal(f[c * B8:(c + 1) * B8], f[:B8], err_msg=f"copy {c}")
        np.testing.assert_array_equal(dx[c * B8:(c + 1) * B8], dx[:B8], err_msg=f"dx of copy {c}")
    assert np.isfinite(dx).all() and np.abs(dx[:B8]).max() > 0
    want_feat, want_loss = g["feat_f64"], float(g["loss_f64"])
    grads = {n: q.grad.detach().double().cpu().numpy() for n, q in m.named_parameters()}
    # measured errors of this run, kept: gpurun_out/parity_<tag>_<dtype>.json (copied to profiles/ per round)
    errs = {"feat_max_abs": float(np.abs(f[:B8] - want_feat).max()), "loss_abs": abs(loss.item() - want_loss),
            "loss": loss.item(), "loss_reference_f64": want_loss, "grad_rel": {}}
    for n, got in grads.items():
        errs["grad_rel"][n] = _grad_rel_err(g, n, got)
    _write_report(f"parity_{tag}_{dtype}.json", errs)
    print(f"measured {tag} {dtype}:", errs)
    if dtype == "f32":
        np.testing.assert_allclose(f[:B8], want_feat, atol=5e-5)
        assert abs(loss.item() - want_loss) < 1e-4                       # north star: distill loss within 1e-4 (fp32)
        assert abs(loss.item() - float(g["loss_f32"])) < 1e-4            # ... of the reference's own f32 CPU run
        for n, got in grads.items():
            _grad_checks(g, n, got, 1e-4, f"{tag} f32")
    else:
        # bf16 operands (8 significant bits), f32 accumulate / state, 2 x T recurrent steps.  Bounds = 2 x the errors
        # measured on MI355X (profiles/r03_parity_*.json)
        b = BF16_BOUNDS[tag]
        assert errs["feat_max_abs"] < b["feat"], errs
        cos = (f[:B8] * want_feat).sum(1) / np.linalg.norm(f[:B8], axis=1) / np.linalg.norm(want_feat, axis=1)
        assert cos.min() > 0.9995
        assert errs["loss_abs"] < b["loss"], errs
        for n, e in errs["grad_rel"].items():
            assert e < b["grad"], (n, e, errs)


def test_cfg2_three_layers_stays_co_resident(cuda):
    """B 256, H 768, L 3: more hand-off groups than XCDs -- the launches of the stream form must not need more
    workgroups resident than the chip has CUs (a partly dispatched launch spins until its time-out)."""
    rng = np.random.default_rng(3)
    B, T, C, H, L, D = 256, 40, 128, 768, 3, 32
    p = lstm.init_params(C, H, L, D, None, seed=7)
    x = rng.standard_normal((B, T, C)).astype(np.float32)
    res = {}
    for dt in (torch.bfloat16, torch.float32):
        m = _model(p, C, H, L, D, dt, cuda)
        feat = m(torch.from_numpy(x).to(cuda))
        feat.square().mean().backward()
        _status_ok(m)
        res[dt] = (feat.detach().cpu().numpy(), m.lstm.weight_hh_l0.grad.cpu().numpy())
    assert np.abs(res[torch.bfloat16][0] - res[torch.float32][0]).max() < 3e-2
    a, b = res[torch.bfloat16][1], res[torch.float32][1]
    assert np.linalg.norm(a - b) < 6e-2 * np.linalg.norm(b)


def _embed(model, filt, x, cuda, batch=256):
    outs = []
    with torch.no_grad():
        for i in range(0, x.shape[0], batch):
            eeg = filt.apply(torch.from_numpy(x[i:i + batch]).to(cuda))          # HIP band-pass + z-score -> [b, T, C]
            outs.append(model(eeg).float())
    return torch.cat(outs)


@pytest.mark.parametrize("tag", ["cfg2", "cfg4"])
def test_bf16_retrieval_top1_within_half_percent_of_cpu_reference(cuda, golden, tag):
    """BASELINE.json north star: 'retrieval top-1 within +-0.5 % of the CPU reference'.  Reference = scipy sosfilt
    + z-score -> the reference's LSTMModel on torch CPU f32 -> exact L2 top-5 (fixture).  Here: HIP filter -> HIP
    LSTM (bf16 fast path, and the f32 path) -> csn_l2_topk.  cfg2 = the headline shapes (2048 gallery / 512 query);
    cfg4 = BASELINE.json configs[3], 'retrieval eval vs CPU ref' at the Spampinato shapes 128 x 440, hidden 1024
    (LstmDistillFromDinoV2TrainSpampinato.py:368; 1536 / 384)."""
    g = golden(f"ref_retrieval_{tag}.npz")
    ng, nq = int(g["n_gallery"]), int(g["n_query"])
    C, T, H, L, D = (128, 500, 768, 2, 384) if tag == "cfg2" else tuple(int(v) for v in g["dims"])
    x, labels = clustered_eeg(ng + nq, T=T, seed=int(g["seed"]), snr=float(g["snr"]))
    np.testing.assert_array_equal(labels, g["labels"])
    p = lstm.init_params(C, H, L, D, None, seed=43)
    filt = EEGFilters(1000, order=3)
    ref_top5, ref_top1 = g["top5"], float(g["top1"])
    report = {}
    for name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        m = _model(p, C, H, L, D, dt, cuda).eval()
        emb = _embed(m, filt, x, cuda)
        _status_ok(m)
        dist, idx = cabi.l2_topk(emb[:ng].contiguous(), emb[ng:].contiguous(), 5)
        idx = idx.cpu().numpy()
        top1 = float((labels[:ng][idx[:, 0]] == labels[ng:]).mean())
        same_nn = float((idx[:, 0] == ref_top5[:, 0]).mean())
        overlap5 = float(np.mean([len(set(a) & set(b)) / 5.0 for a, b in zip(idx, ref_top5)]))
        err = float(np.abs(emb.cpu().numpy()[::16] - g["emb_sample"]).max())
        report[name] = dict(top1=top1, same_nn=same_nn, overlap5=overlap5, emb_err=err)
    print(f"retrieval acceptance {tag}:", dict(reference_top1=ref_top1, **report))
    _write_report(f"retrieval_{tag}.json", dict(reference_top1=ref_top1, **report))
    assert 0.3 < ref_top1 < 0.95                                  # a set on which precision can matter
    assert report["f32"]["emb_err"] < 2e-4
    assert abs(report["f32"]["top1"] - ref_top1) <= 0.005 and report["f32"]["same_nn"] >= 0.99
    assert abs(report["bf16"]["top1"] - ref_top1) <= 0.005, report     # the acceptance criterion
    assert report["bf16"]["emb_err"] < 3e-2 and report["bf16"]["overlap5"] > 0.6


def test_timed_out_handoff_in_an_early_step_is_still_reported(cuda):
    """The status word is sticky: a bounded in-kernel wait that gives up in step k < n must still be visible to the
    check after step n (it used to be cleared by every forward, so only the last step was ever inspected)."""
    from cerebralsignalnetworks_amd.trainer import DistillTrainer
    rng = np.random.default_rng(0)
    B, C, T, H, D = 64, 32, 40, 128, 16
    m = Model(input_size=C, lstm_size=H, lstm_layers=2, output_size=D, include_top=False).to(cuda)
    tr = DistillTrainer(m, None, loss="cosine")
    x = torch.from_numpy(rng.standard_normal((B, C, T)).astype(np.float32)).to(cuda)
    tg = torch.from_numpy(rng.standard_normal((B, D)).astype(np.float32)).to(cuda)
    tr.train_step(x, tg)
    tr.check_device_status()                      # clean so far
    for plan in m.lstm.all_plans():
        plan.inject_timeout()                     # "step 2 timed out"
    for _ in range(3):
        tr.train_step(x, tg)                      # later steps must not erase it (and must not hang)
    with pytest.raises(RuntimeError, match="timed out"):
        tr.check_device_status()
    tr.check_device_status()                      # reported once, then cleared: training can go on


def test_nan_in_the_backward_is_a_diverged_run_not_a_timeout(cuda):
    """With the hand-off by data a NaN in dgates looks, at first sight, like an operand that has not arrived (the
    sentinel is a pair of bf16 NaNs).  The backward tells the two apart by bit pattern: once every piece of the operand
    is seen to be data, a non-finite product is a genuine non-finite gradient -- it is propagated like the reference's
    autograd does (NaN loss, NaN gradients), costs no bounded wait, and raises its OWN status bit, never the
    time-out bit.  A NaN in the forward's h (an arithmetic NaN, not the sentinel pattern) does not stall the forward."""
    import time
    from cerebralsignalnetworks_amd import cabi
    from cerebralsignalnetworks_amd.trainer import DistillTrainer
    rng = np.random.default_rng(1)
    for B, C, T, H, D in ((64, 32, 40, 128, 16), (256, 128, 70, 768, 32)):
        m = Model(input_size=C, lstm_size=H, lstm_layers=2, output_size=D, include_top=False).to(cuda)
        tr = DistillTrainer(m, None, loss="cosine", optimizer="adam")
        x = torch.from_numpy(rng.standard_normal((B, C, T)).astype(np.float32)).to(cuda)
        tg = torch.from_numpy(rng.standard_normal((B, D)).astype(np.float32)).to(cuda)
        tr.train_step(x, tg)
        tr.check_device_status()
        x_bad = x.clone()
        x_bad[3, :, 7] = float("nan")                      # one NaN sample: h of that row is NaN from step 7 on
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loss = tr.train_step(x_bad, tg)
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        assert elapsed < 0.15, elapsed                     # not even ONE bounded wait of 0.2 s
        assert not np.isfinite(float(loss))
        assert not torch.isfinite(m.lstm.weight_hh_l0.grad).any()      # the mean over the batch spreads the NaN, as in torch
        st = 0
        for plan in m.lstm.all_plans():
            st |= plan.status()
        assert st == cabi.STATUS_NONFINITE, st             # its own bit; the time-out bit stays clear
        with pytest.raises(FloatingPointError, match="non-finite"):
            tr.check_device_status()
        tr.check_device_status()                           # reported once, then cleared


def test_plans_are_independent_across_streams_and_threads(cuda):
    """The library keeps no global state: two plans driven from two host threads on two HIP streams at the same
    time give the bits each gives alone (side streams, event pools and profiling events are per plan)."""
    import threading
    rng = np.random.default_rng(11)
    shapes = [(64, 48, 32, 128, 2), (48, 40, 16, 256, 2)]
    jobs = []
    for B, T, C, H, L in shapes:
        p = lstm.init_params(C, H, L, 8, None, seed=B)
        x = torch.from_numpy(rng.standard_normal((B, T, C)).astype(np.float32)).to(cuda)
        jobs.append((p, x, C, H, L))

    def run(job, stream, out, reps):
        p, x, C, H, L = job
        with torch.cuda.stream(stream):
            for _ in range(reps):
                m = _model(p, C, H, L, 8, torch.bfloat16, cuda)
                m.lstm.all_plans()                          # (plans are created lazily, inside forward)
                feat = m(x)
                feat.square().mean().backward()
                stream.synchronize()
                assert all(pl.status() == 0 for pl in m.lstm.all_plans())
                out.append((feat.detach().cpu().numpy(), m.lstm.weight_hh_l1.grad.cpu().numpy()))

    alone = []
    for job in jobs:
        out = []
        run(job, torch.cuda.Stream(device=cuda), out, 1)
        alone.append(out[0])
    outs = [[], []]
    threads = [threading.Thread(target=run, args=(jobs[i], torch.cuda.Stream(device=cuda), outs[i], 4)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for i in range(2):
        assert len(outs[i]) == 4
        for feat, grad in outs[i]:
            np.testing.assert_array_equal(feat, alone[i][0])
            np.testing.assert_array_equal(grad, alone[i][1])


def test_stateless_entry_points_are_independent_across_streams_and_threads(cuda):
    """ABI 3: csn_eeg_bandpass_znorm keeps no device buffer between calls (its chunk-scan basis travels with the kernel
    arguments) and csn_cosine_loss works in caller-owned scratch.  Two host threads on two HIP streams -- one running
    the order-3 band-pass + z-score and cosine losses of batch 256, the other the order-5 band-pass, the order-4
    zero-phase filter and cosine losses of batch 96 -- get, every time, the bits each call gives alone (the former
    per-device basis buffer was rewritten by every call on the caller's stream; the former per-process cosine buffer
    was freed and re-allocated when the batch grew)."""
    import threading
    from scipy.signal import butter
    from cerebralsignalnetworks_amd import cabi
    rng = np.random.default_rng(5)
    sos3 = EEGFilters(1000, order=3).sos
    sos5 = EEGFilters(1000, order=5).sos
    sos_ff = butter(4, [1.0 / 500.0, 50.0 / 500.0], btype="bandpass", output="sos")        # Utilities.remove_noise's band
    xa = torch.from_numpy(eeg_filter.synthetic_eeg(64, 128, 500, seed=3)).to(cuda)
    xb = torch.from_numpy(eeg_filter.synthetic_eeg(24, 96, 440, seed=4)).to(cuda)
    xf = torch.from_numpy(rng.standard_normal((6, 300, 64)).astype(np.float32)).to(cuda)
    sa, ta = (torch.from_numpy(rng.standard_normal((256, 384)).astype(np.float32)).to(cuda) for _ in range(2))
    sb, tb = (torch.from_numpy(rng.standard_normal((96, 200)).astype(np.float32)).to(cuda) for _ in range(2))

    def job_a():
        y = cabi.eeg_bandpass_znorm(xa, sos3, ddof=0, time_major=True)
        l, d = cabi.cosine_loss(sa, ta)
        return [y, l, d]

    def job_b():
        y = cabi.eeg_bandpass_znorm(xb, sos5, ddof=1)
        z = cabi.eeg_filtfilt(xf, sos_ff)
        l, d = cabi.cosine_loss(sb, tb)
        return [y, z, l, d]

    def run(job, stream, out, reps):
        with torch.cuda.stream(stream):
            for _ in range(reps):
                res = job()
                stream.synchronize()
                out.append([r.cpu().numpy() for r in res])

    alone = []
    for job in (job_a, job_b):
        out = []
        run(job, torch.cuda.Stream(device=cuda), out, 1)
        alone.append(out[0])
    outs = [[], []]
    threads = [threading.Thread(target=run, args=(job, torch.cuda.Stream(device=cuda), outs[i], 25))
               for i, job in enumerate((job_a, job_b))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for i in range(2):
        assert len(outs[i]) == 25
        for res in outs[i]:
            for got, want in zip(res, alone[i]):
                np.testing.assert_array_equal(got, want)
    # and the filter result alone is the oracle's (order 5, ddof 1, T = 440: 14 of the 16 chunk lanes active)
    ref = eeg_filter.eeg_bandpass_znorm(xb.cpu().numpy(), sos5, ddof=1)
    assert np.abs(alone[1][0] - ref).max() < 2e-6


def test_dino_self_distillation_step_matches_reference_fixture(cuda, golden):
    """f4 (LstmDistillation.py:518-596): one step of the DINO trainer -- student over 2 global + 4 local temporal
    views, teacher over the 2 global ones, DINOLoss, backward -- on the HIP LSTM (f32 path), against the fixture made
    by executing the reference's MultiCropWrapper / DINOHead / DINOLoss (make_ref_goldens.ref_dino_step)."""
    from cerebralsignalnetworks_amd.dino import DINOHead, DINOLoss, MultiCropWrapper
    g = golden("ref_dino_step.npz")
    B, C, H, L, OUT = (int(v) for v in g["dims"])
    p = lstm.init_params(C, H, L, H, None, seed=int(g["seed_params"]))
    head_sd = {k[len("sd__head."):]: torch.from_numpy(g[k]).float() for k in g.files if k.startswith("sd__head.")}

    def build():
        bb = Model(input_size=C, lstm_size=H, lstm_layers=L, output_size=H, include_top=False, compute_dtype=torch.float32)
        bb.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in p.items()})
        head = DINOHead(H, OUT, nlayers=3, hidden_dim=48, bottleneck_dim=16)
        head.load_state_dict(head_sd)
        return MultiCropWrapper(bb, head).to(cuda)

    student, teacher = build(), build()
    crit = DINOLoss(OUT, 6, 0.04, 0.07, 3, 10).to(cuda)
    views = [torch.from_numpy(g[f"view{i}"]).to(cuda) for i in range(6)]
    with torch.no_grad():
        teacher_outputs = torch.stack([teacher(v) for v in views[:2]], dim=0)
    student_outputs = torch.stack([student(v) for v in views], dim=0)
    loss = crit(student_outputs, teacher_outputs, 0)
    loss.backward()
    torch.cuda.synchronize()
    np.testing.assert_allclose(student_outputs.detach().cpu().numpy(), g["student_out"], atol=2e-5)
    assert abs(loss.item() - float(g["loss"])) < 1e-5
    np.testing.assert_allclose(crit.center.cpu().numpy(), g["center"], atol=1e-6)
    checked = 0
    for name, par in student.named_parameters():
        key = "grad__" + name
        if key in g.files:
            want = g[key]
            np.testing.assert_allclose(par.grad.cpu().numpy(), want, atol=1e-4 * max(1e-3, np.abs(want).max()), err_msg=name)
            checked += 1
    assert checked >= 14


def test_barlow_step_at_cfg5_width(cuda):
    """BASELINE.json configs[4] on one GPU at its real width: the Barlow-Twins loss (EEG-BarlowNetworks/net.py:33-42) on
    the cfg2 encoder's embeddings -- B 256, D 384, 128 x 500 segments through the band-pass, H 768 x 2 layers, bf16 --
    with the HIP off-diagonal reduction on the 384 x 384 cross-correlation matrix and one LARS step (optim.py:17-44).
    The loss is held to the oracle's on the same embeddings, the reduction to the oracle's on the same c, the step to the
    oracle's LARS on the device's gradients (the optimizer itself is pinned to the executed reference in the CPU suite)."""
    from cerebralsignalnetworks_amd.trainer import DistillTrainer
    B, C, T, H, L, D = 256, 128, 500, 768, 2, 384
    x = torch.from_numpy(eeg_filter.synthetic_eeg(B, C, T, seed=51)).to(cuda)
    tgt_np = np.random.default_rng(52).standard_normal((B, D)).astype(np.float32)
    tgt = torch.from_numpy(tgt_np).to(cuda)
    torch.manual_seed(43)
    m = Model(input_size=C, lstm_size=H, lstm_layers=L, output_size=D, include_top=False).to(cuda)
    lr = 0.2
    tr = DistillTrainer(m, EEGFilters(1000, order=3).sos, loss="barlow", optimizer="lars", lr=lr)
    before = {n: p.detach().clone() for n, p in m.named_parameters()}
    with torch.no_grad():
        feat = m(tr.embed(x)).float()
    loss = tr.train_step(x, tgt)
    tr.check_device_status()
    want, c = losses.barlow_loss(feat.double().cpu().numpy(), tgt_np, B)
    assert abs(loss.item() - want) < 1e-4 * want, (loss.item(), want)
    # K7 on the full-width matrix
    got = cabi.barlow_offdiag_sqsum(torch.from_numpy(c.astype(np.float32)).to(cuda)).cpu().numpy()
    c32 = c.astype(np.float32).astype(np.float64)
    np.testing.assert_allclose(got, [((np.diagonal(c32) - 1.0) ** 2).sum(), losses.off_diagonal_sqsum(c32)], rtol=2e-6)
    # one LARS step from zero momentum on the gradients the device produced (they are still in the flat buffer)
    for n, p in m.named_parameters():
        g = p.grad.double().cpu().numpy()
        new_p, _ = losses.lars_step(before[n].double().cpu().numpy(), g, np.zeros_like(g), lr, weight_decay=1e-6,
                                    weight_decay_filter=True, lars_adaptation_filter=True)
        np.testing.assert_allclose(p.detach().cpu().numpy(), new_p, rtol=0, atol=2e-6 * max(1.0, float(np.abs(new_p).max())), err_msg=n)
        assert np.isfinite(g).all() and np.abs(g).max() > 0, n


def test_fused_flat_rmsprop_matches_torch(cuda):
    """csn_rmsprop_step over the flat buffers = torch.optim.RMSprop(params, lr) (LstmDistillFromDinoV2Train.py:329)
    for five steps, and the parameters the model sees ARE the flat buffer (views)."""
    from cerebralsignalnetworks_amd.trainer import FlatGrads, FlatRMSprop
    torch.manual_seed(3)
    net = torch.nn.Sequential(torch.nn.Linear(37, 53), torch.nn.Tanh(), torch.nn.Linear(53, 11)).to(cuda)
    ref = torch.nn.Sequential(torch.nn.Linear(37, 53), torch.nn.Tanh(), torch.nn.Linear(53, 11)).to(cuda)
    ref.load_state_dict(net.state_dict())
    fg = FlatGrads(net.parameters(), flatten_params=True)
    opt = FlatRMSprop(fg, lr=1e-3)
    ropt = torch.optim.RMSprop(ref.parameters(), lr=1e-3)
    x = torch.randn(29, 37, device=cuda)
    for step in range(5):
        fg.zero()
        net(x).pow(2).mean().backward()
        opt.step()
        ropt.zero_grad()
        ref(x).pow(2).mean().backward()
        ropt.step()
    for (n, p), q in zip(net.named_parameters(), ref.parameters()):
        assert p.data_ptr() >= fg.flat_params.data_ptr() and p.data_ptr() < fg.flat_params.data_ptr() + fg.flat_params.numel() * 4
        np.testing.assert_allclose(p.detach().cpu().numpy(), q.detach().cpu().numpy(), rtol=2e-6, atol=1e-7, err_msg=n)
    sd = opt.state_dict()
    opt2 = FlatRMSprop(fg, lr=5e-4)
    opt2.load_state_dict(sd)
    assert opt2.param_groups[0]["lr"] == 1e-3 and torch.equal(opt2.square_avg, opt.square_avg)


def test_bench_line_schema(cuda):
    """bench.py's contract with the driver: ONE JSON line on stdout with the metric of BASELINE.json, the roofline
    object of the dominant kernel (measured with HIP events inside the run) and no model keys in `config`."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1",
                          "--no-cpu-baseline", "--no-retrieval", "--no-f32-line", "--no-parity"],
                         capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["unit"] == "segments/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "bf16" and d["data"] == "synthetic"
    assert d["config"]["workload"].startswith("cfg2") and "model" not in d["config"]
    assert abs(d["value"] - 256 * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.0 < r["frac"] < 1.0
    assert r["us_per_launch"]["fwd"] > 0 and r["us_per_launch"]["bwd"] > 0
