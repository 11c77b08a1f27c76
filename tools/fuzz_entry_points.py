"""Randomised odd shapes through every stateless C-ABI entry point, against numpy / scipy / torch on the host (a bug
hunt, not a test: prints one line per case and a summary; exit code 1 if anything fails or raises).
    python tools/fuzz_entry_points.py [cases per entry point]"""
import os
import sys
import traceback

import numpy as np
import scipy.signal
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cerebralsignalnetworks_amd import cabi, EEGFilters      # noqa: E402

dev = torch.device("cuda:0")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(7)
bad = []


def case(name, fn):
    try:
        err, tol = fn()
        ok = np.isfinite(err) and err <= tol
        print(f"{'ok  ' if ok else 'FAIL'} {name}: err {err:.3g} (tol {tol:g})", flush=True)
        if not ok:
            bad.append(name)
    except Exception as e:      # noqa: BLE001
        print(f"RAISE {name}: {type(e).__name__}: {str(e)[:300]}", flush=True)
        traceback.print_exc(limit=2)
        bad.append(name)


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


# ---- band-pass + z-score ---------------------------------------------------------------------
for i in range(n_cases):
    B, C, T = int(rng.integers(1, 9)), int(rng.integers(1, 140)), int(rng.integers(8, 700))
    order, ddof, tm = int(rng.integers(1, 6)), int(rng.integers(0, 2)), bool(rng.integers(0, 2))

    def f(B=B, C=C, T=T, order=order, ddof=ddof, tm=tm):
        x = rng.standard_normal((B, C, T)).astype(np.float32)
        sos = scipy.signal.butter(order, [14.0, 70.0], btype='band', fs=1000, output='sos')
        y = scipy.signal.sosfilt(sos, x.astype(np.float64), axis=-1)
        ref = (y - y.mean(-1, keepdims=True)) / y.std(-1, ddof=ddof, keepdims=True)
        out = cabi.eeg_bandpass_znorm(t(x), sos, ddof=ddof, time_major=tm).cpu().numpy()      # [B,T,C] or [T,B,C]
        out = out.transpose(1, 2, 0) if tm else out.transpose(0, 2, 1)
        return float(np.abs(out - ref).max()), 5e-6
    case(f"bandpass B{B} C{C} T{T} order{order} ddof{ddof} tm{int(tm)}", f)

# ---- zero-phase filtfilt -----------------------------------------------------------------------
for i in range(n_cases):
    S, T, C = int(rng.integers(1, 6)), int(rng.integers(40, 600)), int(rng.integers(1, 130))
    order = int(rng.integers(1, 5))

    def f(S=S, T=T, C=C, order=order):
        x = rng.standard_normal((S, T, C)).astype(np.float32)
        sos = scipy.signal.butter(order, [14.0, 70.0], btype='band', fs=1000, output='sos')
        ref = scipy.signal.sosfiltfilt(sos, x.astype(np.float64), axis=1)
        out = cabi.eeg_filtfilt(t(x), sos).cpu().numpy()
        return float(np.abs(out - ref).max() / max(1e-9, np.abs(ref).max())), 2e-5
    case(f"filtfilt S{S} T{T} C{C} order{order}", f)

# ---- GEMMs --------------------------------------------------------------------------------------
for i in range(n_cases):
    M, N, K = int(rng.integers(1, 700)), int(rng.integers(1, 700)), int(rng.integers(1, 900))
    for dt, tol in ((torch.float32, 2e-5), (torch.bfloat16, 2e-2)):
        def f(M=M, N=N, K=K, dt=dt, tol=tol):
            a = t(rng.standard_normal((M, K)).astype(np.float32)).to(dt)
            b = t(rng.standard_normal((N, K)).astype(np.float32)).to(dt)
            bias = t(rng.standard_normal(N).astype(np.float32))
            ref = a.double() @ b.double().t() + bias.double()
            out = cabi.gemm_nt(a, b, bias).double()
            return float((out - ref).norm() / ref.norm()), tol
        case(f"gemm_nt {M}x{N}x{K} {str(dt)[6:]}", f)

        def g(M=M, N=N, K=K, dt=dt, tol=tol):
            a = t(rng.standard_normal((K, M)).astype(np.float32)).to(dt)
            b = t(rng.standard_normal((K, N)).astype(np.float32)).to(dt)
            ref = a.double().t() @ b.double()
            out = cabi.gemm_tn(a, b).double()
            return float((out - ref).norm() / ref.norm()), tol
        case(f"gemm_tn {M}x{N}x{K} {str(dt)[6:]}", g)

# ---- losses, optimiser, reductions ---------------------------------------------------------------
for i in range(n_cases):
    B, D = int(rng.integers(1, 300)), int(rng.integers(1, 800))

    def f(B=B, D=D):
        s = rng.standard_normal((B, D)).astype(np.float32)
        tg = rng.standard_normal((B, D)).astype(np.float32)
        st = torch.tensor(s, dtype=torch.float64, requires_grad=True)
        ref = (1.0 - torch.nn.functional.cosine_similarity(st, torch.tensor(tg, dtype=torch.float64), dim=1)).mean()
        ref.backward()
        loss, grad = cabi.cosine_loss(t(s), t(tg))
        e1 = abs(float(loss) - float(ref))
        e2 = float((grad.double().cpu() - st.grad).abs().max())
        return max(e1, e2), 2e-6
    case(f"cosine_loss B{B} D{D}", f)

    n = int(rng.integers(1, 200000))

    def g(n=n):
        p = rng.standard_normal(n).astype(np.float32)
        gr = rng.standard_normal(n).astype(np.float32)
        sq = np.abs(rng.standard_normal(n)).astype(np.float32)
        pt, gt, st = t(p.copy()), t(gr), t(sq.copy())
        cabi.rmsprop_step(pt, gt, st, lr=1e-3)
        sq2 = 0.99 * sq.astype(np.float64) + 0.01 * gr.astype(np.float64) ** 2
        p2 = p - 1e-3 * gr / (np.sqrt(sq2) + 1e-8)
        return float(max(np.abs(pt.cpu().numpy() - p2).max(), np.abs(st.cpu().numpy() - sq2).max())), 1e-6
    case(f"rmsprop n{n}", g)

    Dm = int(rng.integers(1, 500))

    def h(Dm=Dm):
        c = rng.standard_normal((Dm, Dm)).astype(np.float32)
        c64 = c.astype(np.float64)
        ref = np.array([((np.diag(c64) - 1.0) ** 2).sum(), (c64 ** 2).sum() - (np.diag(c64) ** 2).sum()])      # csn_hip.h: out[0], out[1]
        out = cabi.barlow_offdiag_sqsum(t(c)).double().cpu().numpy()
        return float((np.abs(out - ref) / np.maximum(1.0, np.abs(ref))).max()), 1e-5
    case(f"barlow_offdiag D{Dm}", h)

# ---- retrieval ---------------------------------------------------------------------------------
for i in range(n_cases):
    Ng, Nq, D = int(rng.integers(1, 3000)), int(rng.integers(1, 300)), int(rng.integers(1, 500))
    k = int(rng.integers(1, min(Ng, 10) + 1))

    def f(Ng=Ng, Nq=Nq, D=D, k=k):
        g = rng.standard_normal((Ng, D)).astype(np.float32)
        q = rng.standard_normal((Nq, D)).astype(np.float32)
        d2 = ((q.astype(np.float64)[:, None, :] - g.astype(np.float64)[None, :, :]) ** 2).sum(-1) if Ng * Nq * D < 4e7 else (
            (q.astype(np.float64) ** 2).sum(1)[:, None] + (g.astype(np.float64) ** 2).sum(1)[None, :] - 2 * q.astype(np.float64) @ g.astype(np.float64).T)
        ref = np.argsort(d2, axis=1, kind="stable")[:, :k]
        dist, idx = cabi.l2_topk(t(g), t(q), k)
        idx = idx.cpu().numpy()
        # (random data: ties have probability zero; near-ties within f32 rounding are accepted by distance)
        mism = idx != ref
        if mism.any():
            rows = np.nonzero(mism.any(1))[0]
            gap = max(abs(d2[r, idx[r, c]] - d2[r, ref[r, c]]) / max(1e-9, d2[r, ref[r, c]]) for r in rows for c in np.nonzero(mism[r])[0])
            return float(gap), 1e-5
        return 0.0, 1e-5
    case(f"l2_topk Ng{Ng} Nq{Nq} D{D} k{k}", f)

print(f"{len(bad)} failing case(s)" + (": " + "; ".join(bad[:20]) if bad else ""))
sys.exit(1 if bad else 0)
