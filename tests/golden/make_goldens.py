"""Generates the golden fixtures under tests/golden/ (run once, in the build container).

The reference has no tests or fixtures.  Goldens therefore come from
  (a) the reference modules that import here: utils/EEGFilters.py (band edges),
      EEG-BarlowNetworks/optim.py (LARS), utils/utils.py (cosine_scheduler);
  (b) the third-party calls the reference makes on this path, run directly:
      scipy.signal.butter / sosfilt / filtfilt, torch.nn.LSTM (CPU), nn.CosineSimilarity,
      F.cross_entropy, nn.KLDivLoss, nn.BatchNorm1d.
Only arrays are written (inputs + expected outputs); nothing of the reference's text.
/root/reference is read here and never at test time.

    python tests/golden/make_goldens.py
"""
import importlib.util
import os
import sys

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from scipy import signal

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, os.path.join(HERE, "..", ".."))


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def g1_filter_design():
    out = {}
    ef = _load(os.path.join(REF, "utils", "EEGFilters.py"), "ref_eegfilters")
    for fs in (1000, 2048):
        f = ef.EEGFilters(fs)
        out[f"band_norm_fs{fs}"] = np.array([f.low_cutoff_norm, f.high_cutoff_norm])
        out[f"band_hz_fs{fs}"] = np.array([f.low_cutoff, f.high_cutoff])
        for order in (3, 4, 5):
            wn = [f.low_cutoff_norm, f.high_cutoff_norm]
            out[f"sos_fs{fs}_o{order}"] = signal.butter(order, wn, btype="bandpass", output="sos")
            b, a = signal.butter(order, wn, btype="bandpass")
            out[f"b_fs{fs}_o{order}"] = b
            out[f"a_fs{fs}_o{order}"] = a
    np.savez_compressed(os.path.join(HERE, "filter_design.npz"), **out)


def g2_filter_apply():
    rng = np.random.default_rng(0)
    t = np.arange(500) / 1000.0
    x = (rng.standard_normal((2, 16, 500)) + 0.5 * np.sin(2 * np.pi * 40 * t)).astype(np.float32)
    out = {"x": x}
    for order in (3, 4, 5):
        sos = signal.butter(order, [0.1 / 500, 60.0 / 500], btype="bandpass", output="sos")
        y = signal.sosfilt(sos, x.astype(np.float64), axis=-1)
        out[f"sosfilt_o{order}"] = y
        for ddof in (0, 1):
            z = (y - y.mean(-1, keepdims=True)) / y.std(-1, ddof=ddof, keepdims=True)
            out[f"znorm_o{order}_ddof{ddof}"] = z
    # zero-phase variant the reference applies (Utilities.py:411-428): [S,T,C]
    xs = rng.standard_normal((2, 500, 8))
    b, a = signal.butter(4, [1.0 / 500, 50.0 / 500], btype="band")
    ff = np.zeros_like(xs)
    for s in range(xs.shape[0]):
        for c in range(xs.shape[2]):
            ff[s, :, c] = signal.filtfilt(b, a, xs[s, :, c])
    out["filtfilt_x"] = xs
    out["filtfilt_y"] = ff
    np.savez_compressed(os.path.join(HERE, "filter_apply.npz"), **out)


def _torch_model(C, H, L, D, ncls, params, dtype):
    class M(nn.Module):
        def __init__(self):
            super().__init__()
            self.lstm = nn.LSTM(C, H, num_layers=L, batch_first=True)
            self.fc = nn.Linear(H, D)
            if ncls:
                self.class_pred = nn.Linear(D, ncls)

        def forward(self, x):
            y = self.fc(self.lstm(x)[0][:, -1, :])
            return (y, self.class_pred(y)) if ncls else y
    m = M().to(dtype)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)).to(dtype) for k, v in params.items()})
    return m


def g3_lstm():
    from oracle.lstm import init_params
    B, T, C, H, L, D, NC = 4, 32, 128, 64, 2, 48, 40
    params = init_params(C, H, L, D, NC, seed=43)
    rng = np.random.default_rng(7)
    x = rng.standard_normal((B, T, C)).astype(np.float32)
    tgt = rng.standard_normal((B, D)).astype(np.float32)
    out = {"x": x, "target": tgt, "dims": np.array([B, T, C, H, L, D, NC])}
    for name, dt in (("f64", torch.float64), ("f32", torch.float32)):
        m = _torch_model(C, H, L, D, NC, params, dt)
        xt = torch.from_numpy(x).to(dt)
        feat, cls = m(xt)
        loss = 1 - nn.CosineSimilarity()(feat, torch.from_numpy(tgt).to(dt)).mean()
        loss.backward()
        out[f"feat_{name}"] = feat.detach().numpy()
        out[f"cls_{name}"] = cls.detach().numpy()
        out[f"loss_{name}"] = np.array(loss.item())
        for k, p in m.named_parameters():
            if p.grad is not None:
                out[f"grad_{name}__{k}"] = p.grad.numpy()
        # all-timestep output of the LSTM alone
        out[f"yall_{name}"] = m.lstm(xt)[0].detach().numpy()
    for k, v in params.items():
        out["param__" + k] = v
    np.savez_compressed(os.path.join(HERE, "lstm_small.npz"), **out)

    # full-size forward (cfg2 shapes, B=2): params regenerated from the seed at test time
    B, T, C, H, L, D = 2, 500, 128, 768, 2, 384
    params = init_params(C, H, L, D, None, seed=43)
    x = np.random.default_rng(11).standard_normal((B, T, C)).astype(np.float32)
    m = _torch_model(C, H, L, D, None, params, torch.float32)
    with torch.no_grad():
        ylast = m.lstm(torch.from_numpy(x))[0][:, -1, :]
        feat = m.fc(ylast)
    np.savez_compressed(os.path.join(HERE, "lstm_full_fwd.npz"),
                        dims=np.array([B, T, C, H, L, D]), seed_params=np.array(43), seed_x=np.array(11),
                        ylast=ylast.numpy(), feat=feat.numpy())


def g4_losses():
    rng = np.random.default_rng(3)
    s = rng.standard_normal((16, 384)).astype(np.float32)
    t = rng.standard_normal((16, 384)).astype(np.float32)
    cls = rng.standard_normal((16, 40)).astype(np.float32)
    lab = rng.integers(0, 40, 16)
    S, Tt, C, Lb = (torch.from_numpy(a) for a in (s, t, cls, lab))
    S64, T64, C64 = S.double(), Tt.double(), C.double()
    out = {"student": s, "teacher": t, "cls": cls, "labels": lab}
    S64g = S64.clone().requires_grad_(True)
    loss = 1 - nn.CosineSimilarity()(S64g, T64).mean()
    loss.backward()
    out["cosine_loss"] = np.array(loss.item())
    out["cosine_grad"] = S64g.grad.numpy()
    sched = np.concatenate((np.linspace(1.5, 0.22, 50), np.ones(100 - 50) * 0.22))
    out["temp_schedule_100"] = sched
    for ep in (0, 25, 50):
        Tm = sched[ep]
        tl = F.softmax(T64 / Tm, dim=-1)
        sl = F.softmax(S64 / Tm, dim=-1)
        out[f"featdist_ep{ep}"] = np.array((0.5 * F.cross_entropy(C64, Lb) + 0.5 * F.cross_entropy(tl, sl)).item())
    for alpha, temp in ((1.0, 2.0), (0.5, 4.0)):
        kd = nn.KLDivLoss()(F.log_softmax(C64 / temp, dim=1), F.softmax(C64.flip(0) / temp, dim=1)) * (alpha * temp * temp) \
            + F.cross_entropy(C64, Lb) * (1.0 - alpha)
        out[f"kd_a{alpha}_T{temp}"] = np.array(kd.item())
    # Barlow (net.py:33-42) on LSTM-embedding-sized inputs
    z1 = rng.standard_normal((32, 96))
    z2 = z1 + 0.3 * rng.standard_normal((32, 96))
    bn = nn.BatchNorm1d(96, affine=False).double().train()
    c = bn(torch.from_numpy(z1)).T @ bn(torch.from_numpy(z2))
    c.div_(32)
    on = torch.diagonal(c).add(-1).pow(2).sum()
    n = c.shape[0]
    off = c.flatten()[:-1].view(n - 1, n + 1)[:, 1:].flatten().pow(2).sum()
    out.update(barlow_z1=z1, barlow_z2=z2, barlow_c=c.numpy(), barlow_on=np.array(on.item()),
               barlow_off=np.array(off.item()), barlow_loss=np.array((on + 0.0051 * off).item()))
    # LARS: the reference's own optimizer class, imported by path
    lars = _load(os.path.join(REF, "EEG-BarlowNetworks", "optim.py"), "ref_lars")
    w = torch.from_numpy(rng.standard_normal((8, 5))).requires_grad_(True)
    bvec = torch.from_numpy(rng.standard_normal(5)).requires_grad_(True)
    opt = lars.LARS([w, bvec], lr=0.2, weight_decay=1e-3, weight_decay_filter=True, lars_adaptation_filter=True)
    out["lars_w0"], out["lars_b0"] = w.detach().numpy().copy(), bvec.detach().numpy().copy()
    gw, gb = rng.standard_normal((8, 5)), rng.standard_normal(5)
    out["lars_gw"], out["lars_gb"] = gw, gb
    for it in range(2):
        w.grad, bvec.grad = torch.from_numpy(gw).clone(), torch.from_numpy(gb).clone()
        opt.step()
        out[f"lars_w{it + 1}"], out[f"lars_b{it + 1}"] = w.detach().numpy().copy(), bvec.detach().numpy().copy()
    # cosine_scheduler of the vendored DINO utils (utils/utils.py)
    try:
        sys.path.insert(0, REF)
        ru = _load(os.path.join(REF, "utils", "utils.py"), "ref_utils_utils")
        out["cosine_scheduler"] = ru.cosine_scheduler(0.0005, 1e-6, 10, 7, warmup_epochs=2)
    except Exception as e:  # ordinary import error only
        print("cosine_scheduler golden skipped:", repr(e))
    np.savez_compressed(os.path.join(HERE, "losses.npz"), **out)


if __name__ == "__main__":
    torch.manual_seed(0)
    g1_filter_design()
    g2_filter_apply()
    g3_lstm()
    g4_losses()
    for f in sorted(os.listdir(HERE)):
        print(f, os.path.getsize(os.path.join(HERE, f)))
