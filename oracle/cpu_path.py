"""CPU path of the hot loop built from the same third-party calls the reference makes
(test infrastructure only; used by tests and by bench.py's ``cpu_baseline`` leg).

The reference's own trainer cannot run without a GPU (utils/utils.py:487-489
exits) and imports a module that is not in its tree (``models.lstm``,
LstmDistillFromDinoV2Train.py:5), so the CPU baseline is, per BASELINE.md section 3:
``scipy.signal.sosfilt`` + per-channel z-score -> ``torch.nn.LSTM`` (fp32, CPU)
-> last step -> ``nn.Linear`` -> ``1 - CosineSimilarity().mean()`` -> backward
-> ``RMSprop(lr=1e-3).step()`` (optimizer of LstmDistillFromDinoV2Train.py:329).
"""
import os
import time
import numpy as np


def usable_cores():
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:   # cgroup v2
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    try:   # cgroup v1
        quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if quota > 0 and period > 0:
            n = min(n, max(1, quota // period))
    except (OSError, ValueError):
        pass
    if os.environ.get("CSN_CPU_THREADS"):
        n = int(os.environ["CSN_CPU_THREADS"])
    return max(1, min(n, 16))   # 16 = the CPU share of a one-GPU box in this pool


def build_torch_reference_model(input_size, hidden, layers, out_features, n_classes=None, seed=43):
    import torch
    import torch.nn as nn

    class RefLSTM(nn.Module):  # layout of LSTMDistillRetreival.py:85-110 (sequence over time)
        def __init__(self):
            super().__init__()
            self.lstm = nn.LSTM(input_size, hidden, num_layers=layers, batch_first=True)
            self.fc = nn.Linear(hidden, out_features)
            if n_classes:
                self.class_pred = nn.Linear(out_features, n_classes)

        def forward(self, x):
            y = self.fc(self.lstm(x)[0][:, -1, :])
            if n_classes:
                return y, self.class_pred(y)
            return y

    torch.manual_seed(seed)
    return RefLSTM()


def preprocess_scipy(x_bct, sos, ddof=0):
    """sosfilt along time + per-channel z-score, returns float32 [B,T,C]."""
    from scipy.signal import sosfilt
    y = sosfilt(sos, np.asarray(x_bct, np.float64), axis=-1)
    y = (y - y.mean(axis=-1, keepdims=True)) / y.std(axis=-1, ddof=ddof, keepdims=True)
    return np.ascontiguousarray(np.transpose(y, (0, 2, 1))).astype(np.float32)


def time_cpu_train_steps(x_bct, targets, sos, hidden=768, layers=2, steps=2, warmup=1, lr=1e-3, threads=None):
    """Times the CPU path on a bounded sample.  Returns dict(seg_per_s, s_per_step, cores, losses)."""
    import torch
    import torch.nn as nn
    threads = threads or usable_cores()
    torch.set_num_threads(threads)
    B, C, T = x_bct.shape
    model = build_torch_reference_model(C, hidden, layers, targets.shape[1])
    opt = torch.optim.RMSprop(model.parameters(), lr=lr)
    cos = nn.CosineSimilarity()
    tgt = torch.from_numpy(np.asarray(targets, np.float32))
    losses = []
    t_total = 0.0
    for it in range(warmup + steps):
        t0 = time.perf_counter()
        eeg = torch.from_numpy(preprocess_scipy(x_bct, sos))
        opt.zero_grad()
        out = model(eeg)
        loss = 1 - cos(out, tgt).mean()
        loss.backward()
        opt.step()
        dt = time.perf_counter() - t0
        losses.append(float(loss))
        if it >= warmup:
            t_total += dt
    return dict(seg_per_s=B * steps / t_total, s_per_step=t_total / steps,
                cores=int(torch.get_num_threads()), losses=losses)
