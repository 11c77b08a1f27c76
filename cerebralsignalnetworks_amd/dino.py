"""Self-distillation (DINO) pieces for the LSTM encoder -- the trainer of /root/reference/LstmDistillation.py.

What the reference computes, and where:
  projection head            LstmDistillation.py:66-99   MLP -> L2 normalise -> weight-normalised linear prototypes
  multi-view wrapper         LstmDistillation.py:28-64   one backbone pass per run of equal-length views
  loss + centre              LstmDistillation.py:101-159 cross-entropy between sharpened teacher and student softmaxes
  schedules                  utils/utils.py:187-198      linear warm-up then half cosine
  temporal multi-crop        LstmDistillation.py:543-565 2 x 300-sample + 4 x 200-sample windows of the segment
  EMA teacher                LstmDistillation.py:611-615

Written from that math with its own structure (the LSTM inside the wrapped backbone is the HIP one; everything
here is head-sized torch work): the loss is ONE pass -- a single log-softmax over all student views, the teacher
distribution once, and the pairwise sum collapsed algebraically (sum_v q . log p_v = q . sum_v log p_v) instead
of a Python double loop of small kernels.  Module / buffer names follow the reference so its checkpoints load.
Every function is checked against values obtained by executing the reference's definitions
(tests/golden/ref_losses.npz).

Reference behaviour kept under ``compat=True`` (the default, what LstmDistillation.py does): the views arrive
STACKED -- student [V, B, out], teacher [2, B, out] (:583-586) -- and the loss chunks the teacher into ONE chunk
(``chunk(1)``, :128), so both teacher views are compared with every student view except view 0, and the centre
update sums over the view axis, turning the centre into a per-sample [1, B, out] buffer.  ``compat=False`` is
the pairing of the DINO paper (each teacher view against every *other* student view, centre [1, out]).
"""
import itertools

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F


def cosine_scheduler(base_value, final_value, epochs, niter_per_ep, warmup_epochs=0, start_warmup_value=0):
    """Per-iteration schedule: ``warmup_epochs`` of linear ramp start_warmup_value -> base_value (end point
    included), then a half cosine base_value -> final_value over the remaining iterations."""
    total, warm = epochs * niter_per_ep, warmup_epochs * niter_per_ep
    it = np.arange(total, dtype=np.float64)
    ramp = start_warmup_value + (base_value - start_warmup_value) * it / max(warm - 1, 1)
    decay = final_value + 0.5 * (base_value - final_value) * (1 + np.cos(np.pi * (it - warm) / max(total - warm, 1)))
    return np.where(it < warm, ramp, decay)


class DINOHead(nn.Module):
    """in_dim -> (hidden_dim, GELU) x (nlayers - 1) -> bottleneck_dim -> unit sphere -> out_dim prototypes whose
    weight is stored as direction ``weight_v`` and magnitude ``weight_g`` (fixed at 1 with norm_last_layer)."""

    def __init__(self, in_dim, out_dim, use_bn=False, norm_last_layer=True, nlayers=3, hidden_dim=2048,
                 bottleneck_dim=256):
        super().__init__()
        widths = [in_dim] + [hidden_dim] * (max(nlayers, 1) - 1) + [bottleneck_dim]
        stack = []
        for k, (fan_in, fan_out) in enumerate(zip(widths[:-1], widths[1:])):
            stack.append(nn.Linear(fan_in, fan_out))
            if k + 2 < len(widths):                       # every layer but the last: [BatchNorm,] GELU
                stack += ([nn.BatchNorm1d(fan_out)] if use_bn else []) + [nn.GELU()]
        self.mlp = stack[0] if len(stack) == 1 else nn.Sequential(*stack)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=.02)
                nn.init.zeros_(m.bias)
        self.last_layer = nn.utils.weight_norm(nn.Linear(bottleneck_dim, out_dim, bias=False))
        self.last_layer.weight_g.data.fill_(1)
        self.last_layer.weight_g.requires_grad = not norm_last_layer

    def forward(self, x):
        return self.last_layer(F.normalize(self.mlp(x), dim=-1, p=2))


class MultiCropWrapper(nn.Module):
    """Backbone + head over a list of views [B, T_v, C]: consecutive views of equal length share one backbone
    pass (the LSTM takes any T, but a batch needs one T); the head runs once on all rows."""

    def __init__(self, backbone, head):
        super().__init__()
        backbone.fc, backbone.head = nn.Identity(), nn.Identity()
        self.backbone, self.head = backbone, head

    def forward(self, x):
        views = x if isinstance(x, list) else [x]
        rows = []
        for _, run in itertools.groupby(views, key=lambda v: v.shape[1]):
            out = self.backbone(torch.cat(list(run)))
            rows.append(out[0] if isinstance(out, tuple) else out)
        return self.head(torch.cat(rows))


class DINOLoss(nn.Module):
    def __init__(self, out_dim, ncrops, warmup_teacher_temp, teacher_temp, warmup_teacher_temp_epochs, nepochs,
                 student_temp=0.1, center_momentum=0.9, compat=True):
        super().__init__()
        self.student_temp, self.center_momentum, self.ncrops, self.compat = student_temp, center_momentum, ncrops, compat
        self.register_buffer("center", torch.zeros(1, out_dim))
        ramp = np.linspace(warmup_teacher_temp, teacher_temp, warmup_teacher_temp_epochs)
        self.teacher_temp_schedule = np.concatenate((ramp, np.full(max(0, nepochs - warmup_teacher_temp_epochs), teacher_temp)))

    def forward(self, student_output, teacher_output, epoch):
        temp = self.teacher_temp_schedule[epoch]
        q = F.softmax((teacher_output - self.center) / temp, dim=-1).detach()
        log_p = F.log_softmax(student_output / self.student_temp, dim=-1)
        if self.compat:
            # stacked views: q [2, B, out] against student views 1 .. V-1, each term a mean over (2, B):
            #   mean_v mean_{g,b} -sum_o q[g,b,o] log_p[v,b,o]  =  -mean_{g,b} sum_o q[g,b,o] S[b,o] / (V - 1)
            v = log_p.shape[0]
            loss = -(q * log_p[1:].sum(dim=0, keepdim=True)).sum(dim=-1).mean() / (v - 1)
        else:
            # DINO pairing on row-concatenated views: teacher view g against every student view v != g
            g_views, s_views = q.chunk(2), log_p.chunk(self.ncrops)
            total = sum(s_views)
            loss = sum(-(g_views[g] * (total - s_views[g])).sum(dim=-1).mean() for g in range(2)) / (2 * (self.ncrops - 1))
        self.update_center(teacher_output)
        return loss

    @torch.no_grad()
    def update_center(self, teacher_output):
        batch_center = teacher_output.sum(dim=0, keepdim=True)
        world = 1
        if dist.is_available() and dist.is_initialized():
            dist.all_reduce(batch_center)
            world = dist.get_world_size()
        batch_center /= len(teacher_output) * world
        self.center = self.center * self.center_momentum + batch_center * (1 - self.center_momentum)


def temporal_crops(eeg_btc, n_global=2, n_local=4, global_len=300, local_len=200, rng=None):
    """Multi-crop in time: per view one random start shared by the batch (drawn from [0, T)), moved left so the
    window ends inside the segment."""
    rng = rng or np.random
    T = eeg_btc.size(1)
    lengths = [global_len] * n_global + [local_len] * n_local
    views = []
    for length in lengths:
        start = min(int(rng.randint(0, T)), T - length)
        views.append(eeg_btc[:, start:start + length, :])
    return views[:n_global], views[n_global:]


@torch.no_grad()
def ema_update(student, teacher, m):
    """teacher <- m teacher + (1 - m) student, all parameters in two fused calls."""
    tp = [p.data for p in teacher.parameters()]
    sp = [p.detach().data for p in student.parameters()]
    torch._foreach_mul_(tp, m)
    torch._foreach_add_(tp, sp, alpha=1 - m)
