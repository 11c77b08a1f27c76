"""DINO self-distillation pieces for the LSTM encoder, mirroring /root/reference/LstmDistillation.py:
``DINOHead`` (:66-99, = dino/vision_transformer.py:257-291), ``MultiCropWrapper`` (:28-64),
``DINOLoss`` (:101-159), ``cosine_scheduler`` (utils/utils.py:187-198), the temporal multi-crop
sampler (:543-565) and the EMA teacher update (:611-615).  Torch ops only -- the LSTM inside the
wrapped backbone is the HIP one.

Reference quirks kept (``compat=True``): the loss chunks the teacher output with ``chunk(1)`` and the
stacked student output with ``chunk(ncrops)``, so both global teacher views are compared with every
student view except view 0; ``update_center`` sums over the VIEW axis, which turns the centre into a
per-sample ``[1,B,out]`` buffer after the first step.
"""
import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F


def cosine_scheduler(base_value, final_value, epochs, niter_per_ep, warmup_epochs=0, start_warmup_value=0):
    warmup_iters = warmup_epochs * niter_per_ep
    warmup = np.linspace(start_warmup_value, base_value, warmup_iters) if warmup_epochs > 0 else np.array([])
    iters = np.arange(epochs * niter_per_ep - warmup_iters)
    schedule = final_value + 0.5 * (base_value - final_value) * (1 + np.cos(np.pi * iters / len(iters)))
    schedule = np.concatenate((warmup, schedule))
    assert len(schedule) == epochs * niter_per_ep
    return schedule


class DINOHead(nn.Module):
    def __init__(self, in_dim, out_dim, use_bn=False, norm_last_layer=True, nlayers=3, hidden_dim=2048,
                 bottleneck_dim=256):
        super().__init__()
        nlayers = max(nlayers, 1)
        if nlayers == 1:
            self.mlp = nn.Linear(in_dim, bottleneck_dim)
        else:
            layers = [nn.Linear(in_dim, hidden_dim)]
            if use_bn:
                layers.append(nn.BatchNorm1d(hidden_dim))
            layers.append(nn.GELU())
            for _ in range(nlayers - 2):
                layers.append(nn.Linear(hidden_dim, hidden_dim))
                if use_bn:
                    layers.append(nn.BatchNorm1d(hidden_dim))
                layers.append(nn.GELU())
            layers.append(nn.Linear(hidden_dim, bottleneck_dim))
            self.mlp = nn.Sequential(*layers)
        self.apply(self._init_weights)
        self.last_layer = nn.utils.weight_norm(nn.Linear(bottleneck_dim, out_dim, bias=False))
        self.last_layer.weight_g.data.fill_(1)
        if norm_last_layer:
            self.last_layer.weight_g.requires_grad = False

    @staticmethod
    def _init_weights(m):
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)

    def forward(self, x):
        x = self.mlp(x)
        x = F.normalize(x, dim=-1, p=2)
        return self.last_layer(x)


class MultiCropWrapper(nn.Module):
    """One backbone forward per run of equal-length views, then the head on the concatenation."""

    def __init__(self, backbone, head):
        super().__init__()
        backbone.fc, backbone.head = nn.Identity(), nn.Identity()      # LstmDistillation.py:40
        self.backbone, self.head = backbone, head

    def forward(self, x):
        if not isinstance(x, list):
            x = [x]
        lengths = torch.tensor([inp.shape[1] for inp in x])            # crops differ in TIME length
        idx_crops = torch.cumsum(torch.unique_consecutive(lengths, return_counts=True)[1], 0)
        start, outs = 0, []
        for end in idx_crops:
            out = self.backbone(torch.cat(x[start:end]))
            outs.append(out[0] if isinstance(out, tuple) else out)
            start = end
        return self.head(torch.cat(outs))


class DINOLoss(nn.Module):
    def __init__(self, out_dim, ncrops, warmup_teacher_temp, teacher_temp, warmup_teacher_temp_epochs, nepochs,
                 student_temp=0.1, center_momentum=0.9):
        super().__init__()
        self.student_temp, self.center_momentum, self.ncrops = student_temp, center_momentum, ncrops
        self.register_buffer("center", torch.zeros(1, out_dim))
        self.teacher_temp_schedule = np.concatenate((
            np.linspace(warmup_teacher_temp, teacher_temp, warmup_teacher_temp_epochs),
            np.ones(max(0, nepochs - warmup_teacher_temp_epochs)) * teacher_temp))

    def forward(self, student_output, teacher_output, epoch):
        student_out = (student_output / self.student_temp).chunk(self.ncrops)
        temp = self.teacher_temp_schedule[epoch]
        teacher_out = F.softmax((teacher_output - self.center) / temp, dim=-1).detach().chunk(1)
        total_loss, n_loss_terms = 0, 0
        for iq, q in enumerate(teacher_out):
            for v in range(len(student_out)):
                if v == iq:
                    continue
                loss = torch.sum(-q * F.log_softmax(student_out[v], dim=-1), dim=-1)
                total_loss = total_loss + loss.mean()
                n_loss_terms += 1
        total_loss = total_loss / n_loss_terms
        self.update_center(teacher_output)
        return total_loss

    @torch.no_grad()
    def update_center(self, teacher_output):
        batch_center = torch.sum(teacher_output, dim=0, keepdim=True)
        world = 1
        if dist.is_available() and dist.is_initialized():
            dist.all_reduce(batch_center)
            world = dist.get_world_size()
        batch_center = batch_center / (len(teacher_output) * world)
        self.center = self.center * self.center_momentum + batch_center * (1 - self.center_momentum)


def temporal_crops(eeg_btc, n_global=2, n_local=4, global_len=300, local_len=200, rng=None):
    """LstmDistillation.py:543-565: random start per view (shared by the batch), shifted left when the
    window would run past the end."""
    rng = rng or np.random
    T = eeg_btc.size(1)
    views = []
    for n, length in ((n_global, global_len), (n_local, local_len)):
        for _ in range(n):
            start = int(rng.randint(0, T))
            end = start + length
            if end > T:
                start -= end - T
                end = start + length
            views.append(eeg_btc[:, start:end, :])
    return views[:n_global], views[n_global:]


@torch.no_grad()
def ema_update(student, teacher, m):
    for pq, pk in zip(student.parameters(), teacher.parameters()):
        pk.data.mul_(m).add_((1 - m) * pq.detach().data)
