"""Losses of the distillation path, same names / argument meaning as the reference classes.

  CosineSimilarityLoss       LstmDistillFromDinoV2Train.py:36-43   (fused HIP forward+gradient)
  FeatureDistributionLoss    LstmDistillFromDinoV2Train.py:107-140 (torch ops; not hot)
  loss_fn_kd                 LstmDistillFromDinoV2TrainSpampinato.py:107-121
  FeatureDistributionLossKD  LstmDistillFromDinoV2TrainSpampinato.py:125-184 (soft-target KL + CE)
  FeatureDistributionLossSoft LstmDistillFromDinoV2Eval.py:106-146 (soft-target KL alone)
  FeatureDistributionLossMSE LstmDistillation.py:161-172 (global mean / std / MSE match)
  BarlowTwinsLoss            EEG-BarlowNetworks/net.py:33-42 (HIP off-diagonal reduction)
  LARS, barlow_learning_rate EEG-BarlowNetworks/optim.py:5-44, barlow_utils.py:8-21
Every class is checked against values and gradients obtained by executing the reference's own definition
(tests/golden/ref_losses.npz, tests/test_ref_pinned.py).
Reference quirks are kept on purpose.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import cabi


class HyperParams:          # LstmDistillFromDinoV2Train.py:16-25
    learning_rate = 0.001
    T = 0.5
    soft_target_loss_weight = 0.25
    ce_loss_weight = 0.75
    warmup_teacher_temp = 1.5
    teacher_temp = 0.22
    warmup_teacher_temp_epochs = 50
    alpha = 0.5
    beta = 0.5


class _CosineLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, student, teacher):
        loss, ds = cabi.cosine_loss(student, teacher, want_grad=student.requires_grad)
        ctx.save_for_backward(ds) if ds is not None else None
        ctx.has_grad = ds is not None
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        if not ctx.has_grad:
            return None, None
        (ds,) = ctx.saved_tensors
        return ds * g, None


class CosineSimilarityLoss(nn.Module):
    def __init__(self):
        super().__init__()

    def forward(self, student_outputs, teacher_outputs):
        return _CosineLossFn.apply(student_outputs, teacher_outputs)


class FeatureDistributionLoss(nn.Module):
    def __init__(self, nepochs, warmup_teacher_temp, teacher_temp, warmup_teacher_temp_epochs):
        super().__init__()
        self.mse = nn.MSELoss()
        self.teacher_temp_schedule = np.concatenate((
            np.linspace(warmup_teacher_temp, teacher_temp, warmup_teacher_temp_epochs),
            np.ones(max(0, nepochs - warmup_teacher_temp_epochs)) * teacher_temp))

    def forward(self, student_outputs, teacher_outputs, epoch, label, pred_label=None):
        HyperParams.T = self.teacher_temp_schedule[epoch]
        teacher_logits_with_T = F.softmax(teacher_outputs / HyperParams.T, dim=-1)
        student_logits_with_T = F.softmax(student_outputs / HyperParams.T, dim=-1)
        term1 = HyperParams.alpha * F.cross_entropy(pred_label, label)
        # reference: probabilities of the teacher used as *logits*, student probabilities as targets
        term2 = HyperParams.beta * F.cross_entropy(teacher_logits_with_T, student_logits_with_T)
        return term1 + term2


def _soft_target_kl(student_outputs, teacher_outputs, T):
    """sum p_t (log p_t - log_softmax(s / T)) / B * T^2 -- the soft-target term both variants below share."""
    soft_targets = F.softmax(teacher_outputs / T, dim=-1)
    soft_prob = F.log_softmax(student_outputs / T, dim=-1)
    return torch.sum(soft_targets * (soft_targets.log() - soft_prob)) / soft_prob.size()[0] * (T ** 2)


class FeatureDistributionLossKD(FeatureDistributionLoss):
    """The Spampinato trainer's variant (LstmDistillFromDinoV2TrainSpampinato.py:125-184; its HyperParams :16-25
    give the 0.25 / 0.75 weights and a 1.65 -> 0.22 temperature ramp over 50 epochs): student outputs are class logits."""
    SCHEDULE = dict(warmup_teacher_temp=1.65, teacher_temp=0.22, warmup_teacher_temp_epochs=50)

    def forward(self, student_outputs, teacher_outputs, epoch, label):
        HyperParams.T = self.teacher_temp_schedule[epoch]
        return HyperParams.soft_target_loss_weight * _soft_target_kl(student_outputs, teacher_outputs, HyperParams.T) \
            + HyperParams.ce_loss_weight * F.cross_entropy(student_outputs, label)


class FeatureDistributionLossSoft(FeatureDistributionLoss):
    """The Eval script's variant (LstmDistillFromDinoV2Eval.py:106-146; temperatures of its HyperParams :18-25:
    1.7 -> 0.23 over 50 epochs): the soft-target term alone."""
    SCHEDULE = dict(warmup_teacher_temp=1.7, teacher_temp=0.23, warmup_teacher_temp_epochs=50)

    def forward(self, student_outputs, teacher_outputs, epoch):
        HyperParams.T = self.teacher_temp_schedule[epoch]
        return _soft_target_kl(student_outputs, teacher_outputs, HyperParams.T)


class FeatureDistributionLossMSE(nn.Module):
    """LstmDistillation.py:161-172: 0.4 (std_s - std_t)^2 + 0.4 (mean_s - mean_t)^2 + 0.2 MSE, global statistics."""

    def forward(self, student_outputs, teacher_outputs):
        d_mean = student_outputs.mean() - teacher_outputs.mean()
        d_std = student_outputs.std() - teacher_outputs.std()
        return 0.4 * d_std * d_std + 0.4 * d_mean * d_mean + 0.2 * F.mse_loss(student_outputs, teacher_outputs)


def loss_fn_kd(outputs, labels, teacher_outputs, params):
    alpha, T = params.alpha, params.temperature
    return nn.KLDivLoss()(F.log_softmax(outputs / T, dim=1), F.softmax(teacher_outputs / T, dim=1)) * (alpha * T * T) \
        + F.cross_entropy(outputs, labels) * (1. - alpha)


class _BarlowReduce(torch.autograd.Function):
    """(sum_i (c_ii-1)^2, sum_{i!=j} c_ij^2) by the HIP reduction; gradient 2(c-I) on / 2c off the diagonal."""

    @staticmethod
    def forward(ctx, c):
        ctx.save_for_backward(c)
        out = cabi.barlow_offdiag_sqsum(c)
        return out[0], out[1]

    @staticmethod
    def backward(ctx, g_on, g_off):
        (c,) = ctx.saved_tensors
        eye = torch.eye(c.shape[0], device=c.device, dtype=c.dtype)
        return (2 * (c - eye)) * eye * g_on + (2 * c) * (1 - eye) * g_off


def _barlow_reduce_torch(c):
    """The same two sums in torch ops: for CPU tensors only (host-logic tests of the multi-rank plumbing over gloo;
    a device tensor always takes the HIP reduction)."""
    d = torch.diagonal(c)
    on = (d - 1).pow(2).sum()
    return on, c.pow(2).sum() - d.pow(2).sum()


class _AllReduceSum(torch.autograd.Function):
    """c <- sum over ranks of c.  The reference all-reduces c in place, outside autograd (net.py:38), so each rank
    back-propagates d loss(c_global) / d c through its OWN term only and DDP then averages the parameter
    gradients: the backward here is the identity (the gradient all-reduce of the trainer does the averaging)."""

    @staticmethod
    def forward(ctx, x):
        x = x.clone()
        torch.distributed.all_reduce(x)
        return x

    @staticmethod
    def backward(ctx, g):
        return g


class BarlowTwinsLoss(nn.Module):
    """net.py:33-42 on two embedding batches: bn(z1).T @ bn(z2) / batch_size, all_reduce, on + lambd*off.
    ``batch_size`` is the GLOBAL batch (args.batch_size there); BatchNorm statistics stay per rank, as there."""

    def __init__(self, dim, batch_size, lambd=0.0051):
        super().__init__()
        self.bn = nn.BatchNorm1d(dim, affine=False)
        self.batch_size, self.lambd = batch_size, lambd

    def forward(self, z1, z2):
        c = self.bn(z1).T @ self.bn(z2)
        c = c / self.batch_size
        if torch.distributed.is_available() and torch.distributed.is_initialized() \
                and torch.distributed.get_world_size() > 1:
            c = _AllReduceSum.apply(c)
        on_diag, off_diag = _BarlowReduce.apply(c.float()) if c.is_cuda else _barlow_reduce_torch(c)
        return on_diag + self.lambd * off_diag


class LARS(torch.optim.Optimizer):
    """Layer-wise adaptive rate scaling with the interface and semantics of the reference's optimizer
    (/root/reference/EEG-BarlowNetworks/optim.py:5-44): per parameter
        d = grad (+ weight_decay * p),   d *= eta |p| / |d|  (when both norms are positive),
        mu = momentum * mu + d,          p -= lr * mu,
    where the two ``*_filter`` switches exempt 1-D parameters (biases, norm scales) from the decay / the scaling.
    Written as a multi-tensor step: one fused ``_foreach`` call per phase over all parameters of a group instead
    of a Python loop of small kernels per parameter (the cfg2 model has 10 tensors, a DINO student 20+)."""

    def __init__(self, params, lr, weight_decay=0, momentum=0.9, eta=0.001, weight_decay_filter=False,
                 lars_adaptation_filter=False):
        super().__init__(params, dict(lr=lr, weight_decay=weight_decay, momentum=momentum, eta=eta,
                                      weight_decay_filter=weight_decay_filter,
                                      lars_adaptation_filter=lars_adaptation_filter))

    @staticmethod
    def exclude_bias_and_norm(p):
        return p.ndim == 1

    @torch.no_grad()
    def step(self):
        for group in self.param_groups:
            params = [p for p in group["params"] if p.grad is not None]
            if not params:
                continue
            one_d = [self.exclude_bias_and_norm(p) for p in params]
            updates = [p.grad.clone() for p in params]
            decayed = [i for i, flat in enumerate(one_d) if not (group["weight_decay_filter"] and flat)]
            if decayed and group["weight_decay"] != 0:
                torch._foreach_add_([updates[i] for i in decayed], [params[i] for i in decayed], alpha=group["weight_decay"])
            scaled = [i for i, flat in enumerate(one_d) if not (group["lars_adaptation_filter"] and flat)]
            if scaled:
                p_norm = torch.stack(torch._foreach_norm([params[i] for i in scaled]))
                u_norm = torch.stack(torch._foreach_norm([updates[i] for i in scaled]))
                trust = torch.where((p_norm > 0) & (u_norm > 0), group["eta"] * p_norm / u_norm, torch.ones_like(p_norm))
                torch._foreach_mul_([updates[i] for i in scaled], list(trust.unbind(0)))
            mus = []
            for p in params:
                state = self.state[p]
                if "mu" not in state:
                    state["mu"] = torch.zeros_like(p)
                mus.append(state["mu"])
            torch._foreach_mul_(mus, group["momentum"])
            torch._foreach_add_(mus, updates)
            torch._foreach_add_(params, mus, alpha=-group["lr"])


def barlow_learning_rate(step, epochs, steps_per_epoch, batch_size):
    """The schedule of adjust_learning_rate (EEG-BarlowNetworks/barlow_utils.py:8-21) before the per-group weights:
    linear warm-up over 10 epochs to batch_size / 256, then a cosine from there down to 0.1 % of it."""
    import math
    peak = batch_size / 256
    warm, total = 10 * steps_per_epoch, epochs * steps_per_epoch
    if step < warm:
        return peak * step / warm
    phase = 0.5 * (1 + math.cos(math.pi * (step - warm) / (total - warm)))
    return peak * phase + peak * 0.001 * (1 - phase)
