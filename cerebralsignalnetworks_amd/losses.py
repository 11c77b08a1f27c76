"""Losses of the distillation path, same names / argument meaning as the reference classes.

  CosineSimilarityLoss       LstmDistillFromDinoV2Train.py:36-43   (fused HIP forward+gradient)
  FeatureDistributionLoss    LstmDistillFromDinoV2Train.py:107-140 (torch ops; not hot)
  loss_fn_kd                 LstmDistillFromDinoV2TrainSpampinato.py:107-121
  BarlowTwinsLoss            EEG-BarlowNetworks/net.py:33-42 (HIP off-diagonal reduction)
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


class BarlowTwinsLoss(nn.Module):
    """net.py:33-42 on two embedding batches: bn(z1).T @ bn(z2) / batch_size, all_reduce, on + lambd*off."""

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
        on_diag, off_diag = _BarlowReduce.apply(c.float())
        return on_diag + self.lambd * off_diag


class _AllReduceSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.clone()
        torch.distributed.all_reduce(x)
        return x

    @staticmethod
    def backward(ctx, g):
        g = g.clone()
        torch.distributed.all_reduce(g)
        return g


class LARS(torch.optim.Optimizer):
    """LARS of /root/reference/EEG-BarlowNetworks/optim.py:5-44 (bias/norm exclusion = ``ndim == 1``)."""

    def __init__(self, params, lr, weight_decay=0, momentum=0.9, eta=0.001, weight_decay_filter=False,
                 lars_adaptation_filter=False):
        defaults = dict(lr=lr, weight_decay=weight_decay, momentum=momentum, eta=eta,
                        weight_decay_filter=weight_decay_filter, lars_adaptation_filter=lars_adaptation_filter)
        super().__init__(params, defaults)

    @staticmethod
    def exclude_bias_and_norm(p):
        return p.ndim == 1

    @torch.no_grad()
    def step(self):
        for g in self.param_groups:
            for p in g['params']:
                dp = p.grad
                if dp is None:
                    continue
                if not g['weight_decay_filter'] or not self.exclude_bias_and_norm(p):
                    dp = dp.add(p, alpha=g['weight_decay'])
                if not g['lars_adaptation_filter'] or not self.exclude_bias_and_norm(p):
                    param_norm, update_norm = torch.norm(p), torch.norm(dp)
                    one = torch.ones_like(param_norm)
                    q = torch.where(param_norm > 0., torch.where(update_norm > 0, (g['eta'] * param_norm / update_norm), one), one)
                    dp = dp.mul(q)
                state = self.state[p]
                if 'mu' not in state:
                    state['mu'] = torch.zeros_like(p)
                mu = state['mu']
                mu.mul_(g['momentum']).add_(dp)
                p.add_(mu, alpha=-g['lr'])


def barlow_learning_rate(step, epochs, steps_per_epoch, batch_size):
    """adjust_learning_rate of EEG-BarlowNetworks/barlow_utils.py:8-21 (un-weighted lr)."""
    import math
    max_steps = epochs * steps_per_epoch
    warmup_steps = 10 * steps_per_epoch
    base_lr = batch_size / 256
    if step < warmup_steps:
        return base_lr * step / warmup_steps
    step -= warmup_steps
    max_steps -= warmup_steps
    q = 0.5 * (1 + math.cos(math.pi * step / max_steps))
    end_lr = base_lr * 0.001
    return base_lr * q + end_lr * (1 - q)
