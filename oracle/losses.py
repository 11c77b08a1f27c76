"""Oracle: distillation losses of the hot path (test infrastructure only).

numpy float64 restatements of
  * CosineSimilarityLoss      /root/reference/LstmDistillFromDinoV2Train.py:36-43
  * FeatureDistributionLoss   /root/reference/LstmDistillFromDinoV2Train.py:107-140
  * loss_fn_kd                /root/reference/LstmDistillFromDinoV2TrainSpampinato.py:107-121
  * Barlow-Twins loss         /root/reference/EEG-BarlowNetworks/net.py:6-9,33-42
  * LARS step                 /root/reference/EEG-BarlowNetworks/optim.py:17-44
  * adjust_learning_rate      /root/reference/EEG-BarlowNetworks/barlow_utils.py:8-21
Quirks are reproduced on purpose: teacher probabilities are
used as *logits* and student probabilities as *targets* in FeatureDistributionLoss;
``nn.KLDivLoss()`` keeps its default ``reduction='mean'`` (mean over all elements).
Pinned against torch CPU outputs in tests/golden/losses.npz.
"""
import math
import numpy as np


def _softmax(x, axis=-1):
    x = x - x.max(axis=axis, keepdims=True)
    e = np.exp(x)
    return e / e.sum(axis=axis, keepdims=True)


def _log_softmax(x, axis=-1):
    x = x - x.max(axis=axis, keepdims=True)
    return x - np.log(np.exp(x).sum(axis=axis, keepdims=True))


def cosine_similarity_loss(student, teacher, eps=1e-8):
    """1 - mean_b cos(s_b, t_b); nn.CosineSimilarity(dim=1, eps=1e-8)."""
    s = np.asarray(student, np.float64)
    t = np.asarray(teacher, np.float64)
    sn = np.maximum(np.linalg.norm(s, axis=1, keepdims=True), eps)
    tn = np.maximum(np.linalg.norm(t, axis=1, keepdims=True), eps)
    cos = ((s / sn) * (t / tn)).sum(axis=1)
    return 1.0 - cos.mean()


def cosine_similarity_loss_grad(student, teacher, eps=1e-8):
    """d(loss)/d(student) for non-degenerate rows (norms > eps)."""
    s = np.asarray(student, np.float64)
    t = np.asarray(teacher, np.float64)
    B = s.shape[0]
    sn = np.linalg.norm(s, axis=1, keepdims=True)
    tn = np.linalg.norm(t, axis=1, keepdims=True)
    cos = (s * t).sum(axis=1, keepdims=True) / (sn * tn)
    dcos = t / (sn * tn) - cos * s / (sn * sn)
    return -dcos / B


def teacher_temp_schedule(nepochs, warmup_teacher_temp=1.5, teacher_temp=0.22, warmup_epochs=50):
    """LstmDistillFromDinoV2Train.py:112-116."""
    return np.concatenate((np.linspace(warmup_teacher_temp, teacher_temp, warmup_epochs),
                           np.ones(nepochs - warmup_epochs) * teacher_temp))


def cross_entropy_index(logits, labels):
    lp = _log_softmax(np.asarray(logits, np.float64), axis=1)
    return -lp[np.arange(lp.shape[0]), np.asarray(labels)].mean()


def cross_entropy_prob(logits, target_probs):
    lp = _log_softmax(np.asarray(logits, np.float64), axis=1)
    return -(np.asarray(target_probs, np.float64) * lp).sum(axis=1).mean()


def feature_distribution_loss(student, teacher, T, labels, pred_label, alpha=0.5, beta=0.5):
    """alpha*CE(pred_label,label) + beta*CE(softmax(teacher/T) as logits, softmax(student/T) as target)."""
    tp = _softmax(np.asarray(teacher, np.float64) / T)
    sp = _softmax(np.asarray(student, np.float64) / T)
    return alpha * cross_entropy_index(pred_label, labels) + beta * cross_entropy_prob(tp, sp)


def loss_fn_kd(outputs, labels, teacher_outputs, alpha, temperature):
    """KLDiv(log_softmax(s/T), softmax(t/T), reduction='mean')*(alpha*T*T) + CE(s,labels)*(1-alpha)."""
    T = temperature
    lsp = _log_softmax(np.asarray(outputs, np.float64) / T, axis=1)
    tp = _softmax(np.asarray(teacher_outputs, np.float64) / T, axis=1)
    kl = (tp * (np.log(tp) - lsp)).mean()
    return kl * (alpha * T * T) + cross_entropy_index(outputs, labels) * (1.0 - alpha)


def feature_distribution_loss_kd(student, teacher, T, labels, soft_w=0.25, ce_w=0.75):
    """FeatureDistributionLoss of LstmDistillFromDinoV2TrainSpampinato.py:125-184:
    soft_w * [sum p_t (log p_t - log_softmax(s/T)) / B * T^2] + ce_w * CE(s, label)."""
    s = np.asarray(student, np.float64)
    tp = _softmax(np.asarray(teacher, np.float64) / T)
    soft = (tp * (np.log(tp) - _log_softmax(s / T))).sum() / s.shape[0] * (T ** 2)
    return soft_w * soft + ce_w * cross_entropy_index(s, labels)


def feature_distribution_loss_soft(student, teacher, T):
    """FeatureDistributionLoss of LstmDistillFromDinoV2Eval.py:106-146: the soft-target term alone."""
    s = np.asarray(student, np.float64)
    tp = _softmax(np.asarray(teacher, np.float64) / T)
    return (tp * (np.log(tp) - _log_softmax(s / T))).sum() / s.shape[0] * (T ** 2)


def feature_distribution_loss_mse(student, teacher):
    """FeatureDistributionLoss of LstmDistillation.py:161-172: 0.4 (std_s - std_t)^2 + 0.4 (mean_s - mean_t)^2
    + 0.2 MSE, global statistics, torch ``std`` (ddof 1)."""
    s, t = np.asarray(student, np.float64), np.asarray(teacher, np.float64)
    return 0.4 * (s.std(ddof=1) - t.std(ddof=1)) ** 2 + 0.4 * (s.mean() - t.mean()) ** 2 + 0.2 * ((s - t) ** 2).mean()


def dino_loss(student_vbo, teacher_vbo, center, teacher_temp, student_temp=0.1, center_momentum=0.9, world=1):
    """DINOLoss.forward + update_center of LstmDistillation.py:118-159 on stacked views: student [V, B, out],
    teacher [2, B, out] (:583-586).  Quirks kept: ``teacher_out.chunk(1)`` leaves ONE chunk holding both teacher
    views, so the loop compares it (broadcast) with every student view but view 0; ``update_center`` sums over
    dim 0 -- the VIEW axis -- and divides by len(teacher_output) = 2, so the centre becomes [1, B, out].
    Returns (loss, new_center)."""
    so = np.asarray(student_vbo, np.float64) / student_temp
    to = np.asarray(teacher_vbo, np.float64)
    q = _softmax((to - center) / teacher_temp)
    terms = [(-(q * _log_softmax(so[v:v + 1])).sum(-1)).mean() for v in range(1, so.shape[0])]
    batch_center = to.sum(axis=0, keepdims=True) / (to.shape[0] * world)
    return float(np.mean(terms)), center * center_momentum + batch_center * (1 - center_momentum)


def dino_head(x, sd):
    """DINOHead.forward (LstmDistillation.py:66-99): Linear/GELU stack -> L2 normalise -> weight-normed Linear."""
    from math import sqrt
    from scipy.special import erf
    h = np.asarray(x, np.float64)
    keys = sorted({int(k.split(".")[1]) for k in sd if k.startswith("mlp.")})
    for n, i in enumerate(keys):
        h = h @ sd[f"mlp.{i}.weight"].T + sd[f"mlp.{i}.bias"]
        if n + 1 < len(keys):
            h = 0.5 * h * (1.0 + erf(h / sqrt(2.0)))
    h = h / np.maximum(np.linalg.norm(h, axis=-1, keepdims=True), 1e-12)
    v, g = sd["last_layer.weight_v"], sd["last_layer.weight_g"]
    w = v * (g / np.linalg.norm(v, axis=1, keepdims=True))
    return h @ w.T


def cosine_scheduler(base_value, final_value, epochs, niter_per_ep, warmup_epochs=0, start_warmup_value=0):
    """utils/utils.py:187-198."""
    warm = np.linspace(start_warmup_value, base_value, warmup_epochs * niter_per_ep) if warmup_epochs > 0 else np.array([])
    iters = np.arange(epochs * niter_per_ep - len(warm))
    sched = final_value + 0.5 * (base_value - final_value) * (1 + np.cos(np.pi * iters / len(iters)))
    return np.concatenate((warm, sched))


def batchnorm_noaffine(z, eps=1e-5):
    z = np.asarray(z, np.float64)
    mu = z.mean(axis=0, keepdims=True)
    var = z.var(axis=0, keepdims=True)
    return (z - mu) / np.sqrt(var + eps)


def off_diagonal_sqsum(c):
    """sum of squares of off-diagonal elements (net.py:6-9 + :40)."""
    c = np.asarray(c, np.float64)
    return (c ** 2).sum() - (np.diagonal(c) ** 2).sum()


def barlow_loss_sharded(z1, z2, world, lambd=0.0051):
    """net.py:33-42 on ``world`` ranks: BatchNorm statistics per rank (plain BatchNorm1d, no SyncBN in the loss),
    c = sum over ranks of bn(z1_r)^T bn(z2_r) / global batch (the in-place all_reduce, :38).  Also returns the
    gradient each rank's autograd produces for its own (z1_r, z2_r): the all-reduce is invisible to autograd, so it
    is d loss(c_global) / d z_r through the rank's own term only."""
    z1, z2 = np.asarray(z1, np.float64), np.asarray(z2, np.float64)
    n = z1.shape[0] // world
    parts = [(z1[r * n:(r + 1) * n], z2[r * n:(r + 1) * n]) for r in range(world)]
    c = sum(batchnorm_noaffine(a).T @ batchnorm_noaffine(b) for a, b in parts) / z1.shape[0]
    on = ((np.diagonal(c) - 1.0) ** 2).sum()
    loss = on + lambd * off_diagonal_sqsum(c)
    dc = 2 * lambd * c
    dc[np.diag_indices_from(dc)] = 2 * (np.diagonal(c) - 1.0)
    grads = []
    for a, b in parts:
        na, nb = batchnorm_noaffine(a), batchnorm_noaffine(b)
        grads.append((_bn_backward(a, nb @ dc.T / z1.shape[0]), _bn_backward(b, na @ dc / z1.shape[0])))
    return loss, c, grads


def _bn_backward(z, dy, eps=1e-5):
    z = np.asarray(z, np.float64)
    m = z.shape[0]
    mu, var = z.mean(0, keepdims=True), z.var(0, keepdims=True)
    xh = (z - mu) / np.sqrt(var + eps)
    return (dy - dy.mean(0, keepdims=True) - xh * (dy * xh).mean(0, keepdims=True)) / np.sqrt(var + eps) * (m / m)


def barlow_loss(z1, z2, batch_size, lambd=0.0051):
    c = batchnorm_noaffine(z1).T @ batchnorm_noaffine(z2) / batch_size
    on = ((np.diagonal(c) - 1.0) ** 2).sum()
    off = off_diagonal_sqsum(c)
    return on + lambd * off, c


def lars_step(p, grad, mu, lr, weight_decay=0.0, momentum=0.9, eta=0.001,
              weight_decay_filter=False, lars_adaptation_filter=False):
    """One LARS update of one parameter (optim.py:17-44). Returns (new_p, new_mu)."""
    p = np.asarray(p, np.float64)
    dp = np.asarray(grad, np.float64)
    is_1d = p.ndim == 1
    if not weight_decay_filter or not is_1d:
        dp = dp + weight_decay * p
    if not lars_adaptation_filter or not is_1d:
        pn = np.linalg.norm(p)
        un = np.linalg.norm(dp)
        q = (eta * pn / un) if (pn > 0 and un > 0) else 1.0
        dp = dp * q
    mu = momentum * np.asarray(mu, np.float64) + dp
    return p - lr * mu, mu


def barlow_lr(step, epochs, steps_per_epoch, batch_size):
    """adjust_learning_rate (barlow_utils.py:8-21) -- returns the un-weighted lr."""
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
