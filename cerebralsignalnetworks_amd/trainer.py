"""Data-parallel distillation trainer for the hot loop of
/root/reference/LstmDistillFromDinoV2Train.py:351-375 (zero_grad -> forward -> loss ->
backward -> optimiser step), restructured for MI355X:

  * raw EEG segments [N,C,T], teacher embeddings [N,D] and labels stay resident in HBM;
    a step gathers its batch by index on the device (no per-item Python, no JPEG decode,
    no host->device copy, no ``.item()`` sync per step -- SURVEY.md section 7 H6);
  * preprocessing is the fused HIP band-pass + z-score, the encoder the HIP LSTM;
  * one process per GPU; gradients live in ONE flat float32 buffer that is all-reduced
    (SUM, then / world) with a single RCCL collective per step over xGMI -- 31 MB for the
    cfg2 model, so one large message instead of per-parameter buckets;
  * sharding follows DistributedSampler semantics (rank r takes indices r::world of a
    per-epoch permutation seeded by ``seed + epoch``).
"""
import math

import torch
import torch.distributed as dist
import torch.nn as nn

from . import filters
from .losses import (CosineSimilarityLoss, FeatureDistributionLoss, HyperParams, loss_fn_kd, BarlowTwinsLoss,
                     LARS)


def dist_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


class FlatGrads:
    """All parameter gradients as views into one contiguous float32 buffer; with ``flatten_params`` the parameters
    themselves too (same offsets), so that an optimiser can step the whole model in one pass.

    Data-parallel reduction: ``all_reduce_mean()`` after the backward = one blocking all-reduce of the whole buffer
    (31 MB at cfg2).  With ``segments`` (contiguous (start, end) element ranges in readiness order, set by
    ``DistillTrainer``) the reduction is BUCKETED and OVERLAPPED the way DistributedDataParallel does it for the
    reference (LstmDistillation.py:445): ``segment_ready(i)`` -- called from the LSTM backward's gradient-ready hook --
    starts an asynchronous all-reduce of that range on the communication stream while the remaining layers'
    weight-gradient GEMMs are still running; ``all_reduce_mean()`` then reduces what is left, waits for everything and
    scales by 1 / world.

    Which form is the DEFAULT is a measurement this repository could not make (no multi-GPU node was available to a
    builder round): ``CSN_AR_OVERLAP=1`` selects the overlapped form, the default is the single blocking collective.
    Reason for the caution: the kernels the collective would run beside are the weight-gradient GEMMs, grids of exactly
    one 512-thread / 160 KB-LDS workgroup per CU -- an RCCL kernel that holds k CUs when such a grid is dispatched pushes
    k of its tiles into a second wave (a 0.5 ms kernel becomes a 1 ms kernel), which would cost more than the 0.2 ms of
    exposed all-reduce the overlap can hide.  Both forms give the same bits (2-rank test) and the bench line says which
    one ran (``data_parallel.allreduce_overlapped``, ``allreduce_exposed_ms_per_step``)."""

    def __init__(self, params, flatten_params=False):
        self.params = [p for p in params if p.requires_grad]
        total = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(total, dtype=torch.float32, device=self.params[0].device)
        self.flat_params = torch.empty_like(self.flat) if flatten_params else None
        self.timing = None      # a list -> all_reduce_mean() records an event pair per call
        self.offsets = {}       # id(param) -> (first element, number of elements)
        self.segments = None    # [(start, end)] in readiness order, or None = one blocking collective
        self._pending, self._works = None, []
        off = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            if flatten_params:
                self.flat_params[off:off + n].copy_(p.data.reshape(-1))
                p.data = self.flat_params[off:off + n].view_as(p)
            self.offsets[id(p)] = (off, n)
            off += n

    def zero(self):
        self.flat.zero_()
        self._pending = None if self.segments is None else [True] * len(self.segments)
        self._works = []

    def _reduce_runs(self, idx):
        """async all-reduce of the segments ``idx`` (merged into contiguous runs: fewer, larger messages)"""
        runs = []
        for s, e in sorted(self.segments[i] for i in idx):
            if runs and runs[-1][1] == s:
                runs[-1][1] = e
            else:
                runs.append([s, e])
        for s, e in runs:
            if e > s:
                self._works.append(dist.all_reduce(self.flat[s:e], op=dist.ReduceOp.SUM, async_op=True))

    def segment_ready(self, i, also=()):
        """Gradients of segment ``i`` (and of the segments ``also``, if still pending) are final on the current stream."""
        _, world = dist_info()
        if world == 1 or self._pending is None:
            return
        idx = [j for j in (i, *also) if self._pending[j]]
        for j in idx:
            self._pending[j] = False
        self._reduce_runs(idx)

    def all_reduce_mean(self):
        rank, world = dist_info()
        if world > 1:
            ev = None
            if self.timing is not None and self.flat.is_cuda:     # bench.py: event pair on the launching stream
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            if self._pending is None:
                dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            else:
                rest = [j for j, p in enumerate(self._pending) if p]
                self._pending = [False] * len(self._pending)
                self._reduce_runs(rest)
                for w in self._works:
                    w.wait()
                self._works = []
            self.flat.div_(world)
            if ev is not None:
                ev[1].record()
                self.timing.append(ev)

    def all_reduce_ms(self):
        """Mean milliseconds between the end of the backward's launches and the reduced, scaled gradients on the launch
        stream: the whole all-reduce in the blocking form, its EXPOSED part in the overlapped form; blocking."""
        if not self.timing:
            return None
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in self.timing) / len(self.timing)


class FlatRMSprop:
    """``torch.optim.RMSprop(params, lr)`` (LstmDistillFromDinoV2Train.py:329: alpha 0.99, eps 1e-8, no momentum, not
    centred) as ONE fused HIP pass over the flat parameter / gradient / state buffers of a ``FlatGrads`` (csn_rmsprop_step)
    instead of torch's five multi-tensor kernels per step.  ``state_dict`` keeps the optimiser resumable."""

    def __init__(self, flat, lr=1e-3, alpha=0.99, eps=1e-8):
        assert flat.flat_params is not None, "FlatRMSprop needs FlatGrads(..., flatten_params=True)"
        self.flat, self.lr, self.alpha, self.eps = flat, lr, alpha, eps
        self.square_avg = torch.zeros_like(flat.flat)
        self.param_groups = [{"lr": lr, "params": flat.params}]

    def check_views(self):
        """The parameters must still BE the views into the flat buffer this optimiser steps (model.float() / .to(dtype)
        / load_state_dict(assign=True) after construction re-home them and the model would silently stop training)."""
        base, off = self.flat.flat_params.data_ptr(), 0
        for p in self.flat.params:
            if p.data_ptr() != base + 4 * off:
                raise RuntimeError("FlatRMSprop: a parameter no longer lives in the flat buffer (re-homed after the "
                                   "trainer was built); rebuild the trainer")
            off += p.numel()

    @torch.no_grad()
    def step(self):
        from . import cabi
        self.check_views()      # (a dozen data_ptr() compares on the host: every step, so that a re-homed parameter is caught at once)
        cabi.rmsprop_step(self.flat.flat_params, self.flat.flat, self.square_avg, self.param_groups[0]["lr"], self.alpha, self.eps)

    def zero_grad(self, set_to_none=False):
        self.flat.zero()

    def state_dict(self):
        return {"square_avg": self.square_avg, "lr": self.param_groups[0]["lr"], "alpha": self.alpha, "eps": self.eps}

    def load_state_dict(self, sd):
        self.square_avg.copy_(sd["square_avg"])
        self.param_groups[0]["lr"], self.alpha, self.eps = sd["lr"], sd["alpha"], sd["eps"]



def split_indices(n, fractions=(0.8, 0.2), seed=43):
    """The reference's train / validation split: ``torch.utils.data.random_split(dataset, [0.8, 0.2],
    generator=torch.Generator().manual_seed(43))`` (LstmDistillFromDinoV2Train.py:289-290, ...Eval.py:324-325).
    Its semantics (floor of each fraction, the remainder dealt round-robin from the first split, consecutive
    slices of one ``randperm``) are taken from ``random_split`` itself, applied to ``range(n)``.
    Returns one int64 index tensor per fraction."""
    from torch.utils.data import random_split
    parts = random_split(range(n), list(fractions), generator=torch.Generator().manual_seed(seed))
    return [torch.as_tensor(list(part.indices), dtype=torch.long) for part in parts]


def shard_indices(n, epoch, seed, rank, world, shuffle=True, device="cpu"):
    """DistributedSampler semantics: pad to a multiple of world, rank takes r::world."""
    if shuffle:
        g = torch.Generator()
        g.manual_seed(seed + epoch)
        idx = torch.randperm(n, generator=g)
    else:
        idx = torch.arange(n)
    total = int(math.ceil(n / world)) * world
    if total > n:
        idx = torch.cat([idx, idx[: total - n]])
    return idx[rank:total:world].to(device)


def check_device_status(model, collective=True):
    """Blocking: raises if an in-kernel hand-off of a weight-stationary LSTM kernel timed out, or a non-finite gradient
    reached the backward recurrence, in ANY forward / backward of ``model`` since the last check (the status word is
    sticky).  Call it at epoch ends / after a timed region / after an evaluation pass, not per step.  Cleared once
    reported.  With torch.distributed initialised the verdict is all-reduced (MAX) first, so every rank raises together
    instead of one rank leaving the others in the next collective -- which makes the call itself a COLLECTIVE: every
    rank must reach it.  ``collective=False``: this rank's verdict only (work that one rank does alone, e.g. rank 0's
    fixture check before a benchmark's timed region)."""
    from . import cabi
    from .lstm_model import HipLSTM
    torch.cuda.synchronize()
    bad = 0
    dev = None
    for mod in model.modules():
        if isinstance(mod, HipLSTM):
            for plan in mod.all_plans():
                bad |= plan.status(clear=True)
                dev = plan.device
    _, world = dist_info()
    if world > 1 and collective:
        on_gpu = dist.get_backend() == "nccl"
        t = torch.tensor([bad & 1, (bad >> 1) & 1, (bad >> 2) & 1], dtype=torch.int32,
                         device=(dev if dev is not None else torch.device("cuda", torch.cuda.current_device())) if on_gpu else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        bad = int(t[0].item()) | (int(t[1].item()) << 1) | (int(t[2].item()) << 2)
    if bad & cabi.STATUS_TIMEOUT:
        raise RuntimeError("libcsn_hip: a bounded in-kernel wait of the LSTM recurrence timed out "
                           "(is another process using this GPU's CUs?); results are invalid")
    if bad & cabi.STATUS_STALE_SLOT:      # only the debug library (make tags) can raise this
        raise RuntimeError("libcsn_hip (tags build): a hand-off ring slot served its previous occupant; results are invalid")
    if bad & cabi.STATUS_NONFINITE:
        raise FloatingPointError("libcsn_hip: a non-finite (NaN / Inf) gradient reached the LSTM backward: the run has "
                                 "diverged (the reference would carry the NaN through its loss and weights)")


class DistillTrainer:
    def __init__(self, model, sos, ddof=0, loss="cosine", lr=1e-3, optimizer="rmsprop", nepochs=100,
                 kd_params=None, preprocess=True):
        self.model = model
        self.sos, self.ddof, self.preprocess = sos, ddof, preprocess
        self.loss_name = loss
        on_gpu = next(model.parameters()).is_cuda
        self.grads = FlatGrads(model.parameters(), flatten_params=(optimizer == "rmsprop" and on_gpu))
        params = self.grads.params
        if optimizer == "rmsprop" and on_gpu:   # LstmDistillFromDinoV2Train.py:329 -- one fused pass over the flat buffers
            self.opt = FlatRMSprop(self.grads, lr=lr)
        elif optimizer == "rmsprop":
            self.opt = torch.optim.RMSprop(params, lr=lr)
        elif optimizer == "adamw":      # LstmDistillFromDinoV2TrainSpampinato.py:378
            self.opt = torch.optim.AdamW(params, lr=lr)
        elif optimizer == "adam":       # LSTMDistill.py:322
            self.opt = torch.optim.Adam(params, lr=lr)
        elif optimizer == "lars":       # EEG-BarlowNetworks/train.py (weights + biases groups collapsed)
            self.opt = LARS(params, lr=lr, weight_decay=1e-6, weight_decay_filter=True, lars_adaptation_filter=True)
        else:
            raise ValueError(optimizer)
        self.cosine = CosineSimilarityLoss()
        self.featdist = FeatureDistributionLoss(nepochs, HyperParams.warmup_teacher_temp, HyperParams.teacher_temp,
                                                HyperParams.warmup_teacher_temp_epochs)
        self.kd_params = kd_params
        self.barlow = None
        rank, world = dist_info()
        if world > 1:   # identical initial weights on every rank
            for p in model.parameters():
                dist.broadcast(p.data, src=0)
        self._wire_lstm_gradients(world)

    def _wire_lstm_gradients(self, world):
        """The model's HIP LSTM writes its gradients straight into the flat buffer (one forward per step uses it: no
        temporaries, no accumulation pass), and -- data-parallel -- tells the buffer which layer's gradients are
        enqueued, so that their all-reduce overlaps the weight-gradient GEMMs of the layers below (FlatGrads)."""
        import os
        from .lstm_model import HipLSTM
        lstms = [m for m in self.model.modules() if isinstance(m, HipLSTM)]
        self._hook_error = None
        if len(lstms) != 1 or not self.grads.flat.is_cuda:
            return
        lstm = lstms[0]
        lstm.direct_grads = True
        L = lstm.num_layers
        first = [self.grads.offsets[id(getattr(lstm, f"weight_ih_l{k}"))][0] for k in range(L)]
        last = self.grads.offsets[id(getattr(lstm, f"bias_hh_l{L - 1}"))]
        lstm_end = last[0] + last[1]
        contiguous = all(first[k] < first[k + 1] for k in range(L - 1)) and first[0] == min(o for o, _ in self.grads.offsets.values())
        if world == 1 or not os.environ.get("CSN_AR_OVERLAP") or not contiguous:
            return
        total = self.grads.flat.numel()
        # segments: LSTM layer k = k, everything behind the LSTM's parameters (fc, class_pred) = L
        self.grads.segments = [(first[k], first[k + 1] if k + 1 < L else lstm_end) for k in range(L)] + [(lstm_end, total)]
        others = [p for p in self.grads.params if self.grads.offsets[id(p)][0] >= lstm_end]
        self._others_left = len(others)
        self._n_others = len(others)

        def other_done(_p):
            self._others_left -= 1
        for p in others:
            p.register_post_accumulate_grad_hook(other_done)

        def layer_ready(layer):
            try:
                if layer == 0:
                    return                      # the bottom layer's gradients come last: all_reduce_mean() takes them
                # the head's gradients were accumulated before the LSTM's backward ran (autograd order); if not, they wait
                self.grads.segment_ready(layer, also=(L,) if self._others_left == 0 else ())
            except Exception as e:              # (a ctypes callback cannot raise: kept for train_step)
                self._hook_error = e
        lstm.grad_ready_hook = layer_ready

    def embed(self, eeg_bct):
        """raw EEG [B,C,T] (device, float32) -> model input [B,T,C]."""
        if self.preprocess:
            # written time-major [T,B,C] and handed on as a [B,T,C] view: the LSTM's layout pass (x -> fragment-major
            # slabs per timestep) then reads whole cache lines instead of 32-byte pieces 256 KB apart
            return filters.eeg_bandpass_znorm(eeg_bct, self.sos, ddof=self.ddof, time_major=True).transpose(0, 1)
        return eeg_bct.transpose(1, 2).contiguous()

    def compute_loss(self, out, targets, labels, epoch):
        if self.loss_name == "cosine":
            feat = out[0] if isinstance(out, tuple) else out
            return self.cosine(feat, targets)
        if self.loss_name == "featdist":
            feat, cls = out
            return self.featdist(feat, targets, epoch, labels, pred_label=cls)
        if self.loss_name == "kd":
            feat = out[0] if isinstance(out, tuple) else out
            return loss_fn_kd(feat, labels, targets, self.kd_params)
        if self.loss_name == "barlow":
            # BASELINE.json config 5: Barlow-Twins cross-correlation between the LSTM embedding of the EEG view
            # and the (frozen) image embedding; net.py:33-42 with the global batch size and an all-reduced c
            feat = out[0] if isinstance(out, tuple) else out
            if self.barlow is None:
                _, world = dist_info()
                self.barlow = BarlowTwinsLoss(feat.shape[1], feat.shape[0] * world).to(feat.device)
            return self.barlow(feat, targets)
        raise ValueError(self.loss_name)

    def train_step(self, eeg_bct, targets, labels=None, epoch=0):
        """One optimisation step on this rank's shard of the global batch; returns the (device) loss."""
        self.model.train()
        self.grads.zero()
        if self.grads.segments is not None:
            self._others_left = self._n_others
        x = self.embed(eeg_bct)
        out = self.model(x)
        loss = self.compute_loss(out, targets, labels, epoch)
        loss.backward()
        if self._hook_error is not None:
            err, self._hook_error = self._hook_error, None
            raise err
        self.grads.all_reduce_mean()
        self.opt.step()
        return loss.detach()

    def check_device_status(self):
        """The model's sticky device status (time-outs; non-finite gradients where the weight-stationary backward saw
        them) AND, for every path -- the exact-f32 / per-diagonal kernels raise no status bit -- a finiteness test of the
        parameters and of the last step's gradients: a NaN that any earlier step produced is still in the parameters.
        Like the status word it reports ONCE: a state already reported (by this test or by the status word: a timed-out
        run leaves garbage) is not raised again at the next check."""
        bufs = [self.grads.flat] + ([self.grads.flat_params] if self.grads.flat_params is not None else
                                    [p.data for p in self.grads.params])
        finite = torch.stack([torch.isfinite(b).all() for b in bufs]).all()
        _, world = dist_info()
        if world > 1:
            t = (~finite).to(torch.int32).reshape(1)
            t = t if dist.get_backend() == "nccl" else t.cpu()
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            finite = t.item() == 0
        finite = bool(finite)
        reported_before = getattr(self, "_nonfinite_reported", False)
        self._nonfinite_reported = not finite
        check_device_status(self.model)          # (raises for a time-out / a non-finite gradient seen on the device)
        if not finite and not reported_before:
            raise FloatingPointError("non-finite (NaN / Inf) parameters or gradients: the run has diverged")

    @torch.no_grad()
    def embed_all(self, eeg_all, batch):
        self.model.eval()
        outs = []
        for i in range(0, eeg_all.shape[0], batch):
            o = self.model(self.embed(eeg_all[i:i + batch]))
            outs.append((o[0] if isinstance(o, tuple) else o).float())
        return torch.cat(outs)
