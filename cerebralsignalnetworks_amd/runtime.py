"""Host-side runtime helpers with the call surface of the reference's ``utils/utils.py`` -- the part of that module the
hot-path scripts reach (``from utils import utils`` at /root/reference/LstmDistillFromDinoV2Train.py:9,
utils/PerilsEEGDataset.py:9, LstmDistillation.py:14):

    init_distributed_mode(args)            utils/utils.py:467-503   (:234 of the trainer)
    MetricLogger / SmoothedValue           :224-284, :313-400       (PerilsEEGDataset.py:170,325 ``log_every``)
    is_main_process / save_on_master / get_rank / get_world_size / setup_for_distributed / reduce_dict   :286-464
    clip_gradients, cancel_gradients_last_layer, get_params_groups, has_batchnorms, bool_flag, fix_random_seeds,
    restart_from_checkpoint, accuracy      :132-222, :506-514, :636-655   (LstmDistillation.py step loop)

Written for this framework, not transcribed: one process per GPU over RCCL (``backend="nccl"`` with the rank's device
bound at init -- the reference left ``gloo`` switched on at :491-492), meters that reduce their totals with one
collective for the whole logger, gradient clipping with one norm kernel for all parameters.  ``cosine_scheduler``,
``MultiCropWrapper`` (dino.py) and ``LARS`` (losses.py) are re-exported by the ``utils/utils.py`` shim next to these.
"""
import argparse
import builtins
import datetime
import os
import sys
import time
from collections import defaultdict, deque

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn


# ---- process group ------------------------------------------------------------------------------------------------
def is_dist_avail_and_initialized():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank():
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def is_main_process():
    return get_rank() == 0


def save_on_master(*args, **kwargs):
    """``torch.save`` on rank 0 only (utils/utils.py:447-449)."""
    if is_main_process():
        torch.save(*args, **kwargs)


_builtin_print = builtins.print


def setup_for_distributed(is_master):
    """After this call ``print`` is silent on every rank but the master; ``print(..., force=True)`` always prints
    (utils/utils.py:452-464).  Idempotent: the wrapper always forwards to the interpreter's original ``print``."""
    def rank_print(*args, force=False, **kwargs):
        if is_master or force:
            _builtin_print(*args, **kwargs)
    builtins.print = rank_print


def init_distributed_mode(args):
    """Fills ``args.rank / .world_size / .gpu`` and initialises ``torch.distributed`` from, in this order
    (utils/utils.py:467-503): the launcher's environment (``RANK``, ``WORLD_SIZE``, ``LOCAL_RANK``: torchrun), a SLURM
    task (``SLURM_PROCID``), or a single process on one GPU; without a GPU it prints the reference's message and exits 1.
    Differences, on purpose: the backend is RCCL (``nccl``) with the rank's device bound at init, so the first
    collective does not have to guess a device; ``CSN_DIST_BACKEND=gloo`` is the rehearsal switch used by the CPU tests
    and the one-GPU rehearsals (``CSN_SINGLE_DEVICE``: every rank on device 0); the rendezvous address defaults to the
    loopback interface."""
    backend = os.environ.get("CSN_DIST_BACKEND", "nccl")
    have_gpu = torch.cuda.is_available()
    if not have_gpu and backend == "nccl":
        print('Does not support training without GPU.')
        sys.exit(1)
    ndev = max(1, torch.cuda.device_count())
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ:
        args.rank, args.world_size = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        args.gpu = int(os.environ.get("LOCAL_RANK", args.rank % ndev))
    elif "SLURM_PROCID" in os.environ:
        args.rank = int(os.environ["SLURM_PROCID"])
        args.world_size = int(os.environ.get("SLURM_NTASKS", "1"))
        args.gpu = args.rank % ndev
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29501")
    else:
        print('Will run the code on one GPU.')
        args.rank, args.gpu, args.world_size = 0, 0, 1
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29501")
    if os.environ.get("CSN_SINGLE_DEVICE"):
        args.gpu = 0
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    init_method = getattr(args, "dist_url", None) or "env://"
    kw = dict(backend=backend, init_method=init_method, world_size=args.world_size, rank=args.rank)
    if have_gpu:
        torch.cuda.set_device(args.gpu)
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", args.gpu)
    dist.init_process_group(**kw)
    print('| distributed init (rank {}): {}'.format(args.rank, init_method), flush=True)
    dist.barrier()
    setup_for_distributed(args.rank == 0)


def _reduce_device():
    """Where a small tensor must live to be all-reduced by the current backend."""
    if is_dist_avail_and_initialized() and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def reduce_dict(input_dict, average=True):
    """{name: scalar tensor} summed (or averaged) over the ranks with ONE collective; keys are walked in sorted order so
    every rank stacks the same layout (utils/utils.py:286-310)."""
    world = get_world_size()
    if world < 2:
        return input_dict
    with torch.no_grad():
        names = sorted(input_dict.keys())
        stacked = torch.stack([input_dict[k] for k in names], dim=0)
        dist.all_reduce(stacked)
        if average:
            stacked /= world
        return dict(zip(names, stacked))


# ---- meters -------------------------------------------------------------------------------------------------------
class SmoothedValue:
    """A running series: window statistics (median / avg / max / value over the last ``window_size`` updates) and the
    global average (utils/utils.py:224-284)."""

    def __init__(self, window_size=20, fmt=None):
        self.deque = deque(maxlen=window_size)
        self.total, self.count = 0.0, 0
        self.fmt = "{median:.6f} ({global_avg:.6f})" if fmt is None else fmt

    def update(self, value, n=1):
        self.deque.append(value)
        self.count += n
        self.total += value * n

    def synchronize_between_processes(self):
        """count / total summed over the ranks (the window stays local, as in the reference)."""
        if not is_dist_avail_and_initialized():
            return
        t = torch.tensor([self.count, self.total], dtype=torch.float64, device=_reduce_device())
        dist.all_reduce(t)
        self.count, self.total = int(t[0].item()), float(t[1].item())

    @property
    def median(self):
        # (the LOWER middle value of an even window, as torch.median in the reference, utils/utils.py:258-261)
        return float(np.float32(sorted(self.deque)[(len(self.deque) - 1) // 2])) if self.deque else float("nan")

    @property
    def avg(self):
        return float(np.mean(np.asarray(self.deque, dtype=np.float32))) if self.deque else float("nan")

    @property
    def global_avg(self):
        return self.total / self.count if self.count else float("nan")

    @property
    def max(self):
        return max(self.deque)

    @property
    def value(self):
        return self.deque[-1]

    def __str__(self):
        return self.fmt.format(median=self.median, avg=self.avg, global_avg=self.global_avg, max=self.max, value=self.value)


class MetricLogger:
    """Named ``SmoothedValue`` meters + the ``log_every`` progress generator (utils/utils.py:313-400)."""

    def __init__(self, delimiter="\t"):
        self.meters = defaultdict(SmoothedValue)
        self.delimiter = delimiter

    def update(self, **kwargs):
        for name, v in kwargs.items():
            if isinstance(v, torch.Tensor):
                v = v.item()
            if not isinstance(v, (float, int)):
                raise TypeError(f"MetricLogger.update({name}=...): a number or a 0-d tensor, not {type(v).__name__}")
            self.meters[name].update(v)

    def __getattr__(self, attr):
        meters = self.__dict__.get("meters", {})
        if attr in meters:
            return meters[attr]
        raise AttributeError("'{}' object has no attribute '{}'".format(type(self).__name__, attr))

    def __str__(self):
        return self.delimiter.join("{}: {}".format(name, meter) for name, meter in self.meters.items())

    def add_meter(self, name, meter):
        self.meters[name] = meter

    def synchronize_between_processes(self):
        """All meters' (count, total) in ONE all-reduce instead of a barrier + collective per meter."""
        if not is_dist_avail_and_initialized() or not self.meters:
            return
        names = sorted(self.meters.keys())
        t = torch.tensor([[self.meters[n].count, self.meters[n].total] for n in names], dtype=torch.float64,
                         device=_reduce_device())
        dist.all_reduce(t)
        for row, n in zip(t.tolist(), names):
            self.meters[n].count, self.meters[n].total = int(row[0]), float(row[1])

    def log_every(self, iterable, print_freq, header=None):
        """Yields the items of ``iterable`` and prints a progress line (position, eta, the meters, iteration / data time,
        peak device memory) every ``print_freq`` items and at the last one, then the total."""
        header = header or ''
        n = len(iterable)
        iter_time, data_time = SmoothedValue(fmt='{avg:.6f}'), SmoothedValue(fmt='{avg:.6f}')
        width = len(str(n))
        on_gpu = torch.cuda.is_available()
        t_start = t_prev = time.time()
        for i, obj in enumerate(iterable):
            data_time.update(time.time() - t_prev)
            yield obj
            iter_time.update(time.time() - t_prev)
            if i % print_freq == 0 or i == n - 1:
                eta = datetime.timedelta(seconds=int(iter_time.global_avg * (n - i)))
                fields = [header, f"[{i:{width}d}/{n}]", f"eta: {eta}", str(self), f"time: {iter_time}", f"data: {data_time}"]
                if on_gpu:
                    fields.append("max mem: {:.0f}".format(torch.cuda.max_memory_allocated() / (1024.0 * 1024.0)))
                print(self.delimiter.join(fields))
            t_prev = time.time()
        total = time.time() - t_start
        print('{} Total time: {} ({:.6f} s / it)'.format(header, datetime.timedelta(seconds=int(total)), total / max(1, n)))


# ---- step-loop helpers of the DINO trainer (LstmDistillation.py:567-615) -------------------------------------------
def clip_gradients(model, clip):
    """Per-parameter L2 clipping: every gradient whose norm exceeds ``clip`` is scaled to it; returns the norms
    (utils/utils.py:132-141).  One multi-tensor norm + one multi-tensor scale instead of a kernel and a host
    round trip per parameter."""
    grads = [p.grad for p in model.parameters() if p.grad is not None]
    if not grads:
        return []
    norms = torch.stack(torch._foreach_norm(grads, 2))
    coef = (clip / (norms + 1e-6)).clamp(max=1.0)
    torch._foreach_mul_(grads, list(coef.unbind()))
    return norms.tolist()


def cancel_gradients_last_layer(epoch, model, freeze_last_layer):
    """During the first ``freeze_last_layer`` epochs the DINO head's ``last_layer`` gets no update (utils/utils.py:144-149)."""
    if epoch >= freeze_last_layer:
        return
    for name, p in model.named_parameters():
        if "last_layer" in name:
            p.grad = None


def get_params_groups(model):
    """Two optimiser groups: weights (regularised) and biases / 1-d parameters with ``weight_decay`` 0 (:636-647)."""
    decay, no_decay = [], []
    for name, p in model.named_parameters():
        if p.requires_grad:
            (no_decay if name.endswith(".bias") or p.ndim == 1 else decay).append(p)
    return [{'params': decay}, {'params': no_decay, 'weight_decay': 0.}]


def has_batchnorms(model):
    return any(isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d, nn.SyncBatchNorm)) for m in model.modules())


def bool_flag(s):
    """argparse type for on/off flags (utils/utils.py:201-212)."""
    v = s.lower()
    if v in ("on", "true", "1"):
        return True
    if v in ("off", "false", "0"):
        return False
    raise argparse.ArgumentTypeError("invalid value for a boolean flag")


def fix_random_seeds(seed=31):
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)


def restart_from_checkpoint(ckp_path, run_variables=None, **kwargs):
    """Loads ``checkpoint[key]`` into every ``key=object`` given (``load_state_dict``, non-strict where the object
    takes the argument) and copies the entries named in ``run_variables`` (utils/utils.py:152-184).  The file is read
    with ``weights_only=True``: the reference's plain ``torch.load`` would unpickle arbitrary objects."""
    if not os.path.isfile(ckp_path):
        return
    print("Found checkpoint at {}".format(ckp_path))
    checkpoint = torch.load(ckp_path, map_location="cpu", weights_only=True)
    for key, obj in kwargs.items():
        if key not in checkpoint or obj is None:
            print("=> key '{}' not found in checkpoint: '{}'".format(key, ckp_path))
            continue
        try:
            msg = obj.load_state_dict(checkpoint[key], strict=False)
            print("=> loaded '{}' from checkpoint '{}' with msg {}".format(key, ckp_path, msg))
        except TypeError:          # optimisers / loss modules without a ``strict`` argument
            try:
                obj.load_state_dict(checkpoint[key])
                print("=> loaded '{}' from checkpoint: '{}'".format(key, ckp_path))
            except ValueError:
                print("=> failed to load '{}' from checkpoint: '{}'".format(key, ckp_path))
    if run_variables is not None:
        for name in run_variables:
            if name in checkpoint:
                run_variables[name] = checkpoint[name]


def accuracy(output, target, topk=(1,)):
    """Top-k accuracies in percent (utils/utils.py:506-513)."""
    k_max = max(topk)
    pred = output.topk(k_max, dim=1, largest=True, sorted=True).indices          # [B, k_max]
    hit = pred.eq(target.reshape(-1, 1))
    return [hit[:, :k].any(dim=1).float().sum() * 100.0 / target.size(0) for k in topk]
