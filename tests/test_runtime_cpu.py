"""CPU: the host runtime helpers behind ``from utils import utils`` (cerebralsignalnetworks_amd/runtime.py) against a
fixture made by EXECUTING the reference's own definitions (tests/golden/make_ref_goldens.py:ref_runtime ->
ref_runtime.npz: utils/utils.py SmoothedValue, MetricLogger, clip_gradients, cancel_gradients_last_layer,
get_params_groups, bool_flag, accuracy), and ``init_distributed_mode`` over a 2-process gloo group."""
import argparse
import contextlib
import io
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reference_import_paths_resolve():
    """The imports the reference's hot-path scripts make (LstmDistillFromDinoV2Train.py:1-9, utils/PerilsEEGDataset.py:9,
    LstmDistillation.py:14-17) resolve with this repository first on the path."""
    from utils import utils
    from utils.PerilsEEGDataset import EEGDataset  # noqa: F401
    from utils.Utilities import evaluate  # noqa: F401
    from utils.EEGFilters import EEGFilters  # noqa: F401
    from utils.CustomModel import CustomModel  # noqa: F401
    from models.lstm import Model  # noqa: F401
    for name in ("init_distributed_mode", "MetricLogger", "SmoothedValue", "save_on_master", "is_main_process", "get_rank",
                 "get_world_size", "setup_for_distributed", "reduce_dict", "cosine_scheduler", "MultiCropWrapper", "LARS",
                 "clip_gradients", "cancel_gradients_last_layer", "get_params_groups", "has_batchnorms", "bool_flag",
                 "fix_random_seeds", "restart_from_checkpoint", "accuracy"):
        assert callable(getattr(utils, name)), name
    assert utils.get_rank() == 0 and utils.get_world_size() == 1 and utils.is_main_process()


def test_meters_match_the_executed_reference(golden):
    from utils import utils
    g = golden("ref_runtime.npz")
    series = g["series"]
    for win in (20, 4, 5):
        sv = utils.SmoothedValue(window_size=win)
        stats = []
        for i, v in enumerate(series):
            sv.update(float(v), n=1 + (i % 3))
            stats.append([sv.median, sv.avg, sv.global_avg, sv.max, sv.value])
        np.testing.assert_allclose(np.array(stats), g[f"smoothed_w{win}"], rtol=0, atol=1e-6)
        assert str(sv) == str(g[f"smoothed_w{win}_str"])
    ml = utils.MetricLogger(delimiter="  ")
    for i, v in enumerate(series[:11]):
        ml.update(loss=torch.tensor(float(v)), lr=0.001 * (i + 1), step=i)
    assert str(ml) == str(g["logger_str"])
    assert abs(ml.loss.global_avg - float(g["logger_loss_global_avg"])) < 1e-12
    with pytest.raises(AttributeError):
        ml.no_such_meter
    with pytest.raises(TypeError):
        ml.update(loss="high")


def test_log_every_yields_everything_and_prints_the_reference_fields():
    from utils import utils
    ml = utils.MetricLogger(delimiter="  ")
    buf, seen = io.StringIO(), []
    with contextlib.redirect_stdout(buf):
        for item in ml.log_every(list(range(23)), 10, "Epoch: [3]"):
            seen.append(item)
            ml.update(loss=0.5 * item)
    assert seen == list(range(23))
    lines = buf.getvalue().strip().split("\n")
    assert len(lines) == 4 + 1                                   # items 0, 10, 20, 22 + the total
    assert lines[0].startswith("Epoch: [3]  [ 0/23]  eta: ") and "time: " in lines[0] and "data: " in lines[0]
    assert "loss: " in lines[1] and lines[3].startswith("Epoch: [3]  [22/23]")
    assert lines[-1].startswith("Epoch: [3] Total time: ") and lines[-1].endswith("s / it)")


class _Head(nn.Module):
    def __init__(self):
        super().__init__()
        self.mlp = nn.Linear(6, 5)
        self.norm = nn.LayerNorm(5)
        self.last_layer = nn.Linear(5, 3, bias=False)

    def forward(self, x):
        return self.last_layer(self.norm(self.mlp(x)))


def test_step_loop_helpers_match_the_executed_reference(golden):
    from utils import utils
    g = golden("ref_runtime.npz")
    net = _Head()
    names = [n for n, _ in net.named_parameters()]
    assert names == [str(n) for n in g["clip_names"]]
    for clip in (0.3, 3.0):
        for n, p in net.named_parameters():
            p.data = torch.from_numpy(g[f"clip_param__{n}"].copy())
            p.grad = torch.from_numpy(g[f"clip_grad_before__{n}"].copy())
        norms = utils.clip_gradients(net, clip)
        np.testing.assert_allclose(norms, g[f"clip{clip}_norms"], rtol=1e-6)
        for n, p in net.named_parameters():
            np.testing.assert_allclose(p.grad.numpy(), g[f"clip{clip}_grad_after__{n}"], rtol=1e-6, atol=1e-9)
    groups = utils.get_params_groups(net)
    idx = {id(p): n for n, p in net.named_parameters()}
    assert [idx[id(p)] for p in groups[0]["params"]] == [str(n) for n in g["groups_regularized"]]
    assert [idx[id(p)] for p in groups[1]["params"]] == [str(n) for n in g["groups_not_regularized"]]
    assert groups[1]["weight_decay"] == float(g["groups_wd1"]) and "weight_decay" not in groups[0]
    utils.cancel_gradients_last_layer(0, net, 1)
    assert [n for n, p in net.named_parameters() if p.grad is None] == [str(n) for n in g["cancel_epoch0_none"]]
    for p in net.parameters():
        p.grad = torch.zeros_like(p)
    utils.cancel_gradients_last_layer(1, net, 1)                 # past the frozen epochs: nothing cancelled
    assert all(p.grad is not None for p in net.parameters())
    acc = utils.accuracy(torch.from_numpy(g["acc_logits"]), torch.from_numpy(g["acc_target"]), topk=(1, 5))
    np.testing.assert_allclose([float(a) for a in acc], g["acc_top1_5"], rtol=1e-6)
    assert [utils.bool_flag(s) for s in ("on", "True", "1")] == list(g["bool_flag_true"])
    assert [utils.bool_flag(s) for s in ("off", "FALSE", "0")] == list(g["bool_flag_false"])
    with pytest.raises(argparse.ArgumentTypeError):
        utils.bool_flag("maybe")
    assert not utils.has_batchnorms(net) and utils.has_batchnorms(nn.Sequential(nn.Linear(2, 2), nn.BatchNorm1d(2)))


def test_restart_from_checkpoint_and_save_on_master(tmp_path):
    from utils import utils
    a, b = nn.Linear(3, 2), nn.Linear(3, 2)
    opt = torch.optim.SGD(a.parameters(), lr=0.1, momentum=0.9)
    a(torch.ones(1, 3)).sum().backward()
    opt.step()
    path = str(tmp_path / "checkpoint.pth")
    utils.save_on_master({"student": a.state_dict(), "optimizer": opt.state_dict(), "epoch": 7}, path)
    run = {"epoch": 0}
    opt_b = torch.optim.SGD(b.parameters(), lr=0.1, momentum=0.9)
    with contextlib.redirect_stdout(io.StringIO()):
        utils.restart_from_checkpoint(path, run_variables=run, student=b, optimizer=opt_b, teacher=None)
        utils.restart_from_checkpoint(str(tmp_path / "absent.pth"), run_variables=run, student=b)      # no file: no-op
    assert run["epoch"] == 7
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert torch.equal(pa, pb)
    assert "momentum_buffer" in next(iter(opt_b.state_dict()["state"].values()))


def _init_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), CSN_DIST_BACKEND="gloo")
    sys.path.insert(0, ROOT)
    from utils import utils
    args = argparse.Namespace(dist_url="env://")
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        utils.init_distributed_mode(args)
        print("only the master prints this")
        print("every rank prints this", force=True)
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    ml = utils.MetricLogger()
    ml.update(loss=float(rank), acc=10.0 * rank)
    ml.update(loss=float(rank) + 2.0)
    ml.synchronize_between_processes()
    red = utils.reduce_dict({"b": torch.tensor(float(rank)), "a": torch.tensor(1.0 + rank)})
    out[rank] = dict(rank=args.rank, world=args.world_size, gpu=args.gpu, printed=buf.getvalue(), allreduce=float(t.item()),
                     get_rank=utils.get_rank(), get_world=utils.get_world_size(), main=utils.is_main_process(),
                     loss_avg=ml.loss.global_avg, loss_count=ml.loss.count, acc_avg=ml.acc.global_avg,
                     red_a=float(red["a"]), red_b=float(red["b"]))
    dist.barrier()
    dist.destroy_process_group()


def test_init_distributed_mode_two_process_gloo():
    """utils.init_distributed_mode(args) from the launcher's environment (utils/utils.py:467-475 branch): rank / world /
    device slot filled in, the group usable, print silenced off the master, meters and reduce_dict reduced over both ranks."""
    world, port = 2, 29641
    out = mp.Manager().dict()
    mp.spawn(_init_worker, args=(world, port, out), nprocs=world, join=True)
    for r in range(world):
        o = out[r]
        assert (o["rank"], o["world"], o["gpu"]) == (r, world, r)
        assert (o["get_rank"], o["get_world"], o["main"]) == (r, world, r == 0)
        assert o["allreduce"] == 3.0
        assert "| distributed init (rank %d): env://" % r in o["printed"]
        assert ("only the master prints this" in o["printed"]) == (r == 0)
        assert "every rank prints this" in o["printed"]
        # losses 0, 2 (rank 0) and 1, 3 (rank 1): global average 1.5 over 4 updates on both ranks
        assert o["loss_count"] == 4 and abs(o["loss_avg"] - 1.5) < 1e-12 and abs(o["acc_avg"] - 5.0) < 1e-12
        assert abs(o["red_a"] - 1.5) < 1e-6 and abs(o["red_b"] - 0.5) < 1e-6


def test_init_distributed_mode_exits_without_a_gpu_like_the_reference(monkeypatch, capsys):
    """utils/utils.py:487-489: 'Does not support training without GPU.' + exit code 1 (the RCCL backend needs a device)."""
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    from utils import utils
    monkeypatch.delenv("CSN_DIST_BACKEND", raising=False)
    for k in ("RANK", "WORLD_SIZE", "SLURM_PROCID"):
        monkeypatch.delenv(k, raising=False)
    with pytest.raises(SystemExit) as e:
        utils.init_distributed_mode(argparse.Namespace(dist_url="env://"))
    assert e.value.code == 1
    assert "Does not support training without GPU." in capsys.readouterr().out
