#!/usr/bin/env python3
"""Benchmark of the EEG -> LSTM distillation training step (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic input already resident in
HBM: fused band-pass + z-score -> 2-layer LSTM (hidden 768) forward -> fc -> cosine
distillation loss -> backward -> gradient all-reduce (RCCL, N>1) -> RMSprop step.
Workload = BASELINE.json configs[1]: per-GPU batch 256, 128 ch x 500 samples, DINOv2 dim 384,
bf16 MFMA operands with f32 accumulate/state.  Weak scaling: per-GPU work is fixed.
Rank 0 prints ONE JSON line (metric, roofline of the dominant kernel, CPU baseline).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16, MI355X_MICROARCH.md "Peak BF16/FP16 MFMA"
HBM_PEAK_GBS = 8000.0            # HBM3E spec, same guide
# committed rocprofv3 --pmc passes per workload (tools/pmc_traffic.sh: FETCH_SIZE / WRITE_SIZE; tools/pmc_counters.sh:
# MFMA-busy / wait / L2-hit / LDS-conflict), each carrying the kernel-source fingerprint it was collected with
PMC_FILES = {"cfg2": ("r04_pmc_traffic.json", "r04_pmc_counters.json"),
             "cfg4": ("r04_cfg4_pmc_traffic.json", "r04_cfg4_pmc_counters.json"),
             "cfg2/f32": ("r04_f32_pmc_traffic.json", "r04_f32_pmc_counters.json")}      # (--dtype f32: the exact-float32 path)


def csrc_sha16():
    """Fingerprint of the kernel sources; the committed PMC passes record the one they were collected with."""
    import glob
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "cerebralsignalnetworks_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def seeded_params(C, H, L, D, seed=43):
    """= oracle.lstm.init_params(seed=43) (the weights every fixture under tests/golden was made with), restated:
    the oracle is test infrastructure and is not imported by the measured path."""
    rng = np.random.default_rng(seed)
    k = 1.0 / np.sqrt(H)
    sd = {}
    for l in range(L):
        i_sz = C if l == 0 else H
        sd[f"lstm.weight_ih_l{l}"] = rng.uniform(-k, k, (4 * H, i_sz)).astype(np.float32)
        sd[f"lstm.weight_hh_l{l}"] = rng.uniform(-k, k, (4 * H, H)).astype(np.float32)
        sd[f"lstm.bias_ih_l{l}"] = rng.uniform(-k, k, (4 * H,)).astype(np.float32)
        sd[f"lstm.bias_hh_l{l}"] = rng.uniform(-k, k, (4 * H,)).astype(np.float32)
    sd["fc.weight"] = rng.uniform(-k, k, (D, H)).astype(np.float32)
    sd["fc.bias"] = rng.uniform(-k, k, (D,)).astype(np.float32)
    return {n: torch.from_numpy(v) for n, v in sd.items()}


def fixture_parity(tag, device):
    """|distill loss - the reference's| for the exact-f32 path and the benchmarked bf16 path, on the committed fixture
    of the reference's own LSTMModel + CosineSimilarityLoss (tests/golden/ref_lstm_<tag>.npz: 8 segments, torch CPU
    f64 / f32); batch 256 = 32 copies of the 8 segments, so the kernels and the launch shapes are the timed ones."""
    from cerebralsignalnetworks_amd import Model, CosineSimilarityLoss
    from cerebralsignalnetworks_amd.trainer import check_device_status
    g = np.load(os.path.join(ROOT, "tests", "golden", f"ref_lstm_{tag}.npz"), allow_pickle=False)
    B8, T, C, H, L, D = (int(v) for v in g["dims"])
    rng = np.random.default_rng(int(g["seed_x"]))
    x8 = rng.standard_normal((B8, T, C)).astype(np.float32)
    t8 = rng.standard_normal((B8, D)).astype(np.float32)
    x = torch.from_numpy(np.tile(x8, (32, 1, 1))).to(device)
    tg = torch.from_numpy(np.tile(t8, (32, 1))).to(device)
    sd = seeded_params(C, H, L, D, int(g["seed_params"]))
    out = {"fixture": f"tests/golden/ref_lstm_{tag}.npz (reference LSTMModel + CosineSimilarityLoss, torch CPU)",
           "loss_reference_f64": float(g["loss_f64"]), "loss_reference_f32": float(g["loss_f32"])}
    for name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        m = Model(input_size=C, lstm_size=H, lstm_layers=L, output_size=D, include_top=False, compute_dtype=dt)
        m.load_state_dict(sd)
        m = m.to(device)
        feat = m(x)
        loss = CosineSimilarityLoss()(feat, tg)
        loss.backward()
        check_device_status(m, collective=False)      # (rank 0 alone runs this: no collective here)
        gn = float(m.lstm.weight_hh_l0.grad.double().norm())
        out[name] = {"loss": float(loss.item()), "abs_err_vs_f64": abs(float(loss.item()) - float(g["loss_f64"])),
                     "feat_max_abs_err": float(np.abs(feat.detach().cpu().numpy()[:B8] - g["feat_f64"]).max()),
                     "grad_norm_rel_err_w_hh_l0": abs(gn - float(g["gnorm__lstm.weight_hh_l0"])) / float(g["gnorm__lstm.weight_hh_l0"])}
        del m, feat, loss
    out["f32_within_1e-4"] = out["f32"]["abs_err_vs_f64"] < 1e-4
    torch.cuda.empty_cache()
    return out


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch (cfg2: 256)")
    ap.add_argument("--channels", type=int, default=128)
    ap.add_argument("--samples", type=int, default=500)
    ap.add_argument("--hidden", type=int, default=768)
    ap.add_argument("--layers", type=int, default=2)
    ap.add_argument("--dim", type=int, default=384)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--pool", type=int, default=4, help="resident synthetic batches per GPU")
    ap.add_argument("--config", default="cfg2", choices=["cfg2", "cfg4"],
                    help="cfg2 = BASELINE.json configs[1] (the headline); cfg4 = Spampinato shapes 128 x 440, hidden 1024")
    ap.add_argument("--loss", default="cosine", choices=["cosine", "barlow"],
                    help="cosine = the headline distillation step; barlow = BASELINE.json configs[4]: Barlow-Twins "
                         "cross-correlation loss on the LSTM embeddings (HIP off-diagonal reduction) + LARS")
    ap.add_argument("--no-retrieval", action="store_true", help="skip the bf16-vs-CPU-reference retrieval acceptance")
    ap.add_argument("--no-f32-line", action="store_true", help="skip the short measurement of the exact-f32 path")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the fixture check of the f32 / bf16 loss before the timed region")
    ap.add_argument("--no-kernel-timing", action="store_true")
    return ap.parse_args(argv)


def launcher_command(n_gpus, argv, port=None):
    """argv + environment of the child job that runs N ranks of this file, one per GPU (the shape of the reference's
    own launch, EEG-BarlowNetworks/train.py:71,76-78: one worker per GPU, rank = GPU index, tcp rendezvous on the
    loopback address).  The parent that calls this never touches the GPU."""
    import socket
    if port is None:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on these hosts (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "4")
    if env.get("CSN_SINGLE_DEVICE"):
        # rehearsal: N ranks on ONE device.  A weight-stationary launch needs every CU of the device for itself (one
        # workgroup per CU, all co-resident): two ranks' launches would wait for each other until the bounded spins give
        # up -- the per-diagonal launches are the form that can share a device
        env.setdefault("CSN_NO_PERSIST", "1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return cmd, env


def launch_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher around it: start N fresh rank processes and relay rank 0's line.
    Nothing here initialises HIP (device_count() does not on this image); the children are new processes, not an
    exec of this one."""
    import subprocess
    have = torch.cuda.device_count()
    if have < args.gpus and not os.environ.get("CSN_SINGLE_DEVICE"):
        print(f"bench.py: --gpus {args.gpus} asked for but this host shows {have} GPU(s); refusing to report a "
              f"{args.gpus}-GPU number from fewer devices", file=sys.stderr)
        return 2
    cmd, env = launcher_command(args.gpus, argv)
    log("launcher: " + " ".join(cmd))
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    for ln in proc.stdout.splitlines():
        if not ln.startswith("{"):
            print(ln, file=sys.stderr)
    if proc.returncode != 0 or len(lines) != 1:
        print(f"bench.py: the {args.gpus}-rank job exited with {proc.returncode} and {len(lines)} result line(s)",
              file=sys.stderr)
        return proc.returncode or 3
    if json.loads(lines[0]).get("n_gpus") != args.gpus:
        print("bench.py: result line does not carry the requested rank count", file=sys.stderr)
        return 4
    print(lines[0], flush=True)
    return 0


def synthetic_pool(n, C, T, D, rank, device):
    """x = N(0,1) + 0.5 sin(2 pi 40 t / 1000) (utils/GenerateRandomEEGNoise.py:4-19), seeds 43/44/45."""
    g = torch.Generator(device="cpu").manual_seed(43 + 1000 * rank)
    t = torch.arange(T, dtype=torch.float32) / 1000.0
    x = torch.randn(n, C, T, generator=g) + 0.5 * torch.sin(2 * np.pi * 40 * t)
    tg = torch.randn(n, D, generator=torch.Generator().manual_seed(44 + 1000 * rank))
    lab = torch.randint(0, 40, (n,), generator=torch.Generator().manual_seed(45 + 1000 * rank))
    return x.to(device), tg.to(device), lab.to(device)


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def retrieval_acceptance(model_bf16, filt, device, tag="cfg2"):
    """North star: 'retrieval top-1 within +-0.5 % of the CPU reference'.  The seeded clustered set (2048 gallery /
    512 query, 40 classes) is embedded with the benchmarked bf16 path (HIP filter -> HIP LSTM) and searched with
    csn_l2_topk; the CPU reference's neighbour lists for the same set and the same weights (scipy sosfilt + z-score ->
    the reference's LSTMModel on torch CPU f32, made by tests/golden/make_ref_goldens.py) are a committed fixture."""
    from cerebralsignalnetworks_amd import cabi
    from cerebralsignalnetworks_amd.dataset import clustered_eeg
    g = np.load(os.path.join(ROOT, "tests", "golden", f"ref_retrieval_{tag}.npz"), allow_pickle=False)
    ng, nq = int(g["n_gallery"]), int(g["n_query"])
    T = int(g["dims"][1]) if "dims" in g.files else 500
    x, labels = clustered_eeg(ng + nq, T=T, seed=int(g["seed"]), snr=float(g["snr"]))
    outs = []
    with torch.no_grad():
        for i in range(0, ng + nq, 256):
            outs.append(model_bf16(filt.apply(torch.from_numpy(x[i:i + 256]).to(device))).float())
    emb = torch.cat(outs)
    _, idx = cabi.l2_topk(emb[:ng].contiguous(), emb[ng:].contiguous(), 5)
    idx = idx.cpu().numpy()
    top1 = float((labels[:ng][idx[:, 0]] == labels[ng:]).mean())
    ref = float(g["top1"])
    return {"set": f"{ng} gallery / {nq} query, 40 classes, seeded clustered EEG (snr {float(g['snr'])})",
            "top1_bf16": top1, "top1_cpu_reference": ref, "delta": top1 - ref, "within_half_percent": abs(top1 - ref) <= 0.005,
            "same_nearest_neighbour": float((idx[:, 0] == g["top5"][:, 0]).mean()),
            "top5_overlap": float(np.mean([len(set(a) & set(b)) / 5.0 for a, b in zip(idx, g["top5"])]))}


def main():
    args = parse()
    if args.config == "cfg4":      # LstmDistillFromDinoV2TrainSpampinato.py:368 shapes (BASELINE.json configs[3])
        args.samples, args.hidden = 440, 1024
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
              f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...)",
              file=sys.stderr)
        sys.exit(2)
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("CSN_SINGLE_DEVICE"):      # rehearsal of the multi-rank path on a one-GPU box
        local = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        backend = os.environ.get("CSN_DIST_BACKEND", "nccl")     # "nccl" is RCCL on ROCm; gloo only for rehearsals
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend=backend)
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)

    from cerebralsignalnetworks_amd import Model, EEGFilters
    from cerebralsignalnetworks_amd.trainer import DistillTrainer

    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    B, C, T, H, L, D = args.batch, args.channels, args.samples, args.hidden, args.layers, args.dim
    torch.manual_seed(43)
    model = Model(input_size=C, lstm_size=H, lstm_layers=L, output_size=D, include_top=False,
                  compute_dtype=dtype).to(device)
    filt = EEGFilters(1000, order=3)
    if args.loss == "barlow":     # EEG-BarlowNetworks/train.py: LARS, lr from the warm-up / cosine schedule (its peak at batch 256 x world)
        trainer = DistillTrainer(model, filt.sos, loss="barlow", lr=0.2, optimizer="lars")
        args.no_parity = args.no_retrieval = args.no_f32_line = True      # (those companions describe the cosine step)
    else:
        trainer = DistillTrainer(model, filt.sos, loss="cosine", lr=1e-3, optimizer="rmsprop")
    x, tg, lab = synthetic_pool(B * args.pool, C, T, D, rank, device)
    parity = None
    ptag = {(128, 500, 768, 2, 384): "cfg2", (128, 440, 1024, 2, 384): "cfg4"}.get((C, T, H, L, D))
    if rank == 0 and ptag is not None and not args.no_parity:
        log("parity: f32 and bf16 paths on the reference-made fixture (before the timed region)")
        parity = fixture_parity(ptag, device)

    def step(i):
        j = (i % args.pool) * B
        return trainer.train_step(x[j:j + B], tg[j:j + B], lab[j:j + B])

    log(f"rank {rank}/{world}: pool resident, {args.warmup} warm-up steps")
    for i in range(args.warmup):
        step(i)
        torch.cuda.synchronize()
        log(f"warm-up step {i} done")
    from cerebralsignalnetworks_amd import cabi
    prof = []
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loss = None
    step_losses = []
    if world > 1:
        trainer.grads.timing = []
    for i in range(args.steps):
        if i == args.steps - 1 and rank == 0 and not args.no_kernel_timing:
            # HIP events around every recurrence launch (same stream), in the LAST timed step only: recorded in
            # every step they cost 0.25 ms per step (70 event records between dependent launches)
            train_plan = [pl for pl in model.lstm.all_plans() if pl.training][0]
            train_plan.profile_enable(True)
        loss = step(args.warmup + i)
        step_losses.append(loss)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    final_loss = float(loss.item())
    dp = None
    if world > 1:
        # proof that N ranks ran and stayed in step: every rank adds one; after K all-reduced steps from a broadcast
        # start every rank must hold the same parameter bits (the checksum is an exact integer sum of the f32 words)
        ones = torch.ones(1, device=device, dtype=torch.float64)
        dist.all_reduce(ones)
        pbuf = trainer.grads.flat_params if trainer.grads.flat_params is not None else torch.cat(
            [p.detach().reshape(-1) for p in trainer.grads.params])
        csum = pbuf.view(torch.int32).to(torch.int64).sum().reshape(1)
        lo, hi = csum.clone(), csum.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        dp = {"ranks_seen": int(ones.item()), "param_checksum": int(csum.item()),
              "param_checksum_equal_on_all_ranks": bool(lo.item() == hi.item()),
              "allreduce_exposed_ms_per_step": trainer.grads.all_reduce_ms(),      # end of the backward's launches -> reduced, scaled gradients
              "allreduce_ms_per_step": None if trainer.grads.segments is not None else trainer.grads.all_reduce_ms(),   # (blocking form: the whole collective)
              "allreduce_overlapped": trainer.grads.segments is not None,
              "allreduce_segments_bytes": [4 * (e - s_) for s_, e in (trainer.grads.segments or [(0, trainer.grads.flat.numel())])],
              "allreduce_bytes": trainer.grads.flat.numel() * 4,
              "backend": dist.get_backend(), "single_device_rehearsal": bool(os.environ.get("CSN_SINGLE_DEVICE"))}
        if dp["ranks_seen"] != args.gpus or not dp["param_checksum_equal_on_all_ranks"]:
            print(f"bench.py: rank {rank}: data-parallel check failed: {dp}", file=sys.stderr)
            sys.exit(5)
    trainer.check_device_status()       # a timed-out in-kernel hand-off would invalidate the run: fail loudly
    log(f"timed region done: {elapsed:.3f}s for {args.steps} steps; losses " +
        " ".join(f"{float(l):.4f}" for l in step_losses))

    if rank == 0:
        seg_per_s = world * B * args.steps / elapsed
        flops_per_seg = 3.0 * 2.0 * T * sum(4 * H * ((C if l == 0 else H) + H) for l in range(L))
        res = {
            "metric": f"EEG-segments/sec training ({C}ch x {T}, hidden={H})" + (", Barlow-Twins loss" if args.loss == "barlow" else ""),
            "value": seg_per_s, "unit": "segments/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": (("cfg5 (1-GPU leg)" if args.loss == "barlow" else "cfg2") if (B, C, T, H, L, D) == (256, 128, 500, 768, 2, 384) else
                                    "cfg4" if (C, T, H, L, D) == (128, 440, 1024, 2, 384) else "custom") +
                                   (": fused EEG band-pass+z-score -> LSTM fwd/bwd -> Barlow-Twins cross-correlation loss "
                                    "(BatchNorm, c = z1^T z2 / batch, HIP off-diagonal reduction) -> LARS, random image-embedding view"
                                    if args.loss == "barlow" else
                                    ": fused EEG band-pass+z-score -> LSTM fwd/bwd -> cosine distill "
                                    "-> RMSprop, precomputed random DINOv2-dim targets"),
                       "per_gpu_batch": B, "global_batch": B * world, "channels": C, "samples": T, "hidden": H,
                       "layers": L, "embed_dim": D, "parallelism": f"dp{world}"},
            "final_loss": final_loss,
            "model_tflops": seg_per_s * flops_per_seg / 1e12,
        }
        if dp is not None:
            res["data_parallel"] = dp
            if dp["single_device_rehearsal"]:
                # N ranks shared one device: plumbing evidence only, never a throughput number
                res["rehearsal_value"], res["value"], res["vs_baseline"] = res["value"], None, None
                res["note"] = "CSN_SINGLE_DEVICE rehearsal: all ranks on one GPU, per-diagonal launches; `value` withheld"
        if parity is not None:
            res["parity"] = parity
        if not args.no_kernel_timing:
            # dominant kernel = the recurrence kernel with the larger total time in the step; its average launch
            # duration comes from HIP events recorded on the launch stream around its launches of the LAST timed step
            pr = train_plan.profile_read()
            peak = MFMA_BF16_PEAK_TFLOPS if args.dtype == "bf16" else 157.3
            us = {k: (1e3 * pr[k + "_ms"] / max(1, pr[k + "_launches"])) for k in ("fwd", "bwd")}
            dom = "bwd" if pr["bwd_ms"] >= pr["fwd_ms"] else "fwd"      # largest total time in the step
            knames = dict(zip(("fwd", "bwd"), train_plan.kernel_names()))      # from the plan's path (csn_lstm_plan_kernel_name)
            persist = {k: "persist" in knames[k] or "_ns_" in knames[k] for k in ("fwd", "bwd")}     # weight-stationary: one launch per chunk
            kname = knames[dom]
            cells_per_launch = pr[dom + "_cells"] / max(1, pr[dom + "_launches"])
            # algorithmic work of one cell problem (one layer, one timestep):
            #   flops: the recurrent product 2 * B * 4H * H
            #   HBM bytes, backward: read gates (bf16 4H) + c_{t-1} (f32 H) + dy (f32 H), write dgates (bf16 4H)
            #   HBM bytes, forward : read the input projection (f32 4H), write gates (bf16 4H) + c (f32 H) + h (bf16 H)
            flops_per_launch = 2.0 * B * 4 * H * H * cells_per_launch
            bytes_per_cell = B * H * (24.0 if dom == "bwd" else 30.0)
            f32_ws = args.dtype != "bf16" and persist[dom]
            if f32_ws:
                # float32 weight-stationary recurrence (lstm_f32_persist.hip), per (row, unit) and step:
                #   backward: read gates 16 + c, c_prev, dy 12, write dgates row-major 16 + fragment-major 16 = 60 bytes
                #   forward : read the projection 16, write gates 16 + c 4 + h row-major 4 + fragment-major 4 = 44 bytes
                bytes_per_cell = B * H * (60.0 if dom == "bwd" else 44.0)
            bytes_per_launch = bytes_per_cell * cells_per_launch
            fused_fwd = args.dtype == "bf16" and persist["fwd"] and C == 128 and H != 512 and not os.environ.get("CSN_NO_FUSE_X")
            fused_x = dom == "fwd" and fused_fwd
            if fused_x:
                # layer 0 multiplies x_t itself (make_layout's fuse_x in lstm.hip): its cells read x (bf16, C per row) instead of
                # a float32 projection (4H per row) and carry the projection's flops; 1 / L of a launch's cells are layer 0's
                share0 = cells_per_launch / L
                bytes_per_launch += share0 * B * (2.0 * C - 16.0 * H)
                flops_per_launch += share0 * 2.0 * B * 4 * H * C
            # (lstm.hip:backward_persist's own condition: the launch has idle workgroups only while a group is <= 28 slices
            #  of 32 units, i.e. H <= 896 -- at H = 1024 the 8 groups fill all 256 CUs and the GEMM is a kernel of its own)
            beside = (args.dtype == "bf16" and dom == "bwd" and persist["bwd"] and 1 < L <= 4 and H // 32 <= 28
                      and not os.environ.get("CSN_NO_BESIDE"))
            if beside:
                # the backward launches also carry the input-gradient GEMMs of the layers above layer 0 on their idle
                # workgroups: dx[T*B, H] = dgates[T*B, 4H] W_ih -> 2*T*B*4H*H flops, read dgates bf16, write dx f32
                n_l = max(1, pr["bwd_launches"])
                flops_per_launch += (L - 1) * 2.0 * T * B * 4 * H * H / n_l
                bytes_per_launch += (L - 1) * T * B * (4 * H * 2.0 + H * 4.0) / n_l
            t_launch = us[dom] * 1e-6
            if t_launch <= 0.0:
                # (the per-step float32 path records no launch events: its line is the headline's `f32_path` companion)
                res["roofline"] = None
                res["roofline_note"] = "no per-launch timing on this path (per-step cell kernels); see profiles/ and DESIGN.md section 6"
                t_launch = None
        if not args.no_kernel_timing and t_launch is not None:
            ach_tf = flops_per_launch / t_launch / 1e12
            ach_gb = bytes_per_launch / t_launch / 1e9
            # the binding roofline is the one with the larger minimum time
            hbm_bound = bytes_per_launch / (HBM_PEAK_GBS * 1e9) >= flops_per_launch / (peak * 1e12)
            # from the committed PMC passes of the same workload (profiles/): NOT measured in this run -- the source is
            # named beside the values, and they are dropped when the kernel sources changed since the pass
            traffic = mfma_util = None
            traffic_source = None
            try:
                wl = {(256, 128, 500, 768, 2): "cfg2", (256, 128, 440, 1024, 2): "cfg4"}.get((B, C, T, H, L))
                if wl is not None and args.dtype != "bf16":
                    wl = wl + "/f32" if wl + "/f32" in PMC_FILES else None
                if wl is None:
                    raise KeyError("PMC passes are committed for the cfg2 and cfg4 workloads (bf16) and cfg2 (float32) only")
                PMC_TRAFFIC_FILE, PMC_COUNTERS_FILE = PMC_FILES[wl]
                pmc = json.load(open(os.path.join(ROOT, "profiles", PMC_TRAFFIC_FILE)))
                cnt = json.load(open(os.path.join(ROOT, "profiles", PMC_COUNTERS_FILE)))
                now = csrc_sha16()
                if pmc.get("csrc_sha16") != now or cnt.get("csrc_sha16") != now:
                    traffic_source = (f"profiles/{PMC_TRAFFIC_FILE}, profiles/{PMC_COUNTERS_FILE}: STALE (collected with "
                                      f"kernel sources {pmc.get('csrc_sha16')}, this build is {now}) -- values dropped")
                else:
                    traffic = pmc["kernels"][kname]["hbm_bytes_per_launch_corrected"]
                    mfma_util = cnt["kernels"][kname]["derived"]["mfma_util"]
                    traffic_source = (f"profiles/{PMC_TRAFFIC_FILE} + profiles/{PMC_COUNTERS_FILE}: committed rocprofv3 --pmc "
                                      f"passes of this workload (kernel sources {now}); not measured in this run")
            except (OSError, KeyError, ValueError) as e:
                traffic_source = f"none ({e})"
            res["roofline"] = {"bound": "hbm" if hbm_bound else "mfma", "kernel": kname,
                               "achieved": ach_gb if hbm_bound else ach_tf,
                               "peak": HBM_PEAK_GBS if hbm_bound else peak,
                               "unit": "GB/s" if hbm_bound else "TFLOP/s",
                               "frac": (ach_gb / HBM_PEAK_GBS) if hbm_bound else (ach_tf / peak), "traffic": traffic,
                               "mfma_util": mfma_util, "traffic_source": traffic_source,
                               "algorithmic_bytes_per_launch": bytes_per_launch, "flops_per_launch": flops_per_launch,
                               "other_roofline": {"bound": "mfma" if hbm_bound else "hbm",
                                                  "achieved": ach_tf if hbm_bound else ach_gb,
                                                  "frac": (ach_tf / peak) if hbm_bound else (ach_gb / HBM_PEAK_GBS)},
                               "us_per_launch": us, "launches": {k: pr[k + "_launches"] for k in ("fwd", "bwd")},
                               "cells_per_launch": cells_per_launch, "input_gradient_gemm_in_launch": bool(beside),
                               "layer0_projection_in_launch": bool(fused_x), "layer0_projection_in_forward_launch": bool(fused_fwd),
                               "note": "recurrent GEMM chain with one hand-off between workgroups per timestep; neither "
                                       "roofline binds: the step is paced by the per-step operand stream from L2 and "
                                       "the hand-off latency (DESIGN.md section 3)"}
        rtag = {(128, 500, 768, 2, 384): "cfg2", (128, 440, 1024, 2, 384): "cfg4"}.get((C, T, H, L, D))
        if world == 1 and args.dtype == "bf16" and rtag is not None and not args.no_retrieval:
            # same architecture with the fixture's seeded weights (the timed model's weights have been trained on noise)
            from cerebralsignalnetworks_amd.trainer import check_device_status
            log("retrieval acceptance (bf16 path vs CPU-reference fixture)")
            rm = Model(input_size=C, lstm_size=H, lstm_layers=L, output_size=D, include_top=False, compute_dtype=dtype)
            rm.load_state_dict(seeded_params(C, H, L, D))
            rm = rm.to(device).eval()
            res["retrieval"] = retrieval_acceptance(rm, filt, device, rtag)
            check_device_status(rm)
            del rm
        if world == 1 and args.dtype == "bf16" and not args.no_f32_line:
            # the exact-f32 path (the one held to "distill loss within 1e-4"): a short measurement beside the headline
            log("exact-f32 path: 1 warm-up + 10 timed steps")
            m32 = Model(input_size=C, lstm_size=H, lstm_layers=L, output_size=D, include_top=False,
                        compute_dtype=torch.float32).to(device)
            t32 = DistillTrainer(m32, filt.sos, loss="cosine", lr=1e-3, optimizer="rmsprop")
            t32.train_step(x[:B], tg[:B], lab[:B])
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(10):
                t32.train_step(x[:B], tg[:B], lab[:B])
            torch.cuda.synchronize()
            dt32 = (time.perf_counter() - t1) / 10
            # floor of ANY exact-f32 MFMA path: the step's 22.4 GFLOP / segment at the 157.3 TFLOP/s f32 matrix peak
            k32 = t32.model.lstm.all_plans()[0].kernel_names()
            res["f32_path"] = {"value": B / dt32, "unit": "segments/s", "ms_per_step": 1e3 * dt32, "steps": 10,
                               "mfma_f32_floor_ms": 1e3 * B * flops_per_seg / 157.3e12, "recurrence_kernels": list(k32),
                               "note": "compute_dtype=float32: exact-f32 MFMA (v_mfma_f32_16x16x4_f32), " +
                                       ("weight-stationary recurrence, one launch per layer" if "persist" in k32[0]
                                        else "per-step K-split cell kernels") +
                                       " + generic GEMMs: the path that holds every gradient to 1e-4 (parity path)"}
            del m32, t32
        if world == 1 and not args.no_cpu_baseline:
            from oracle import cpu_path, eeg_filter
            log(f"CPU baseline on {cpu_path.usable_cores()} cores")
            nb = 16                                   # batch 16, 1 warm-up + 3 timed steps
            xs = eeg_filter.synthetic_eeg(nb, C, T, seed=43)
            ts = np.random.default_rng(44).standard_normal((nb, D)).astype(np.float32)
            cb = cpu_path.time_cpu_train_steps(xs, ts, filt.sos, hidden=H, layers=L, steps=3, warmup=1)
            t1 = time.perf_counter()
            for _ in range(3):
                cpu_path.preprocess_scipy(xs, filt.sos)
            pre = 3 * nb / (time.perf_counter() - t1)
            res["cpu_baseline"] = {"value": cb["seg_per_s"], "unit": "segments/s", "cores": cb["cores"],
                                   "kind": "port", "preprocessing_only_seg_per_s": pre,
                                   "sample": f"batch {nb}, 1 warm-up + 3 timed steps of scipy sosfilt + z-score -> "
                                             f"torch.nn.LSTM({C}->{H}x{L}) fp32 -> Linear -> cosine -> backward -> RMSprop "
                                             f"(torch threads = cores); preprocessing alone: scipy sosfilt + z-score, 1 thread"}
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
