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


def parse():
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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    return ap.parse_args()


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


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
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
    trainer = DistillTrainer(model, filt.sos, loss="cosine", lr=1e-3, optimizer="rmsprop")
    x, tg, lab = synthetic_pool(B * args.pool, C, T, D, rank, device)

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
    trainer.check_device_status()       # a timed-out in-kernel hand-off would invalidate the run: fail loudly
    log(f"timed region done: {elapsed:.3f}s for {args.steps} steps; losses " +
        " ".join(f"{float(l):.4f}" for l in step_losses))

    if rank == 0:
        seg_per_s = world * B * args.steps / elapsed
        flops_per_seg = 3.0 * 2.0 * T * sum(4 * H * ((C if l == 0 else H) + H) for l in range(L))
        res = {
            "metric": "EEG-segments/sec training (128ch x 500, hidden=768)",
            "value": seg_per_s, "unit": "segments/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": ("cfg2" if (B, C, T, H, L, D) == (256, 128, 500, 768, 2, 384) else "custom") +
                                   ": fused EEG band-pass+z-score -> LSTM fwd/bwd -> cosine distill "
                                   "-> RMSprop, precomputed random DINOv2-dim targets",
                       "per_gpu_batch": B, "global_batch": B * world, "channels": C, "samples": T, "hidden": H,
                       "layers": L, "embed_dim": D, "parallelism": f"dp{world}"},
            "final_loss": final_loss,
            "model_tflops": seg_per_s * flops_per_seg / 1e12,
        }
        if not args.no_kernel_timing:
            # dominant kernel = the recurrence kernel with the larger total time in the step; its average launch
            # duration comes from HIP events recorded on the launch stream around its launches of the LAST timed step
            pr = train_plan.profile_read()
            peak = MFMA_BF16_PEAK_TFLOPS if args.dtype == "bf16" else 157.3
            us = {k: (1e3 * pr[k + "_ms"] / max(1, pr[k + "_launches"])) for k in ("fwd", "bwd")}
            dom = "bwd" if pr["bwd_ms"] >= pr["fwd_ms"] else "fwd"      # largest total time in the step
            persist = {k: pr[k + "_launches"] < T for k in ("fwd", "bwd")}     # weight-stationary: one launch per chunk
            kname = ("lstm_bwd_persist_kernel" if persist["bwd"] else "lstm_cell_bwd_il_kernel") if dom == "bwd" else (
                "lstm_fwd_persist_kernel" if persist["fwd"] else "lstm_cell_fwd_il_kernel")
            cells_per_launch = pr[dom + "_cells"] / max(1, pr[dom + "_launches"])
            # algorithmic work of one cell problem (one layer, one timestep):
            #   flops: the recurrent product 2 * B * 4H * H
            #   HBM bytes, backward: read gates (bf16 4H) + c_{t-1} (f32 H) + dy (f32 H), write dgates (bf16 4H)
            #   HBM bytes, forward : read the input projection (f32 4H), write gates (bf16 4H) + c (f32 H) + h (bf16 H)
            flops_per_launch = 2.0 * B * 4 * H * H * cells_per_launch
            bytes_per_cell = B * H * (24.0 if dom == "bwd" else 30.0)
            bytes_per_launch = bytes_per_cell * cells_per_launch
            beside = dom == "bwd" and persist["bwd"] and L > 1 and not os.environ.get("CSN_NO_BESIDE")
            if beside:
                # the backward launches also carry the input-gradient GEMMs of the layers above layer 0 on their idle
                # workgroups: dx[T*B, H] = dgates[T*B, 4H] W_ih -> 2*T*B*4H*H flops, read dgates bf16, write dx f32
                n_l = max(1, pr["bwd_launches"])
                flops_per_launch += (L - 1) * 2.0 * T * B * 4 * H * H / n_l
                bytes_per_launch += (L - 1) * T * B * (4 * H * 2.0 + H * 4.0) / n_l
            t_launch = us[dom] * 1e-6
            ach_tf = flops_per_launch / t_launch / 1e12
            ach_gb = bytes_per_launch / t_launch / 1e9
            # the binding roofline is the one with the larger minimum time
            hbm_bound = bytes_per_launch / (HBM_PEAK_GBS * 1e9) >= flops_per_launch / (peak * 1e12)
            traffic = None      # HBM bytes per launch from the committed PMC passes (profiles/), same workload
            try:
                if (B, C, T, H, L) != (256, 128, 500, 768, 2):
                    raise KeyError("PMC passes were collected for cfg2 only")
                pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_e_pmc_traffic.json")))
                traffic = pmc["kernels"][kname]["hbm_bytes_per_launch_corrected"]
            except (OSError, KeyError, ValueError):
                pass
            res["roofline"] = {"bound": "hbm" if hbm_bound else "mfma", "kernel": kname,
                               "achieved": ach_gb if hbm_bound else ach_tf,
                               "peak": HBM_PEAK_GBS if hbm_bound else peak,
                               "unit": "GB/s" if hbm_bound else "TFLOP/s",
                               "frac": (ach_gb / HBM_PEAK_GBS) if hbm_bound else (ach_tf / peak), "traffic": traffic,
                               "algorithmic_bytes_per_launch": bytes_per_launch, "flops_per_launch": flops_per_launch,
                               "other_roofline": {"bound": "mfma" if hbm_bound else "hbm",
                                                  "achieved": ach_tf if hbm_bound else ach_gb,
                                                  "frac": (ach_tf / peak) if hbm_bound else (ach_gb / HBM_PEAK_GBS)},
                               "us_per_launch": us, "launches": {k: pr[k + "_launches"] for k in ("fwd", "bwd")},
                               "cells_per_launch": cells_per_launch, "input_gradient_gemm_in_launch": bool(beside),
                               "note": "recurrent GEMM chain with one hand-off between workgroups per timestep; neither "
                                       "roofline binds: the step is paced by the per-step operand stream from L2 and "
                                       "the hand-off latency (DESIGN.md section 3)"}
        if world == 1 and not args.no_cpu_baseline:
            from oracle import cpu_path, eeg_filter
            log(f"CPU baseline on {cpu_path.usable_cores()} cores")
            nb = 8
            xs = eeg_filter.synthetic_eeg(nb, C, T, seed=43)
            ts = np.random.default_rng(44).standard_normal((nb, D)).astype(np.float32)
            cb = cpu_path.time_cpu_train_steps(xs, ts, filt.sos, hidden=H, layers=L, steps=2, warmup=1)
            res["cpu_baseline"] = {"value": cb["seg_per_s"], "unit": "segments/s", "cores": cb["cores"],
                                   "kind": "port",
                                   "sample": f"batch {nb}, 1 warm-up + 2 timed steps of scipy sosfilt + z-score -> "
                                             f"torch.nn.LSTM({C}->{H}x{L}) fp32 -> Linear -> cosine -> backward -> RMSprop"}
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
