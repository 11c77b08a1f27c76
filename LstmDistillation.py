#!/usr/bin/env python3
"""DINO self-distillation of the LSTM encoder on temporal multi-crops -- MI355X counterpart of the
reference's LstmDistillation.py (views :518-565, step :567-615, checkpoint :634-646): student and
EMA teacher are ``Model(input, 128, 4 layers, 128, include_top=False)`` wrapped with a ``DINOHead``;
2 global crops of 300 samples + 4 local crops of 200 samples; DINOLoss with a centre all-reduce;
AdamW with cosine LR / weight-decay / momentum schedules; ``checkpoint.pth`` holds
``{student, teacher, optimizer, epoch, args, dino_loss}``.

    python LstmDistillation.py --synthetic 512 --batch_size_per_gpu 64 --epochs 2
"""
import argparse
import json
import os
import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def build_parser():
    p = argparse.ArgumentParser('DINO-LSTM')
    p.add_argument('--out_dim', default=384, type=int)
    p.add_argument('--norm_last_layer', default=True, type=lambda s: s.lower() in ("1", "true", "on"))
    p.add_argument('--momentum_teacher', default=0.996, type=float)
    p.add_argument('--use_bn_in_head', default=False, type=lambda s: s.lower() in ("1", "true", "on"))
    p.add_argument('--warmup_teacher_temp', default=0.04, type=float)
    p.add_argument('--teacher_temp', default=0.04, type=float)
    p.add_argument('--warmup_teacher_temp_epochs', default=30, type=int)
    p.add_argument('--weight_decay', type=float, default=0.04)
    p.add_argument('--weight_decay_end', type=float, default=0.4)
    p.add_argument('--clip_grad', type=float, default=3.0)
    p.add_argument('--batch_size_per_gpu', default=8, type=int)
    p.add_argument('--epochs', default=200, type=int)
    p.add_argument('--freeze_last_layer', default=1, type=int)
    p.add_argument('--lr', default=0.0005, type=float)
    p.add_argument('--warmup_epochs', default=10, type=int)
    p.add_argument('--min_lr', type=float, default=1e-06)
    p.add_argument('--local_crops_number', type=int, default=4)
    p.add_argument('--eeg_dataset', type=str, default="./data/eeg/theperils/spampinato-1-IMAGE_RAPID_RAW_with_mean_std.pth")
    p.add_argument('--images_root', type=str, default="./data/images/imageNet_images")
    p.add_argument('--log_dir', type=str, default='./logs/DinoLstm/')
    p.add_argument('--seed', default=43, type=int)
    p.add_argument('--saveckp_freq', default=10, type=int)
    p.add_argument('--num_workers', default=0, type=int)
    p.add_argument("--dist_url", default="env://", type=str)
    p.add_argument("--local_rank", default=0, type=int)
    # additions
    p.add_argument('--synthetic', type=int, default=0)
    p.add_argument('--embed_dim', type=int, default=128)
    p.add_argument('--lstm_layers', type=int, default=4)
    p.add_argument('--dtype', type=str, default="bf16", choices=["bf16", "f32"])
    p.add_argument('--time_low', type=int, default=0)
    p.add_argument('--time_high', type=int, default=495)
    return p


def main(argv=None):
    from LstmDistillFromDinoV2Train import init_distributed
    from cerebralsignalnetworks_amd import Model
    from cerebralsignalnetworks_amd.dataset import EEGDataset
    from cerebralsignalnetworks_amd.dino import (DINOHead, DINOLoss, MultiCropWrapper, cosine_scheduler, ema_update,
                                                 temporal_crops)
    from cerebralsignalnetworks_amd.trainer import FlatGrads, check_device_status, shard_indices, split_indices

    FLAGS, _ = build_parser().parse_known_args(argv)
    rank, world, local = init_distributed()
    device = torch.device("cuda", local)
    torch.manual_seed(FLAGS.seed)
    np.random.seed(FLAGS.seed)
    os.makedirs(FLAGS.log_dir, exist_ok=True)
    if FLAGS.synthetic:
        dataset = EEGDataset(synthetic=FLAGS.synthetic, synthetic_channels=96, synthetic_samples=512,
                             time_low=FLAGS.time_low, time_high=FLAGS.time_high, seed=FLAGS.seed, device=device)
    else:
        dataset = EEGDataset(eeg_signals_path=FLAGS.eeg_dataset, imagesRoot=FLAGS.images_root,
                             time_low=FLAGS.time_low, time_high=FLAGS.time_high, device=device)
    N, C = len(dataset), dataset.eeg_all.shape[1]
    train_idx = split_indices(N, (0.8, 0.2), seed=43)[0].to(device)      # random_split([0.8, 0.2], seed 43)

    dtype = torch.bfloat16 if FLAGS.dtype == "bf16" else torch.float32
    def make():
        backbone = Model(input_size=C, lstm_size=FLAGS.embed_dim, lstm_layers=FLAGS.lstm_layers,
                         output_size=FLAGS.embed_dim, include_top=False, compute_dtype=dtype)
        return backbone
    student = MultiCropWrapper(make(), DINOHead(FLAGS.embed_dim, FLAGS.out_dim, FLAGS.use_bn_in_head,
                                                FLAGS.norm_last_layer)).to(device)
    teacher = MultiCropWrapper(make(), DINOHead(FLAGS.embed_dim, FLAGS.out_dim, FLAGS.use_bn_in_head)).to(device)
    teacher.load_state_dict(student.state_dict())
    for p in teacher.parameters():
        p.requires_grad = False
    if world > 1:
        for p in student.parameters():
            dist.broadcast(p.data, src=0)
    grads = FlatGrads(student.parameters())

    dino_loss = DINOLoss(FLAGS.out_dim, FLAGS.local_crops_number + 2, FLAGS.warmup_teacher_temp, FLAGS.teacher_temp,
                         FLAGS.warmup_teacher_temp_epochs, FLAGS.epochs).to(device)
    regularized = [p for n, p in student.named_parameters() if p.requires_grad and not (n.endswith(".bias") or p.ndim == 1)]
    not_reg = [p for n, p in student.named_parameters() if p.requires_grad and (n.endswith(".bias") or p.ndim == 1)]
    optimizer = torch.optim.AdamW([{"params": regularized}, {"params": not_reg, "weight_decay": 0.}])
    per_rank = len(shard_indices(len(train_idx), 0, FLAGS.seed, rank, world))
    niter = max(1, per_rank // FLAGS.batch_size_per_gpu)                  # drop_last=True
    lr_schedule = cosine_scheduler(FLAGS.lr * (FLAGS.batch_size_per_gpu * world) / 256., FLAGS.min_lr, FLAGS.epochs,
                                   niter, warmup_epochs=min(FLAGS.warmup_epochs, FLAGS.epochs))
    wd_schedule = cosine_scheduler(FLAGS.weight_decay, FLAGS.weight_decay_end, FLAGS.epochs, niter)
    momentum_schedule = cosine_scheduler(FLAGS.momentum_teacher, 1, FLAGS.epochs, niter)

    history = []
    for EPOCH in range(FLAGS.epochs):
        student.train()
        shard = train_idx[shard_indices(len(train_idx), EPOCH, FLAGS.seed, rank, world).to(device)]
        losses = []
        for it_local in range(niter):
            it = niter * EPOCH + it_local
            for i, group in enumerate(optimizer.param_groups):
                group["lr"] = lr_schedule[it]
                if i == 0:
                    group["weight_decay"] = wd_schedule[it]
            b = shard[it_local * FLAGS.batch_size_per_gpu:(it_local + 1) * FLAGS.batch_size_per_gpu]
            eeg = dataset.eeg_all[b].transpose(1, 2).contiguous()                       # [B,T,C]
            gviews, lviews = temporal_crops(eeg, 2, FLAGS.local_crops_number)
            with torch.no_grad():
                teacher_outputs = torch.stack([teacher(v.contiguous()) for v in gviews], dim=0)
            student_outputs = torch.stack([student(v.contiguous()) for v in gviews + lviews], dim=0)
            loss = dino_loss(student_outputs, teacher_outputs, EPOCH)
            grads.zero()
            loss.backward()
            grads.all_reduce_mean()
            if FLAGS.clip_grad:
                for p in student.parameters():
                    if p.grad is not None:
                        clip_coef = FLAGS.clip_grad / (p.grad.norm(2) + 1e-6)
                        p.grad.mul_(torch.clamp(clip_coef, max=1.0))
            if EPOCH < FLAGS.freeze_last_layer:
                for n, p in student.named_parameters():
                    if "last_layer" in n and p.grad is not None:
                        p.grad.zero_()
            optimizer.step()
            ema_update(student, teacher, momentum_schedule[it])
            losses.append(loss.detach())
        epoch_loss = float(torch.stack(losses).mean().item())
        check_device_status(student)       # per epoch (sticky word): student and teacher share no plans
        check_device_status(teacher)
        history.append(epoch_loss)
        if rank == 0:
            save_dict = {'student': student.state_dict(), 'teacher': teacher.state_dict(),
                         'optimizer': optimizer.state_dict(), 'epoch': EPOCH + 1, 'args': vars(FLAGS),
                         'dino_loss': dino_loss.state_dict()}
            torch.save(save_dict, os.path.join(FLAGS.log_dir, 'checkpoint.pth'))
            if FLAGS.saveckp_freq and EPOCH % FLAGS.saveckp_freq == 0:
                torch.save(save_dict, os.path.join(FLAGS.log_dir, f'checkpoint{EPOCH:04}.pth'))
            with (Path(FLAGS.log_dir) / "log.txt").open("a") as f:
                f.write(json.dumps({"train_loss": epoch_loss, "train_lr": float(lr_schedule[it]), "epoch": EPOCH}) + "\n")
            print(f"Epoch: [{EPOCH}/{FLAGS.epochs}] loss: {epoch_loss:.6f} lr: {lr_schedule[it]:.6f}")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return history


if __name__ == "__main__":
    main()
