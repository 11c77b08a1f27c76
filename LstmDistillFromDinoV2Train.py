#!/usr/bin/env python3
"""MI355X trainer with the command-line / checkpoint surface of the reference's
LstmDistillFromDinoV2Train.py (flags :150-225, loop :351-424, checkpoint :411-419).

    python LstmDistillFromDinoV2Train.py --synthetic 256 --batch_size 16 --num_epochs 1
    torchrun --nproc-per-node 8 LstmDistillFromDinoV2Train.py --synthetic 65536 --batch_size 256 ...

Differences that the hardware forces, all opt-in or documented in DESIGN.md: data and frozen
teacher embeddings are resident on the GPU; one process per GPU with RCCL gradient all-reduce
(the reference initialises gloo and never shards, :234,286); ``--hyperprams`` is parsed with
``ast.literal_eval`` (the reference ``eval``s it, :247); DINOv2 embeddings come from
``--teacher_features`` (.npy) or are synthetic -- ``torch.hub`` (:144-146) needs a network.
"""
import argparse
import ast
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def build_parser():
    p = argparse.ArgumentParser('LSTM distillation from DINOv2 embeddings (MI355X).')
    p.add_argument('--learning_rate', type=float, default=0.001)
    p.add_argument('--num_epochs', type=int, default=100)
    p.add_argument('--batch_size', type=int, default=16, help='per-process batch size')
    p.add_argument('--log_dir', type=str, default='./logs/DinoV2LstmDistillv2sdsad/')
    p.add_argument('--gallery_subject', type=int, default=1, choices=[0, 1, 2, 3, 4, 5, 6])
    p.add_argument('--query_subject', type=int, default=1, choices=[0, 1, 2, 3, 4, 5, 6])
    p.add_argument('--eeg_dataset', type=str,
                   default="./data/eeg/theperils/spampinato-1-IMAGE_BLOCK_RAW_with_mean_std.pth")
    p.add_argument('--images_root', type=str, default="./data/images/imageNet_images")
    p.add_argument('--eeg_dataset_split', type=str, default="./data/eeg/block_splits_by_image_all.pth")
    p.add_argument('--mode', type=str, default="train")
    p.add_argument('--custom_model_weights', type=str, default="")
    p.add_argument('--search_gallery', type=str, default="train")
    p.add_argument('--query_gallery', type=str, default="test")
    p.add_argument('--topK', type=int, default=5)
    p.add_argument('--gallery_tranformation_type', type=str, default="eeg2eeg", choices=["img", "img2eeg", "eeg", "eeg2eeg"])
    p.add_argument('--query_tranformation_type', type=str, default="eeg2eeg", choices=["img", "img2eeg", "eeg", "eeg2eeg"])
    p.add_argument('--hyperprams', type=str,
                   default="{'ce_loss_weight': 0.50, 'soft_target_loss_weight':0.50,'alpha': 1,'temperature':2}")
    p.add_argument('--seed', default=43, type=int)
    p.add_argument('--num_workers', default=4, type=int)
    p.add_argument("--dist_url", default="env://", type=str)
    p.add_argument("--local_rank", default=0, type=int)
    # --- additions (SURVEY.md section 5 "Config / flags") ---
    p.add_argument('--synthetic', type=int, default=0, help='train on N synthetic 128x500 segments')
    p.add_argument('--teacher_features', type=str, default="", help='.npy [N,D] precomputed frozen-teacher embeddings')
    p.add_argument('--hidden_size', type=int, default=96, help='lstm_size (reference call site: 96)')
    p.add_argument('--lstm_layers', type=int, default=2)
    p.add_argument('--output_size', type=int, default=384)
    p.add_argument('--loss', type=str, default="featdist", choices=["featdist", "cosine", "kd", "barlow"])
    p.add_argument('--dtype', type=str, default="bf16", choices=["bf16", "f32"])
    p.add_argument('--fs', type=float, default=1000.0, help='sampling rate for the band-pass design')
    p.add_argument('--filter_order', type=int, default=3, choices=[0, 3, 4, 5], help='0 = no band-pass')
    p.add_argument('--time_low', type=int, default=20)
    p.add_argument('--time_high', type=int, default=480)
    p.add_argument('--validation_frequency', type=int, default=5)
    p.add_argument('--compat_label_bug', action='store_true',
                   help='reproduce the batch-local label lookup of transformEEGDataLSTMByList')
    return p


class _KdParams:
    alpha = 0.5
    temperature = 2.0


def init_distributed():
    """One process per GPU; RCCL when launched by torchrun, single process otherwise
    (utils/utils.py:467-503 exits without a GPU -- so does this)."""
    if not torch.cuda.is_available():
        print('Does not support training without GPU.')
        sys.exit(1)
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) > 1:
        local = int(os.environ.get("LOCAL_RANK", 0))
        torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        return dist.get_rank(), dist.get_world_size(), local
    torch.cuda.set_device(0)
    return 0, 1, 0


def main(argv=None):
    from cerebralsignalnetworks_amd import Model, EEGFilters
    from cerebralsignalnetworks_amd.dataset import EEGDataset
    from cerebralsignalnetworks_amd.retrieval import evaluate_full
    from cerebralsignalnetworks_amd.trainer import DistillTrainer, shard_indices, split_indices
    from cerebralsignalnetworks_amd.losses import HyperParams

    FLAGS, _unparsed = build_parser().parse_known_args(argv)
    rank, world, local = init_distributed()
    is_main = rank == 0
    if is_main:
        print(FLAGS)
    os.makedirs(FLAGS.log_dir, exist_ok=True)
    hyper = ast.literal_eval(FLAGS.hyperprams)
    kd = _KdParams()
    kd.alpha = hyper.get("alpha", kd.alpha)
    kd.temperature = hyper.get("temperature", kd.temperature)
    torch.manual_seed(FLAGS.seed)
    device = torch.device("cuda", local)

    if FLAGS.synthetic:
        dataset = EEGDataset(synthetic=FLAGS.synthetic, time_low=0, time_high=500, seed=FLAGS.seed, device=device,
                             feature_dim=FLAGS.output_size, compat_label_bug=FLAGS.compat_label_bug)
    else:
        dataset = EEGDataset(eeg_signals_path=FLAGS.eeg_dataset, eeg_splits_path=None, imagesRoot=FLAGS.images_root,
                             time_low=FLAGS.time_low, time_high=FLAGS.time_high, device=device,
                             compat_label_bug=FLAGS.compat_label_bug)
        if not FLAGS.teacher_features:
            raise SystemExit("--teacher_features is required with a real dataset (torch.hub DINOv2 needs a network)")
        dataset.set_features(np.load(FLAGS.teacher_features, allow_pickle=False))
    N = len(dataset)
    features_length = dataset.features_all.shape[1]
    C = dataset.eeg_all.shape[1]

    train_idx, val_idx = (ix.to(device) for ix in split_indices(N, (0.8, 0.2), seed=43))      # random_split, :289-290

    dtype = torch.bfloat16 if FLAGS.dtype == "bf16" else torch.float32
    include_top = FLAGS.loss == "featdist"
    model = Model(input_size=C, lstm_size=FLAGS.hidden_size, lstm_layers=FLAGS.lstm_layers,
                  output_size=features_length, include_top=include_top, compute_dtype=dtype).to(device)
    if FLAGS.custom_model_weights:
        sd = torch.load(FLAGS.custom_model_weights, map_location="cpu", weights_only=True)
        model.load_state_dict(sd, strict=False)
    sos = EEGFilters(FLAGS.fs, order=FLAGS.filter_order).sos if FLAGS.filter_order else None
    trainer = DistillTrainer(model, sos, loss=FLAGS.loss, lr=FLAGS.learning_rate, optimizer="rmsprop",
                             nepochs=max(FLAGS.num_epochs, HyperParams.warmup_teacher_temp_epochs + 1), kd_params=kd)

    def batches(idx, epoch, shuffle):
        shard = idx[shard_indices(len(idx), epoch, FLAGS.seed, rank, world, shuffle=shuffle).to(device)]
        for s in range(0, len(shard), FLAGS.batch_size):
            yield shard[s:s + FLAGS.batch_size]

    def labels_of(ix):
        return [dataset.getLabelbyIndex(int(i)) for i in ix.cpu()]

    best_val_loss, best_val_loss_epoch, history = None, -1, []
    for EPOCH in range(FLAGS.num_epochs):
        losses = []
        for b in batches(train_idx, EPOCH, True):
            losses.append(trainer.train_step(dataset.eeg_all[b], dataset.features_all[b], dataset.labels_dev[b], EPOCH))
        trainer.check_device_status()      # per epoch: a timed-out in-kernel hand-off must not pass silently
        epoch_loss = float(torch.stack(losses).mean().item())                    # one sync per epoch, not per step
        if EPOCH % FLAGS.validation_frequency == 0 and EPOCH > 0:
            model.eval()
            gallery = trainer.embed_all(dataset.eeg_all[train_idx], FLAGS.batch_size)
            query = trainer.embed_all(dataset.eeg_all[val_idx], FLAGS.batch_size)
            vlosses = []
            with torch.no_grad():
                for b in batches(val_idx, 0, False):
                    out = model(trainer.embed(dataset.eeg_all[b]))
                    vlosses.append(trainer.compute_loss(out, dataset.features_all[b], dataset.labels_dev[b], EPOCH))
            val_epoch_loss = float(torch.stack(vlosses).mean().item())
            if is_main:
                r = evaluate_full(FLAGS, list(gallery.cpu().numpy()), list(query.cpu().numpy()), labels_of(train_idx),
                                  labels_of(val_idx), dataset)
                print(f"Overall Recall :{r['Recall_Total']} Overall Precision: {r['Precision_Total']} top1: {r['top1']:.4f}")
                if best_val_loss is None or val_epoch_loss < best_val_loss:
                    best_val_loss, best_val_loss_epoch = val_epoch_loss, EPOCH
                    torch.save(model.state_dict(), f"{FLAGS.log_dir}/lstm_dinov2_best_loss.pth")
                print(f"EPOCH {EPOCH} train_loss: {round(epoch_loss, 6)} val_loss: {round(val_epoch_loss, 6)} "
                      f"T: {HyperParams.T} best val loss: {best_val_loss} on epoch: {best_val_loss_epoch}")
        elif is_main:
            print(f"EPOCH {EPOCH} train_loss: {round(epoch_loss, 6)} T: {HyperParams.T}")
        history.append(epoch_loss)
    if is_main and best_val_loss is None:
        torch.save(model.state_dict(), f"{FLAGS.log_dir}/lstm_dinov2_best_loss.pth")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return history


if __name__ == "__main__":
    main()
