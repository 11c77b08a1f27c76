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


class Flavour:
    """What differs between the reference's two trainers of this loop.  `perils` = LstmDistillFromDinoV2Train.py
    (RMSprop 1e-3 :329, Model(128, 96, 2, include_top=True) + FeatureDistributionLoss :323-365, ONE checkpoint name
    :414,419); `spampinato` = LstmDistillFromDinoV2TrainSpampinato.py (AdamW 1e-4 :378, Model(128, 128, layers=4,
    include_top=False) :368, loss_fn_kd with alpha / temperature from --hyperprams :107-121,288-296, weights-only resume
    from --custom_model_weights when the file exists :369-371, first best checkpoint lstm_dinov2_epoch_{E}_best_loss.pth
    then lstm_dinov2_epochs_{EPOCHS}_best_loss.pth :467-475, utils/EEGDataset.py with subject + split file :310-317)."""

    def __init__(self, name, learning_rate, num_epochs, eeg_dataset, eeg_dataset_split, hyperprams, optimizer, loss,
                 hidden_size, lstm_layers, dataset_flavour):
        self.name, self.learning_rate, self.num_epochs = name, learning_rate, num_epochs
        self.eeg_dataset, self.eeg_dataset_split, self.hyperprams = eeg_dataset, eeg_dataset_split, hyperprams
        self.optimizer, self.loss, self.hidden_size, self.lstm_layers = optimizer, loss, hidden_size, lstm_layers
        self.dataset_flavour = dataset_flavour

    def checkpoint_path(self, log_dir, epoch, epochs, first_best):
        if self.name == "spampinato":
            return (f"{log_dir}/lstm_dinov2_epoch_{epoch}_best_loss.pth" if first_best
                    else f"{log_dir}/lstm_dinov2_epochs_{epochs}_best_loss.pth")
        return f"{log_dir}/lstm_dinov2_best_loss.pth"


PERILS = Flavour("perils", 0.001, 100, "./data/eeg/theperils/spampinato-1-IMAGE_BLOCK_RAW_with_mean_std.pth",
                 "./data/eeg/block_splits_by_image_all.pth",
                 "{'ce_loss_weight': 0.50, 'soft_target_loss_weight':0.50,'alpha': 1,'temperature':2}",
                 "rmsprop", "featdist", 96, 2, "perils")
SPAMPINATO = Flavour("spampinato", 0.0001, 200, "./data/eeg/spampinato/eeg_signals_raw_with_mean_std.pth",
                     "./data/eeg/spampinato/block_splits_by_image_all.pth",
                     "{'ce_loss_weight': 0.50, 'soft_target_loss_weight':0.50,'alpha': 0,'temperature':2}",
                     "adamw", "kd", 128, 4, "spampinato")


def build_parser(flavour=PERILS):
    p = argparse.ArgumentParser('LSTM distillation from DINOv2 embeddings (MI355X).')
    p.add_argument('--learning_rate', type=float, default=flavour.learning_rate)
    p.add_argument('--num_epochs', type=int, default=flavour.num_epochs)
    p.add_argument('--batch_size', type=int, default=16, help='per-process batch size')
    p.add_argument('--log_dir', type=str, default='./logs/DinoV2LstmDistillv2sdsad/')
    p.add_argument('--gallery_subject', type=int, default=1, choices=[0, 1, 2, 3, 4, 5, 6])
    p.add_argument('--query_subject', type=int, default=1, choices=[0, 1, 2, 3, 4, 5, 6])
    p.add_argument('--eeg_dataset', type=str, default=flavour.eeg_dataset)
    p.add_argument('--images_root', type=str, default="./data/images/imageNet_images")
    p.add_argument('--eeg_dataset_split', type=str, default=flavour.eeg_dataset_split)
    p.add_argument('--mode', type=str, default="train")
    p.add_argument('--custom_model_weights', type=str, default="")
    p.add_argument('--search_gallery', type=str, default="train")
    p.add_argument('--query_gallery', type=str, default="test")
    p.add_argument('--topK', type=int, default=5)
    p.add_argument('--gallery_tranformation_type', type=str, default="eeg2eeg", choices=["img", "img2eeg", "eeg", "eeg2eeg"])
    p.add_argument('--query_tranformation_type', type=str, default="eeg2eeg", choices=["img", "img2eeg", "eeg", "eeg2eeg"])
    p.add_argument('--hyperprams', type=str, default=flavour.hyperprams)
    p.add_argument('--seed', default=43, type=int)
    p.add_argument('--num_workers', default=4, type=int)
    p.add_argument("--dist_url", default="env://", type=str)
    p.add_argument("--local_rank", default=0, type=int)
    # --- additions (SURVEY.md section 5 "Config / flags") ---
    p.add_argument('--synthetic', type=int, default=0, help='train on N synthetic 128x500 segments')
    p.add_argument('--synthetic_samples', type=int, default=440 if flavour.name == "spampinato" else 500,
                   help='samples per synthetic segment (BASELINE.json: 128 x 500; Spampinato split 128 x 440)')
    p.add_argument('--teacher_features', type=str, default="", help='.npy [N,D] precomputed frozen-teacher embeddings')
    p.add_argument('--hidden_size', type=int, default=flavour.hidden_size, help='lstm_size (reference call sites: 96 / 128)')
    p.add_argument('--lstm_layers', type=int, default=flavour.lstm_layers)
    p.add_argument('--output_size', type=int, default=384)
    p.add_argument('--loss', type=str, default=flavour.loss, choices=["featdist", "cosine", "kd", "barlow"])
    p.add_argument('--optimizer', type=str, default=flavour.optimizer, choices=["rmsprop", "adamw", "adam", "lars"])
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


class _LoaderEmbedder:
    """What the reference hands to ``dataset.transformEEGDataLSTMByList`` is its model; items of a loader over the dataset
    are ``eeg[T, C]`` as ``__getitem__`` returns them, so here the callable also applies the step's preprocessing (the
    fused band-pass + z-score takes the stored channel-first layout)."""

    def __init__(self, trainer):
        self.trainer = trainer

    def __call__(self, eeg_btc):
        return self.trainer.model(self.trainer.embed(eeg_btc.transpose(1, 2).contiguous()))


def init_distributed():
    """One process per GPU; RCCL when launched by torchrun, single process otherwise
    (utils/utils.py:467-503 exits without a GPU -- so does this)."""
    if not torch.cuda.is_available():
        print('Does not support training without GPU.')
        sys.exit(1)
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) > 1:
        local = int(os.environ.get("LOCAL_RANK", 0))
        if os.environ.get("CSN_SINGLE_DEVICE"):      # rehearsal of the multi-rank path on a one-GPU box (as in bench.py)
            local = 0
        torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("CSN_DIST_BACKEND", "nccl")     # "nccl" is RCCL on ROCm; gloo only for rehearsals
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend=backend)
        return dist.get_rank(), dist.get_world_size(), local
    torch.cuda.set_device(0)
    return 0, 1, 0


def main(argv=None, flavour=PERILS):
    from cerebralsignalnetworks_amd import Model, EEGFilters
    from cerebralsignalnetworks_amd.dataset import EEGDataset
    from cerebralsignalnetworks_amd.retrieval import evaluate_distributed, evaluate_full
    from torch.utils.data import DataLoader, Subset
    from cerebralsignalnetworks_amd.trainer import DistillTrainer, shard_indices, split_indices
    from cerebralsignalnetworks_amd.losses import HyperParams

    FLAGS, _unparsed = build_parser(flavour).parse_known_args(argv)
    rank, world, local = init_distributed()
    is_main = rank == 0
    if is_main:
        print(FLAGS)
    os.makedirs(FLAGS.log_dir, exist_ok=True)
    hyper = ast.literal_eval(FLAGS.hyperprams)
    kd = _KdParams()
    kd.alpha = hyper.get("alpha", kd.alpha)                 # (Parameters.alpha / .temperature of the Spampinato script, :288-296)
    kd.temperature = hyper.get("temperature", kd.temperature)
    torch.manual_seed(FLAGS.seed)
    device = torch.device("cuda", local)

    if FLAGS.synthetic:
        dataset = EEGDataset(synthetic=FLAGS.synthetic, synthetic_samples=FLAGS.synthetic_samples, time_low=0,
                             time_high=FLAGS.synthetic_samples, seed=FLAGS.seed, device=device,
                             feature_dim=FLAGS.output_size, compat_label_bug=FLAGS.compat_label_bug)
    else:
        spamp = flavour.dataset_flavour == "spampinato"       # utils/EEGDataset.py: subject + split file (:310-317)
        dataset = EEGDataset(eeg_signals_path=FLAGS.eeg_dataset, imagesRoot=FLAGS.images_root,
                             eeg_splits_path=FLAGS.eeg_dataset_split if spamp and os.path.exists(FLAGS.eeg_dataset_split) else None,
                             subject=FLAGS.gallery_subject if spamp else 1, flavour=flavour.dataset_flavour,
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
    if flavour.name == "spampinato":
        if os.path.exists(FLAGS.custom_model_weights):           # :369-371: resume only if the file is there, strict keys
            model.load_state_dict(torch.load(FLAGS.custom_model_weights, map_location="cpu", weights_only=True))
            if is_main:
                print(f"loaded  {FLAGS.custom_model_weights}")
    elif FLAGS.custom_model_weights:
        sd = torch.load(FLAGS.custom_model_weights, map_location="cpu", weights_only=True)
        model.load_state_dict(sd, strict=False)
    sos = EEGFilters(FLAGS.fs, order=FLAGS.filter_order).sos if FLAGS.filter_order else None
    trainer = DistillTrainer(model, sos, loss=FLAGS.loss, lr=FLAGS.learning_rate, optimizer=FLAGS.optimizer,
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
            if FLAGS.compat_label_bug:
                # the reference's own retrieval leg, call for call (:383-389): both loaders go through
                # dataset.transformEEGDataLSTMByList, whose labels are looked up by the position INSIDE the batch
                # (PerilsEEGDataset.py:336-338) -- Recall / Precision come out as the reference prints them, not as
                # retrieval quality.  Every rank walks the whole loaders, like the reference's ranks.
                embedder = _LoaderEmbedder(trainer)
                loaders = [DataLoader(Subset(dataset, ix.tolist()), batch_size=FLAGS.batch_size, shuffle=False)
                           for ix in (train_idx, val_idx)]
                gallery_features, gallery_labels = dataset.transformEEGDataLSTMByList(model=embedder, data_loader=loaders[0])
                query_features, query_labels = dataset.transformEEGDataLSTMByList(model=embedder, data_loader=loaders[1])
                r = evaluate_full(FLAGS, gallery_features, query_features, gallery_labels, query_labels, dataset)
            else:
                # each rank embeds ITS shard of the gallery and of the queries; the neighbour lists are gathered
                g_ix = train_idx[shard_indices(len(train_idx), 0, FLAGS.seed, rank, world, shuffle=False).to(device)]
                q_ix = val_idx[shard_indices(len(val_idx), 0, FLAGS.seed, rank, world, shuffle=False).to(device)]
                if world > 1:       # (the sampler pads a shard by wrapping around: drop the duplicates again)
                    g_ix = g_ix[: len(range(rank, len(train_idx), world))]
                    q_ix = q_ix[: len(range(rank, len(val_idx), world))]
                gallery = trainer.embed_all(dataset.eeg_all[g_ix], FLAGS.batch_size)
                query = trainer.embed_all(dataset.eeg_all[q_ix], FLAGS.batch_size)
                r = evaluate_distributed(FLAGS, list(gallery.cpu().numpy()), list(query.cpu().numpy()), labels_of(g_ix),
                                         labels_of(q_ix), dataset)
            # validation loss over ALL ranks' shards: sum of the batch losses and their count, one all-reduce, so that
            # the best-checkpoint decision below is the same on every rank and covers the whole validation split
            vsum = torch.zeros(2, device=device, dtype=torch.float64)
            with torch.no_grad():
                for b in batches(val_idx, 0, False):
                    out = model(trainer.embed(dataset.eeg_all[b]))
                    vsum[0] += trainer.compute_loss(out, dataset.features_all[b], dataset.labels_dev[b], EPOCH).double()
                    vsum[1] += 1
            if world > 1:
                if dist.get_backend() == "nccl":
                    dist.all_reduce(vsum)
                else:                       # gloo rehearsal: host tensors
                    host = vsum.cpu()
                    dist.all_reduce(host)
                    vsum = host
            val_epoch_loss = float((vsum[0] / vsum[1]).item())
            improved = best_val_loss is None or val_epoch_loss < best_val_loss
            if improved:
                first = best_val_loss is None
                best_val_loss, best_val_loss_epoch = val_epoch_loss, EPOCH
                if is_main:
                    torch.save(model.state_dict(), flavour.checkpoint_path(FLAGS.log_dir, EPOCH, FLAGS.num_epochs, first))
            if is_main:
                print(f"Overall Recall :{r['Recall_Total']} Overall Precision: {r['Precision_Total']} top1: {r['top1']:.4f}")
                print(f"EPOCH {EPOCH} train_loss: {round(epoch_loss, 6)} val_loss: {round(val_epoch_loss, 6)} "
                      f"T: {HyperParams.T} best val loss: {best_val_loss} on epoch: {best_val_loss_epoch}")
        elif is_main:
            print(f"EPOCH {EPOCH} train_loss: {round(epoch_loss, 6)} T: {HyperParams.T}")
        history.append(epoch_loss)
    if is_main and best_val_loss is None and flavour.name == "perils":
        torch.save(model.state_dict(), f"{FLAGS.log_dir}/lstm_dinov2_best_loss.pth")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return history


if __name__ == "__main__":
    main()
