#!/usr/bin/env python3
"""Retrieval evaluation with the surface of the reference's LstmDistillFromDinoV2Eval.py
(flags :157-235; checkpoint loading :309-313; outputs :492-522): embeds the train / test split
with the LSTM, exact L2 top-K on the GPU (csn_l2_topk instead of faiss.IndexFlatL2), the
reference's per-class Recall / Precision, and additionally top-1 accuracy (the reference never
prints it; derived from I[:,0])."""
import csv
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from LstmDistillFromDinoV2Train import build_parser  # noqa: E402


def load_checkpoint_into(model, path):
    """Plain state_dict (Train.py:414) or a DINO-style dict {"teacher": ...} with "backbone." keys (Eval.py:309-313)."""
    sd = torch.load(path, map_location="cpu", weights_only=True)
    if isinstance(sd, dict) and "teacher" in sd:
        sd = sd["teacher"]
    sd = {k.replace("module.", "").replace("backbone.", ""): v for k, v in sd.items()}
    return model.load_state_dict(sd, strict=False)


def main(argv=None):
    from cerebralsignalnetworks_amd import Model, EEGFilters
    from cerebralsignalnetworks_amd.dataset import EEGDataset
    from cerebralsignalnetworks_amd.retrieval import evaluate_distributed, evaluate_full
    from cerebralsignalnetworks_amd.trainer import DistillTrainer, split_indices

    p = build_parser()
    p.add_argument('--dino_base_model_weights', type=str, default="")
    FLAGS, _ = p.parse_known_args(argv)
    if not torch.cuda.is_available():
        print('Does not support evaluation without GPU.')
        sys.exit(1)
    # one process per GPU when launched with torchrun: the gallery and the queries are sharded over the ranks
    import LstmDistillFromDinoV2Train as train_cli
    rank, world, local = train_cli.init_distributed()
    device = torch.device("cuda", local)
    os.makedirs(FLAGS.log_dir, exist_ok=True)
    t0 = time.perf_counter()
    if FLAGS.synthetic:
        dataset = EEGDataset(synthetic=FLAGS.synthetic, time_low=0, time_high=500, seed=FLAGS.seed, device=device,
                             feature_dim=FLAGS.output_size, compat_label_bug=FLAGS.compat_label_bug)
    else:
        dataset = EEGDataset(eeg_signals_path=FLAGS.eeg_dataset, eeg_splits_path=None, imagesRoot=FLAGS.images_root,
                             time_low=FLAGS.time_low, time_high=FLAGS.time_high, device=device,
                             compat_label_bug=FLAGS.compat_label_bug)
    C = dataset.eeg_all.shape[1]
    dtype = torch.bfloat16 if FLAGS.dtype == "bf16" else torch.float32
    model = Model(input_size=C, lstm_size=FLAGS.hidden_size, lstm_layers=FLAGS.lstm_layers,
                  output_size=FLAGS.output_size, include_top=False, compute_dtype=dtype).to(device)
    if FLAGS.custom_model_weights:
        print(load_checkpoint_into(model, FLAGS.custom_model_weights))
    sos = EEGFilters(FLAGS.fs, order=FLAGS.filter_order).sos if FLAGS.filter_order else None
    trainer = DistillTrainer(model, sos, loss="cosine")
    N = len(dataset)
    if FLAGS.compat_label_bug:
        # the reference's own calls (LstmDistillFromDinoV2Eval.py:324-334): loaders over the random_split halves, both
        # through dataset.transformEEGDataLSTMByList with its batch-local label lookup; every rank walks everything
        from torch.utils.data import DataLoader, Subset
        embedder = train_cli._LoaderEmbedder(trainer)
        tr, te = split_indices(N, (0.8, 0.2), seed=43)
        model.eval()
        loaders = [DataLoader(Subset(dataset, ix.tolist()), batch_size=FLAGS.batch_size, shuffle=False) for ix in (tr, te)]
        gallery, glab = dataset.transformEEGDataLSTMByList(model=embedder, data_loader=loaders[0])
        query, qlab = dataset.transformEEGDataLSTMByList(model=embedder, data_loader=loaders[1])
        trainer.check_device_status()
        r = evaluate_full(FLAGS, gallery, query, glab, qlab, dataset)
    else:
        tr, te = (ix[rank::world].to(device) for ix in split_indices(N, (0.8, 0.2), seed=43))      # random_split, Eval.py:324-325
        gallery = trainer.embed_all(dataset.eeg_all[tr], FLAGS.batch_size).cpu().numpy()
        query = trainer.embed_all(dataset.eeg_all[te], FLAGS.batch_size).cpu().numpy()
        trainer.check_device_status()      # every embedding batch above: a timed-out in-kernel hand-off invalidates the run
        glab = [dataset.getLabelbyIndex(int(i)) for i in tr.cpu()]
        qlab = [dataset.getLabelbyIndex(int(i)) for i in te.cpu()]
        r = evaluate_distributed(FLAGS, list(gallery), list(query), glab, qlab, dataset)
    dt = time.perf_counter() - t0
    if rank != 0:
        return r
    print(f"Overall Recall :{r['Recall_Total']} Overall Precision: {r['Precision_Total']} top1: {r['top1']:.4f}")
    base = f"{FLAGS.log_dir}/Theperils_sub_{FLAGS.query_subject}_Scores"
    out = {"data": r["class_scores"], "metadata": {"processing_time": f"{dt:.2f}s", "flags": vars(FLAGS),
                                                   "Recall_Total": r["Recall_Total"],
                                                   "Precision_Total": r["Precision_Total"], "top1": r["top1"]}}
    torch.save(out, base + ".pth")
    with open(base + ".txt", "w") as f:
        f.write(json.dumps(out, default=lambda o: o.tolist() if isinstance(o, np.ndarray) else str(o)))
    with open(base + "_.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["class", "Recall", "Precision", "TP", "TotalClass"])
        for k, v in r["class_scores"].items():
            w.writerow([k, v["Recall"], v["Precision"], v["TP"], v["TotalClass"]])
    return r


if __name__ == "__main__":
    main()
