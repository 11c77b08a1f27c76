"""Retrieval evaluation: exact L2 top-k on the GPU + the reference's Recall/Precision bookkeeping.

``evaluate`` keeps the signature of /root/reference/utils/Utilities.py:28
(``evaluate(FLAGS, gallery_features, query_features, gallery_labels, query_labels, dataset)``
-> ``(Recall_Total, Precision_Total)``); the ``faiss.IndexFlatL2`` add/search of :45-55 is the
HIP kernel ``csn_l2_topk``.  ``evaluate_full`` additionally returns top-1 accuracy and (D, I).
"""
import numpy as np
import torch

from . import cabi


def l2_search(gallery_features, query_features, k, device=None):
    """-> (D[nq,k] float32 squared distances, I[nq,k] int64) as numpy arrays."""
    device = device or torch.device("cuda", torch.cuda.current_device())
    g = torch.as_tensor(np.asarray(gallery_features, dtype=np.float32)).reshape(len(gallery_features), -1).to(device)
    q = torch.as_tensor(np.asarray(query_features, dtype=np.float32)).reshape(len(query_features), -1).to(device)
    D, I = cabi.l2_topk(g, q, k)
    return D.cpu().numpy(), I.cpu().numpy()


def _bookkeeping(I, gallery_labels, query_labels, class_id_to_str, class_str_to_id, topK):
    class_scores = {}
    top1 = 0
    for query_idx, search_res in enumerate(I):
        test_label = query_labels[query_idx]
        test_strlabel = class_id_to_str[test_label["ClassId"]]
        name = test_label["ClassName"]
        if name not in class_scores:
            class_scores[name] = {"TP": 0, "classIntanceRetrival": 0, "TotalRetrival": 0, "TotalClass": 0,
                                  "GroundTruths": [], "Predicted": [], "Recall": "", "Precision": ""}
        labels_str = [class_id_to_str[gallery_labels[int(g)]["ClassId"]] for g in search_res]
        count = sum(1 for s in labels_str if s == test_strlabel)
        rec = class_scores[name]
        if name in labels_str:
            rec["TP"] += 1
            rec["classIntanceRetrival"] += count
            rec["Predicted"].append(test_label["ClassId"])
        else:
            rec["Predicted"].append(class_str_to_id[labels_str[0]])
        rec["TotalRetrival"] += topK
        rec["TotalClass"] += 1
        rec["GroundTruths"].append(test_label["ClassId"])
        rec["Recall"] = round((rec["TP"] * 100) / rec["TotalClass"], 2)
        rec["Precision"] = round((rec["classIntanceRetrival"] * 100) / rec["TotalRetrival"], 2)
        top1 += int(gallery_labels[int(search_res[0])]["ClassId"] == test_label["ClassId"])
    recall = float(np.array([v["Recall"] for v in class_scores.values()]).mean())
    precision = float(np.array([v["Precision"] for v in class_scores.values()]).mean())
    return recall, precision, class_scores, top1 / max(1, len(I))


def evaluate_full(FLAGS, gallery_features, query_features, gallery_labels, query_labels, dataset):
    topK = FLAGS.topK
    D, I = l2_search(gallery_features, query_features, topK)
    recall, precision, scores, top1 = _bookkeeping(I, gallery_labels, query_labels, dataset.class_id_to_str,
                                                   dataset.class_str_to_id, topK)
    return dict(Recall_Total=recall, Precision_Total=precision, class_scores=scores, top1=top1, D=D, I=I)


def evaluate(FLAGS, gallery_features, query_features, gallery_labels, query_labels, dataset):
    r = evaluate_full(FLAGS, gallery_features, query_features, gallery_labels, query_labels, dataset)
    print(f"Overall Recall :{r['Recall_Total']} Overall Precision: {r['Precision_Total']}")
    return r["Recall_Total"], r["Precision_Total"]


def _all_gather_rows(local, group=None):
    """Concatenate row blocks of different lengths from all ranks, in rank order (host tensors: the blocks are
    embeddings / label ids computed once per evaluation, a few MB)."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    # RCCL moves device tensors only; gloo (CPU tests) host tensors
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    local = torch.as_tensor(np.asarray(local)).contiguous().to(dev)
    n = torch.tensor([local.shape[0]], dtype=torch.long, device=dev)
    sizes = [torch.zeros(1, dtype=torch.long, device=dev) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(v.item()) for v in sizes]
    pad = torch.zeros((max(sizes),) + tuple(local.shape[1:]), dtype=local.dtype, device=dev)
    pad[:local.shape[0]] = local
    parts = [torch.zeros_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[:k] for p, k in zip(parts, sizes)]).cpu().numpy(), sizes


def evaluate_distributed(FLAGS, gallery_features, query_features, gallery_labels, query_labels, dataset,
                         search_fn=None, group=None):
    """Multi-rank evaluation: every rank holds the embeddings of ITS shard of the gallery
    and of the queries (the pattern of PerilsEEGDataset.py:191-215, where shards are gathered to rank 0).  The
    gallery is all-gathered and kept replicated, every rank searches its own queries against it (csn_l2_topk),
    and the per-query neighbour lists are gathered so that every rank reports the same Recall / Precision / top-1
    as a single-process ``evaluate_full`` over the concatenated data.  Labels are the reference's label dicts.
    ``search_fn(gallery, query, k) -> (D, I)`` defaults to the HIP search."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return evaluate_full(FLAGS, gallery_features, query_features, gallery_labels, query_labels, dataset)
    topK = FLAGS.topK
    search = search_fn or l2_search
    by_id = {v["ClassId"]: v for v in list(gallery_labels) + list(query_labels)}
    gal, _ = _all_gather_rows(np.asarray(gallery_features, dtype=np.float32).reshape(len(gallery_features), -1), group)
    gal_ids, _ = _all_gather_rows(np.array([l["ClassId"] for l in gallery_labels], dtype=np.int64), group)
    qry = np.asarray(query_features, dtype=np.float32).reshape(len(query_features), -1)
    D_loc, I_loc = search(gal, qry, topK) if len(qry) else (np.zeros((0, topK), np.float32), np.zeros((0, topK), np.int64))
    I_all, _ = _all_gather_rows(np.asarray(I_loc, dtype=np.int64), group)
    D_all, _ = _all_gather_rows(np.asarray(D_loc, dtype=np.float32), group)
    q_ids, _ = _all_gather_rows(np.array([l["ClassId"] for l in query_labels], dtype=np.int64), group)
    # label dicts by class id: every rank needs the dict of every class that occurs anywhere
    ids_known = sorted(by_id)
    names = [None] * dist.get_world_size(group)
    dist.all_gather_object(names, {k: by_id[k] for k in ids_known}, group=group)
    for d in names:
        by_id.update(d)
    recall, precision, scores, top1 = _bookkeeping(I_all, [by_id[int(k)] for k in gal_ids], [by_id[int(k)] for k in q_ids],
                                                   dataset.class_id_to_str, dataset.class_str_to_id, topK)
    return dict(Recall_Total=recall, Precision_Total=precision, class_scores=scores, top1=top1, D=D_all, I=I_all)

