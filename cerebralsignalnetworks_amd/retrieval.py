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
