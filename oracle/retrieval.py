"""Oracle: exact L2 top-k retrieval + the reference's Recall/Precision bookkeeping
(test infrastructure only).

Restates /root/reference/utils/Utilities.py:28-169 (``evaluate``).  The search
itself is ``faiss.IndexFlatL2(d).add/search`` there -- faiss is a third-party
dependency that is neither in /root/reference nor installed here (unpinned: the
reference has no requirements file), so its *published* semantics are restated:
squared-L2 distances, k smallest per query in ascending order.  Ties are broken
by the lower gallery index.  parity unpinned for the search itself (no fixture of
the reference covers it); the bookkeeping follows the reference text line by line.
"""
import numpy as np


def l2_topk(gallery, query, k):
    """Returns (D[nq,k] squared distances float64, I[nq,k] int64), ascending, ties -> lower index."""
    g = np.asarray(gallery, np.float64)
    q = np.asarray(query, np.float64)
    nq = q.shape[0]
    D = np.empty((nq, k), np.float64)
    I = np.empty((nq, k), np.int64)
    for i in range(nq):
        d = ((g - q[i][None, :]) ** 2).sum(axis=1)
        order = np.lexsort((np.arange(d.shape[0]), d))[:k]
        I[i] = order
        D[i] = d[order]
    return D, I


def evaluate_from_indices(I, gallery_labels, query_labels, class_id_to_str, topK):
    """Utilities.py:63-164 given the search result ``I``.

    labels are dicts with "ClassId"/"ClassName" (PerilsEEGDataset.py:84).
    Returns (Recall_Total, Precision_Total, per_class dict, top1_accuracy).
    """
    scores = {}
    top1_hits = 0
    for qi, res in enumerate(I):
        test_label = query_labels[qi]
        test_str = class_id_to_str[test_label["ClassId"]]
        name = test_label["ClassName"]
        if name not in scores:
            scores[name] = dict(TP=0, classIntanceRetrival=0, TotalRetrival=0, TotalClass=0)
        strs = [class_id_to_str[gallery_labels[int(g)]["ClassId"]] for g in res]
        count = sum(1 for s in strs if s == test_str)
        if name in strs:
            scores[name]["TP"] += 1
            scores[name]["classIntanceRetrival"] += count
        scores[name]["TotalRetrival"] += topK
        scores[name]["TotalClass"] += 1
        if gallery_labels[int(res[0])]["ClassId"] == test_label["ClassId"]:
            top1_hits += 1
    for v in scores.values():
        v["Recall"] = round((v["TP"] * 100) / v["TotalClass"], 2)
        v["Precision"] = round((v["classIntanceRetrival"] * 100) / v["TotalRetrival"], 2)
    recall = float(np.array([v["Recall"] for v in scores.values()]).mean())
    precision = float(np.array([v["Precision"] for v in scores.values()]).mean())
    top1 = top1_hits / max(1, len(I))
    return recall, precision, scores, top1


def evaluate(gallery_features, query_features, gallery_labels, query_labels, class_id_to_str, topK=5):
    g = np.asarray(gallery_features, np.float32).reshape(len(gallery_features), -1)
    q = np.asarray(query_features, np.float32).reshape(len(query_features), -1)
    D, I = l2_topk(g, q, topK)
    return evaluate_from_indices(I, gallery_labels, query_labels, class_id_to_str, topK) + (D, I)
