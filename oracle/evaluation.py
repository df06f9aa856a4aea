"""Oracle: the reference caller of the fit path in eval mode (numpy + scipy's Hungarian solver).

Test infrastructure only -- see oracle/__init__.py.
Follows /root/reference/Fitting_patches_and_edges/residual_utils.py:86-152 (fitting_loss), :215-331 (residual_eval_mode),
:333-378 (separate_losses), /root/reference/src/fitting_utils.py:362-376 (match), /root/reference/src/segment_utils.py:
609-627 (relaxed_iou_fast), :140-185 (SIOU_matched_segments), :359-421 (mean_IOU_primitive_segment), :509-517
(primitive_type_segment_torch). lapsolver.solve_dense -> scipy.optimize.linear_sum_assignment (same optimum; the
fixtures have a unique one)."""
import numpy as np
from scipy.optimize import linear_sum_assignment

from . import fit as ofit
from . import mean_shift as oms

F32 = np.float32


def one_hot(labels, maxx=50):
    """segment_utils.py:536-545."""
    out = np.zeros((labels.shape[0], maxx), F32)
    out[np.arange(labels.shape[0]), labels.astype(np.int64)] = 1
    return out


def relaxed_iou(pred, gt):
    """segment_utils.py:609-627 for one cloud: pred [N,K], gt [N,K] one-hot -> [K,K]."""
    dots = pred.T @ gt
    return dots / (pred.sum(0)[:, None] + gt.sum(0)[None, :] - dots + F32(1e-7))


def match(target, pred_labels):
    """fitting_utils.py:362-376 -> (rids, cids, unique_target, unique_pred)."""
    cost = 1.0 - relaxed_iou(one_hot(pred_labels), one_hot(target))
    rids, cids = linear_sum_assignment(cost)
    return rids, cids, np.unique(target), np.unique(pred_labels)


def fold_types(a):
    """segment_utils.py:156-164: {0, 6, 7} -> 9 (closed spline), 8 -> 2 (open spline)."""
    a = np.array(a)
    a[(a == 0) | (a == 6) | (a == 7)] = 9
    a[a == 8] = 2
    return a


def siou_matched_segments(target, pred_labels, primitives_pred, primitives, weights):
    """segment_utils.py:140-185 + :359-421 -> (segment IoU, type IoU, matching, [gt, pred] type pairs, recall).
    weights [N,K]: membership of the points in the predicted segments."""
    primitives, primitives_pred = fold_types(primitives), fold_types(primitives_pred)
    rids, cids = linear_sum_assignment(1.0 - relaxed_iou(one_hot(pred_labels), one_hot(target)))
    prim_pred = np.argmax(one_hot(primitives_pred, 10).T @ weights, 0)          # :509-517
    iou_b, recall_b, prim_b, pairs = [], [], [], []
    for r, c in zip(rids, cids):
        pi, gi = pred_labels == r, target == c
        if gi.sum() == 0 or pi.sum() == 0 or gi.sum() < 100:                    # :388-393
            continue
        tp = np.sum(pi & gi)
        iou_b.append(tp / (np.sum(pi | gi) + 1e-8))
        recall_b.append(tp / (tp + np.sum(~pi & gi) + 1e-8))
        gt_t, pred_t = primitives[gi][0], prim_pred[r]
        prim_b.append(gt_t == pred_t)
        pairs.append([gt_t, pred_t])
    return np.mean(iou_b), np.mean(prim_b), [[rids, cids]], pairs, np.mean(recall_b)


def mode(a):
    """scipy.stats.mode(...)[0]: most frequent value, ties -> smallest."""
    return int(np.bincount(np.asarray(a).astype(np.int64)).argmax())


def separate_losses(distance, gt_points, lamb=1.0):
    """residual_utils.py:333-378 -> [Loss, geometric_loss, spline_loss]."""
    loss, geometric, spline = [], [], []
    for v in sorted(gt_points.keys()):
        if gt_points[v] is None:
            continue
        kind, d = distance[v]
        if d > 1:
            d = F32(0.1)
        if kind in ("closed-spline", "open-spline"):
            spline.append(float(d)); loss.append(d * lamb)
        else:
            geometric.append(float(d)); loss.append(d)
    return [np.mean(np.array(loss, F32)) if loss else F32(0), np.mean(geometric) if geometric else None,
            np.mean(spline) if spline else None]


def fitting_loss_eval(embedding, points, normals, labels, primitives, log_prob, quantile, iterations, lamb=1.0):
    """Evaluation.fitting_loss(eval=True) for ONE cloud: embedding [N,d], points/normals [N,3], labels/primitives [N],
    log_prob [C,N] -> ([Loss, geometric, spline, s_iou, p_iou], parameters {pred label id: [...] or None}, cluster_ids,
    per-segment residuals {pred label id: mean sqrt distance of the matched ground-truth points})."""
    X = embedding / np.maximum(np.linalg.norm(embedding, axis=1, keepdims=True), 1e-12)      # F.normalize (:103)
    _, _, cluster_ids, _ = oms.guard_mean_shift(X.astype(F32), quantile, iterations)          # :69-84
    pred_prim = np.argmax(log_prob, 0)                                                       # :111
    rows, cols, _, unique_pred = match(labels, cluster_ids)                                  # :236
    params, gt_points = {}, {}
    for index, i in enumerate(unique_pred):
        gi, pi = labels == cols[index], cluster_ids == i                                     # :247-248 (sic: index)
        if gi.sum() == 0 or pi.sum() == 0:
            continue
        t = mode(pred_prim[pi])                                                              # :259
        seg = ofit.fit_segments_eval(points[pi], normals[pi], np.zeros(int(pi.sum()), np.int64), [t])[0]
        params[i] = seg                                                                      # keyed by label_index = i
        gt_points[i] = points[gi] if seg is not None else None
    distance = {k: [v[0], ofit.residual(gt_points[k], v, sqrt=True)] for k, v in params.items() if v is not None}
    loss = separate_losses(distance, gt_points, lamb)
    n_seg = np.unique(cluster_ids).shape[0]
    s_iou, p_iou, _, _, _ = siou_matched_segments(labels, cluster_ids, pred_prim, primitives,
                                                  one_hot(cluster_ids, n_seg))                # :145-150
    return loss + [s_iou, p_iou], params, cluster_ids, {k: float(v[1]) for k, v in distance.items()}
