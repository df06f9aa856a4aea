"""Segment helpers on the hot path -- /root/reference/src/segment_utils.py:536-545 (to_one_hot) plus the
seg-IoU used to check "seg-IoU within 1e-3 of the reference"  and the evaluation the script logs.
The per-cloud functions keep the reference's signatures (Hungarian matching on the host via scipy instead of lapsolver);
SIOU_matched_segments_usecd_batch is the same evaluation for a whole batch on the device (sednet_hip.ops.segment_metrics:
one wave per cloud solves the assignment). The remaining metrics of that file are out of scope."""
import numpy as np
import torch
from scipy.optimize import linear_sum_assignment


def to_one_hot(target, maxx=50, device_id=0):
    """segment_utils.py:536-545."""
    if isinstance(target, np.ndarray):
        target = torch.from_numpy(target.astype(np.int64))
        if torch.cuda.is_available():
            target = target.cuda(device_id)
    N = target.shape[0]
    target_one_hot = torch.zeros((N, maxx), device=target.device)
    return target_one_hot.scatter_(1, target.unsqueeze(1).long(), 1)


def seg_iou(pred_labels, gt_labels):
    """Hungarian-matched mean segment IoU (restates the matching step of segment_utils.py:194-242 /
    relaxed_iou_fast :609-627 for hard labels): mean IoU over matched (pred, gt) pairs whose gt is non-empty."""
    pred = np.asarray(pred_labels).astype(np.int64)
    gt = np.asarray(gt_labels).astype(np.int64)
    P, G = pred.max() + 1, gt.max() + 1
    inter = np.zeros((P, G))
    np.add.at(inter, (pred, gt), 1)
    union = inter.sum(1, keepdims=True) + inter.sum(0, keepdims=True) - inter
    iou = inter / np.maximum(union, 1)
    r, c = linear_sum_assignment(1.0 - iou)
    keep = inter.sum(0)[c] > 0
    return float(iou[r, c][keep].mean())


def relaxed_iou_fast(pred, gt, max_clusters=50):
    """segment_utils.py:609-627: pred/gt one-hot [B,N,K] -> relaxed IoU cost [B,K,K]."""
    norms_p = torch.unsqueeze(torch.sum(pred, 1), 2)
    norms_g = torch.unsqueeze(torch.sum(gt, 1), 1)
    dots = pred.transpose(1, 2) @ gt
    return dots / (norms_p + norms_g - dots + 1e-7)


def matching_iou(matching, predicted_labels, labels):
    """segment_utils.py:548-577."""
    IOU = []
    for b in range(labels.shape[0]):
        iou_b = []
        rows, cols = matching[b]
        for r, c in zip(rows, cols):
            pi, gi = predicted_labels[b] == r, labels[b] == c
            if gi.sum() == 0 and pi.sum() == 0:
                continue
            iou_b.append(np.logical_and(pi, gi).sum() / (np.logical_or(pi, gi).sum() + 1e-8))
        IOU.append(np.mean(iou_b))
    return np.mean(IOU)


def match(target, pred_labels):
    """fitting_utils.py:362-376: Hungarian matching of predicted to ground-truth segments on the relaxed IoU
    (scipy.optimize.linear_sum_assignment in place of lapsolver.solve_dense; host side, <= 50 x 50)."""
    lo = to_one_hot(np.asarray(target)).cpu()
    co = to_one_hot(np.asarray(pred_labels)).cpu()
    cost = relaxed_iou_fast(co.unsqueeze(0).float(), lo.unsqueeze(0).float())
    rids, cids = linear_sum_assignment(1.0 - cost[0].numpy())
    return rids, cids, np.unique(target), np.unique(pred_labels)


def SIOU_matched_segments(target, pred_labels, primitives_pred, primitives, weights=None):
    """segment_utils.py:140-185 + mean_IOU_primitive_segment :359-421 -> (segment IoU, primitive-type IoU over matched
    segments, matching, [gt, pred] type pairs, recall). Type ids are folded like the reference ({0,6,7} -> 9, 8 -> 2);
    ground-truth segments with fewer than 100 points are skipped (:392); the ground-truth type of a segment is its
    first point's (:406), the predicted type of a predicted segment argmax_L sum_n onehot(pred)[n,L] weights[n,k]
    (:509-517; weights [N,K], default = one-hot of pred_labels)."""
    fold = lambda a: np.where(np.isin(a, (0, 6, 7)), 9, np.where(a == 8, 2, a))
    target, pred_labels = np.asarray(target), np.asarray(pred_labels)
    primitives, primitives_pred = fold(np.asarray(primitives)), fold(np.asarray(primitives_pred))
    rids, cids, _, _ = match(target, pred_labels)
    if weights is None:
        weights = to_one_hot(pred_labels, int(pred_labels.max()) + 1)
    weights = torch.as_tensor(weights).float().cpu()
    prim_pred = primitive_type_segment_torch(to_one_hot(primitives_pred, 10).cpu(), weights).numpy()
    iou_b, recall_b, prim_b, pairs = [], [], [], []
    for r, c in zip(rids, cids):
        pi, gi = pred_labels == r, target == c
        if gi.sum() == 0 or pi.sum() == 0 or gi.sum() < 100:
            continue
        tp = np.sum(pi & gi)
        iou_b.append(tp / (np.sum(pi | gi) + 1e-8))
        recall_b.append(tp / (tp + np.sum(~pi & gi) + 1e-8))
        gt_t, pred_t = primitives[gi][0], prim_pred[r]
        prim_b.append(gt_t == pred_t)
        pairs.append([gt_t, pred_t])
    return np.mean(iou_b), np.mean(prim_b), [[rids, cids]], pairs, np.mean(recall_b)


def primitive_type_segment_torch(pred, weights):
    """Type of every predicted segment = argmax_L sum_n pred[n,L] weights[n,k] (segment_utils.py:509-517)."""
    return torch.max(pred.t().float() @ weights.float(), 0)[1]


def mean_IOU_primitive_segment_usecd(matching, predicted_labels, labels, pred_prim, gt_prim, points):
    """segment_utils.py:424-494: IoU and type agreement over matched segments, plus the chamfer recall -- a matched pair
    counts as recalled when the chamfer distance between its two point sets is below 0.2 (cd / 2 < 0.1). The per-pair
    chamfer runs on the HIP kernels (src/utils.py)."""
    from src.utils import chamfer_distance
    IOU, RECALL, IOU_prim = [], [], []
    pairs = []
    for b in range(labels.shape[0]):
        iou_b, prim_b, pairs = [], [], []
        n_gt = np.unique(labels[b]).shape[0]
        rows, cols = matching[b]
        recalled = 0
        for r, c in zip(rows, cols):
            pi, gi = predicted_labels[b] == r, labels[b] == c
            if gi.sum() == 0 or pi.sum() == 0:
                continue
            iou_b.append(np.sum(pi & gi) / (np.sum(pi | gi) + 1e-8))
            sel_p = torch.as_tensor(np.where(pi)[0], device=points.device)
            sel_g = torch.as_tensor(np.where(gi)[0], device=points.device)
            if float(chamfer_distance(points[sel_p][None], points[sel_g][None])) / 2 < 0.1:
                recalled += 1
            gt_type, pred_type = gt_prim[b][gi][0], pred_prim[b][r]
            prim_b.append(gt_type == pred_type)
            pairs.append([gt_type, pred_type])
        IOU.append(np.mean(iou_b))
        RECALL.append(float(recalled) / n_gt)
        IOU_prim.append(np.mean(prim_b))
    return np.mean(IOU), np.mean(IOU_prim), pairs, np.mean(RECALL)


def SIOU_matched_segments_usecd(target, pred_labels, primitives_pred, primitives, weights, points):
    """segment_utils.py:194-242 (what generate_predictions_aug.py:389 logs): Hungarian matching of predicted and true
    segments on the relaxed IoU, then (segment IoU, type IoU, matching, [gt, pred] type pairs, chamfer recall).
    target / pred_labels / primitives* are numpy [N] (type ids folded in place like the reference: {0,6,7} -> 9,
    8 -> 2), weights [N,K] one-hot / soft membership, points [N,3] on the device."""
    for a in (primitives, primitives_pred):
        a[(a == 0) | (a == 6) | (a == 7)] = 9
        a[a == 8] = 2
    dev = points.device
    cost = relaxed_iou_fast(to_one_hot(pred_labels).to(dev).unsqueeze(0).float(),
                            to_one_hot(target).to(dev).unsqueeze(0).float())
    rids, cids = linear_sum_assignment(1.0 - cost[0].cpu().numpy())
    matching = [[rids, cids]]
    prim_hot = to_one_hot(primitives_pred, 10).to(dev).float()
    prim_pred = primitive_type_segment_torch(prim_hot, weights.to(dev)).cpu().numpy()
    s_iou, p_iou, pairs, recall = mean_IOU_primitive_segment_usecd(
        matching, np.expand_dims(pred_labels, 0), np.expand_dims(target, 0), np.expand_dims(prim_pred, 0),
        np.expand_dims(primitives, 0), points)
    return s_iou, p_iou, matching, pairs, recall


def SIOU_matched_segments_usecd_batch(target, pred_labels, primitives_pred, primitives, points, K=50):
    """segment_utils.py:194-242 for a batch, hard labels, on the device: target / pred_labels / primitives_pred / primitives
    [B,N] integer device tensors (type ids unfolded: the kernels fold {0,6,7} -> 9, 8 -> 2 themselves and leave the inputs alone),
    points [B,N,3]. -> (segment IoU [B], type IoU [B], matching [B,K] (column of every row), type pairs [B,K,2], chamfer recall [B])
    as device tensors; what generate_predictions_aug.py:441 logs is the mean of each over the clouds."""
    from sednet_hip import ops
    m, col, pairs = ops.segment_metrics(pred_labels, target, primitives_pred, primitives, points, K)
    return m[:, 0], m[:, 1], col, pairs, m[:, 2]
