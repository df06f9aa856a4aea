"""Segment helpers on the hot path -- /root/reference/src/segment_utils.py:536-545 (to_one_hot) plus the
seg-IoU used to check "seg-IoU within 1e-3 of the reference" (host-side, tiny;
Hungarian matching via scipy instead of lapsolver). The remaining metrics of that file are out of scope."""
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
