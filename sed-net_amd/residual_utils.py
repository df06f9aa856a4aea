"""Evaluation -- the reference caller of the fit path, /root/reference/Fitting_patches_and_edges/residual_utils.py:49-378,
on the MI355X kernels: embedding -> guarded mean-shift -> Hungarian match -> per-segment type
vote -> batched LSQ fits -> closed-form residuals -> separate_losses. Same method names, arguments and return structure;
eval mode is the supported (inference) path, train mode computes the same forward values without autograd. Spline
segments are dropped like segments that are too small (parameters[id] = None): SplineNet is out of scope."""
import numpy as np
import torch

from src.fitting_optimization import FittingModule
from src.fitting_utils import match, to_one_hot, weights_normalize
from src.mean_shift import MeanShift
from src.primitive_forward import fit_one_shape_torch
from src.primitives import ResidualLoss
from src.segment_utils import SIOU_matched_segments


def _mode(a):
    """scipy.stats.mode(...)[0] for small non-negative ints: most frequent, ties -> smallest (residual_utils.py:188, :259)."""
    return int(np.bincount(np.asarray(a).astype(np.int64)).argmax())


class Evaluation:
    def __init__(self, userspace=None, closed_path=None, open_path=None):
        self.res_loss = ResidualLoss()
        self.fitter = FittingModule(closed_path, open_path)
        self.ms = MeanShift()

    def guard_mean_shift(self, embedding, quantile, iterations, kernel_type="gaussian"):
        """residual_utils.py:69-84."""
        while True:
            _, center, bandwidth, cluster_ids = self.ms.mean_shift(embedding, 10000, quantile, iterations,
                                                                   kernel_type=kernel_type)
            if torch.unique(cluster_ids).shape[0] > 49:
                quantile *= 1.2
            else:
                break
        return center, bandwidth, cluster_ids

    def fitting_loss(self, embedding, points, normals, labels, primitives, primitives_log_prob, quantile=0.125,
                     iterations=5, lamb=1.0, debug=False, eval=False):
        """residual_utils.py:86-152. embedding [B,N,d], points/normals [B,N,3], labels/primitives numpy [B,N],
        primitives_log_prob [B,C,N] -> (loss list + [s_iou, p_iou], [parameters, cluster_ids, weights])."""
        embedding = torch.nn.functional.normalize(embedding, p=2, dim=2)
        prim_pred = torch.max(primitives_log_prob, 1)[1].cpu().numpy()
        for b in range(embedding.shape[0]):
            center, bandwidth, cluster_ids = self.guard_mean_shift(embedding[b], quantile, iterations)
            weights = center @ torch.transpose(embedding[b], 1, 0)
            if not eval:
                loss, parameters, _, rows, cols, distance = self.residual_train_mode(
                    points[b], normals[b], labels[b], cluster_ids, primitives[b], weights, bandwidth, lamb=lamb)
            else:
                loss, parameters, _ = self.residual_eval_mode(
                    points[b], normals[b], labels[b], cluster_ids, primitives[b], prim_pred[b], weights, bandwidth,
                    lamb=lamb)
                ids = cluster_ids.cpu().numpy()
                weights = to_one_hot(ids, np.unique(ids).shape[0]).T
            ids_np = cluster_ids.cpu().numpy()
            hard = to_one_hot(ids_np, np.unique(ids_np).shape[0]) if eval else weights.T         # :145-150 (weights.T)
            s_iou, p_iou = SIOU_matched_segments(labels[b], ids_np, prim_pred[b], primitives[b], hard)[:2]
            loss = loss + [s_iou, p_iou]
        return loss, [parameters, cluster_ids.cpu().numpy(), weights]

    def _segments(self, points, normals, labels, cluster_ids, type_source, train):
        rows, cols, unique_target, unique_pred = match(labels, cluster_ids)
        data = []
        for index, i in enumerate(unique_pred):
            gt_i = labels == (cols[i] if train else cols[index])        # residual_utils.py:176 vs :247 (sic)
            pred_i = cluster_ids == i
            if gt_i.sum() == 0 or pred_i.sum() == 0:
                continue
            if train:
                data.append([points, normals, _mode(type_source[gt_i]), points[torch.as_tensor(gt_i, device=points.device)],
                             None, (index, i)])
            else:
                sel = torch.as_tensor(pred_i, device=points.device)
                data.append([points[sel], normals[sel], _mode(type_source[pred_i]),
                             points[torch.as_tensor(gt_i, device=points.device)], pred_i, (index, i)])
        return data, rows, cols

    def residual_train_mode(self, points, normals, labels, cluster_ids, primitives, weights, bw, lamb=1.0):
        """residual_utils.py:154-213 (forward values; no autograd through the HIP fits)."""
        if not isinstance(cluster_ids, np.ndarray):
            cluster_ids = cluster_ids.cpu().numpy()
        data, rows, cols = self._segments(points, normals, labels, cluster_ids, primitives, train=True)
        w = torch.transpose(weights_normalize(weights, float(bw)), 1, 0)
        gt_points, _ = fit_one_shape_torch(data, self.fitter, w, bw, eval=False)
        distance = self.res_loss.residual_loss(gt_points, self.fitter.fitting.parameters)
        return self.separate_losses(distance, gt_points, lamb=lamb), self.fitter.fitting.parameters, None, rows, cols, distance

    def residual_eval_mode(self, points, normals, labels, cluster_ids, primitives, pred_primitives, weights, bw, lamb=1.0,
                           sample_points=False, if_optimize=False, if_visualize=False, epsilon=None):
        """residual_utils.py:215-331."""
        if not isinstance(cluster_ids, np.ndarray):
            cluster_ids = cluster_ids.cpu().numpy()
        one_hot = to_one_hot(cluster_ids, np.unique(cluster_ids).shape[0]).T                 # :232-234
        data, _, _ = self._segments(points, normals, labels, cluster_ids, pred_primitives, train=False)
        w = torch.transpose(weights_normalize(one_hot.float(), float(bw)), 1, 0)            # :300-301
        w = to_one_hot(torch.max(w, 1)[1], w.shape[1])                                      # :302-304
        gt_points, _ = fit_one_shape_torch(data, self.fitter, w, bw, eval=True)
        distance = self.res_loss.residual_loss(gt_points, self.fitter.fitting.parameters, sqrt=True)
        return self.separate_losses(distance, gt_points, lamb=lamb), self.fitter.fitting.parameters, None

    def separate_losses(self, distance, gt_points, lamb=1.0):
        """residual_utils.py:333-378."""
        Loss, geometric_loss, spline_loss = [], [], []
        for v in sorted(gt_points.keys()):
            if gt_points[v] is None:
                continue
            if distance[v][1] > 1:
                distance[v][1] = torch.ones(1, device=distance[v][1].device)[0] * 0.1
            if distance[v][0] in ["closed-spline", "open-spline"]:
                spline_loss.append(distance[v][1].item())
                Loss.append(distance[v][1] * lamb)
            else:
                geometric_loss.append(distance[v][1].item())
                Loss.append(distance[v][1])
        Loss = torch.mean(torch.stack(Loss)) if Loss else torch.zeros(1)
        return [Loss, np.mean(geometric_loss) if geometric_loss else None, np.mean(spline_loss) if spline_loss else None]
