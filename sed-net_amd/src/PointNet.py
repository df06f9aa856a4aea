"""Dynamic-graph operators -- same surface as /root/reference/src/PointNet.py:62-208, executed on
gfx950 by libsedhip.so. Tensors follow the reference conventions (x [B,C,N] channel-major in, int64
indices out); layout changes at this boundary are torch plumbing, the distance/selection work is HIP.
"""
import numpy as np
import torch

from sednet_hip import ops


def _subsample(k1, k2, normal):
    if normal:
        raise NotImplementedError("normal=True neighbour sampling (PointNet.py:66-71) is never used by SED-Net")
    return np.arange(0, k2, k2 // k1)                       # PointNet.py:65


def _point_major(x):
    """[B,C,N] -> padded contiguous [B,N,D]."""
    return ops.pad_features(x.detach().transpose(1, 2))


def knn(x, k1, k2, normal=False):
    """PointNet.py:62-87: indices of the k2 nearest points in feature space (self first), subsampled
    to k1 -> [B,N,k1] int64."""
    sel = _subsample(k1, k2, normal)
    with torch.no_grad():
        idx = ops.knn_features(_point_major(x), k2, C=x.shape[1])
        if sel.shape[0] != k2:
            idx = idx[:, :, torch.as_tensor(sel, device=idx.device)]
    return idx.long()


def knn_points_normals(x, k1, k2, normal_metric_W=1., normal=False):
    """PointNet.py:90-137: kNN under Dp * (1 + W * Dn) on x [B,6,N] (xyz, unit normals)."""
    sel = _subsample(k1, k2, normal)
    with torch.no_grad():
        idx = ops.knn_points_normals(x.detach(), k2, normal_metric_W)
        if sel.shape[0] != k2:
            idx = idx[:, :, torch.as_tensor(sel, device=idx.device)]
    return idx.long()


def _gather_feature(x, idx):
    """cat(x_j - x_i, x_i) -> [B,2C,N,k] (PointNet.py:150-171). Kept for callers that want the
    materialised tensor; the encoder itself never builds it (fused EdgeConv kernel)."""
    B, C, N = x.shape
    k = idx.shape[-1]
    xt = x.transpose(2, 1).contiguous()                                    # [B,N,C]
    flat = (idx + torch.arange(B, device=x.device).view(-1, 1, 1) * N).view(-1)
    feature = xt.view(B * N, C)[flat].view(B, N, k, C)
    ctr = xt.view(B, N, 1, C).expand(B, N, k, C)
    return torch.cat((feature - ctr, ctr), dim=3).permute(0, 3, 1, 2).contiguous()


def get_graph_feature(x, k1=20, k2=20, idx=None, Norm_sample=False):
    """PointNet.py:140-171."""
    x = x.contiguous().view(x.size(0), -1, x.size(2))
    if idx is None:
        idx = knn(x, k1=k1, k2=k2, normal=Norm_sample)
    return _gather_feature(x, idx)


def get_graph_feature_with_normals(x, k1=20, k2=20, idx=None, normal_metric_W=1., Norm_sample=False):
    """PointNet.py:174-208."""
    x = x.contiguous().view(x.size(0), -1, x.size(2))
    if idx is None:
        idx = knn_points_normals(x, k1=k1, k2=k2, normal_metric_W=normal_metric_W, normal=Norm_sample)
    return _gather_feature(x, idx)
