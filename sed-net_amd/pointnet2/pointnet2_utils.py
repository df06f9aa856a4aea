"""PointNet++ operators -- forward surface of
/root/reference/Fitting_patches_and_edges/pointnet2/pointnet2_utils.py:48-290 (furthest_point_sample, gather_operation,
three_nn, three_interpolate, grouping_operation, ball_query) on the gfx950 kernels of pointops.hip. The forward
kernels are HIP; the three differentiable operators (gather / grouping / three_interpolate) carry their backward as a
deterministic device-side scatter-add under torch.autograd (the reference's group_points_grad / interpolate_grad
kernels, group_points_gpu.cu:47-80, interpolate_gpu.cu:111-148) -- nothing in SED-Net's models calls them, so no
dedicated gradient kernel was written. CPU tensors raise (the reference's extension also refuses them:
ball_query.cpp:33 "CPU not supported")."""
import torch

from sednet_hip._lib import check, lib, ptr, stream


def furthest_point_sample(xyz, npoint):
    """xyz [B,N,3] -> idx [B,npoint] int32 (pointnet2_utils.py:48-72)."""
    xyz = xyz.contiguous().float()
    B, N, _ = xyz.shape
    idx = torch.empty(B, npoint, dtype=torch.int, device=xyz.device)
    tmp = torch.empty(B * N, device=xyz.device)
    check(lib.sed_furthest_point_sampling_f32(B, N, npoint, ptr(xyz), ptr(tmp), ptr(idx), stream()), "fps")
    return idx


def _scatter_grad(grad, idx_flat, N, weight=None):
    """d/d features of out[b,c,j] = w[b,j] * features[b,c,idx[b,j]]: grad [B,C,J], idx_flat [B,J] -> [B,C,N]."""
    B, C, J = grad.shape
    g = grad if weight is None else grad * weight[:, None, :]
    return torch.zeros(B, C, N, dtype=grad.dtype, device=grad.device).scatter_add_(
        2, idx_flat.long()[:, None, :].expand(B, C, J), g.contiguous())


class _Gather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, idx):
        B, C, N = features.shape
        m = idx.shape[1]
        out = torch.empty(B, C, m, device=features.device)
        check(lib.sed_group_points_f32(B, C, N, m, 1, ptr(features), ptr(idx), ptr(out), stream()), "gather_points")
        ctx.save_for_backward(idx)
        ctx.N = N
        return out

    @staticmethod
    def backward(ctx, grad):
        return _scatter_grad(grad.contiguous(), ctx.saved_tensors[0], ctx.N), None


def gather_operation(features, idx):
    """features [B,C,N], idx [B,npoint] -> [B,C,npoint] (:78-112)."""
    return _Gather.apply(features.contiguous().float(), idx.contiguous().int())


def three_nn(unknown, known):
    """unknown [B,n,3], known [B,m,3] -> (dist [B,n,3] = sqrt of squared distances, idx [B,n,3]) (:118-143)."""
    unknown, known = unknown.contiguous().float(), known.contiguous().float()
    B, n, _ = unknown.shape
    m = known.shape[1]
    dist2 = torch.empty(B, n, 3, device=unknown.device)
    idx = torch.empty(B, n, 3, dtype=torch.int, device=unknown.device)
    check(lib.sed_three_nn_f32(B, n, m, ptr(unknown), ptr(known), ptr(dist2), ptr(idx), stream()), "three_nn")
    return torch.sqrt(dist2), idx


class _Interpolate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, idx, weight):
        B, c, m = features.shape
        n = idx.shape[1]
        out = torch.empty(B, c, n, device=features.device)
        check(lib.sed_three_interpolate_f32(B, c, m, n, ptr(features), ptr(idx), ptr(weight), ptr(out), stream()),
              "three_interpolate")
        ctx.save_for_backward(idx, weight)
        ctx.m = m
        return out

    @staticmethod
    def backward(ctx, grad):
        idx, weight = ctx.saved_tensors
        B, c, n = grad.shape
        g3 = grad.contiguous()[:, :, :, None].expand(B, c, n, 3).reshape(B, c, 3 * n)
        return _scatter_grad(g3, idx.reshape(B, 3 * n), ctx.m, weight.reshape(B, 3 * n)), None, None


def three_interpolate(features, idx, weight):
    """features [B,c,m], idx/weight [B,n,3] -> [B,c,n] (:149-198)."""
    return _Interpolate.apply(features.contiguous().float(), idx.contiguous().int(), weight.contiguous().float())


class _Group(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, idx):
        B, C, N = features.shape
        _, npoint, nsample = idx.shape
        out = torch.empty(B, C, npoint, nsample, device=features.device)
        check(lib.sed_group_points_f32(B, C, N, npoint, nsample, ptr(features), ptr(idx), ptr(out), stream()),
              "group_points")
        ctx.save_for_backward(idx)
        ctx.N = N
        return out

    @staticmethod
    def backward(ctx, grad):
        idx = ctx.saved_tensors[0]
        B, C = grad.shape[:2]
        return _scatter_grad(grad.contiguous().reshape(B, C, -1), idx.reshape(B, -1), ctx.N), None


def grouping_operation(features, idx):
    """features [B,C,N], idx [B,npoint,nsample] -> [B,C,npoint,nsample] (:204-243)."""
    return _Group.apply(features.contiguous().float(), idx.contiguous().int())


def ball_query(radius, nsample, xyz, new_xyz):
    """xyz [B,N,3], new_xyz [B,npoint,3] -> idx [B,npoint,nsample] int32 (:249-280)."""
    xyz, new_xyz = xyz.contiguous().float(), new_xyz.contiguous().float()
    B, N, _ = xyz.shape
    m = new_xyz.shape[1]
    idx = torch.zeros(B, m, nsample, dtype=torch.int, device=xyz.device)
    check(lib.sed_ball_query_f32(B, N, m, float(radius), int(nsample), ptr(new_xyz), ptr(xyz), ptr(idx), stream()),
          "ball_query")
    return idx
