"""Oracle: DGCNN encoder (mode 5) + SED-Net heads (numpy, fp32 data, fp64 GroupNorm statistics).

Test infrastructure only -- see oracle/__init__.py.
Follows /root/reference/src/SEDNet.py:19-98 (encoder) and :216-342 (heads).
`params` is a dict keyed exactly like the reference state-dict.
"""
import numpy as np

from . import graph

F32 = np.float32


def group_norm(x, G, gamma, beta, eps=1e-5):
    """torch.nn.GroupNorm over x [B,C,*]: statistics per (sample, group) over all
    remaining positions, biased variance (SEDNet.py:31-45 use nn.GroupNorm defaults)."""
    B, C = x.shape[:2]
    xs = x.reshape(B, G, -1).astype(np.float64)
    mean = xs.mean(-1, keepdims=True)
    var = xs.var(-1, keepdims=True)
    y = ((xs - mean) / np.sqrt(var + eps)).astype(F32).reshape(x.shape)
    shape = (1, C) + (1,) * (x.ndim - 2)
    return y * gamma.reshape(shape).astype(F32) + beta.reshape(shape).astype(F32)


def conv1x1(x, W, b=None):
    """Conv1d/Conv2d with kernel 1: x [B,Cin,...] , W [Cout,Cin(,1(,1))] -> [B,Cout,...]."""
    W = W.reshape(W.shape[0], W.shape[1]).astype(F32)
    B = x.shape[0]
    flat = x.reshape(B, x.shape[1], -1)
    y = np.stack([W @ flat[i] for i in range(B)], 0).astype(F32)
    if b is not None:
        y = y + b.reshape(1, -1, 1).astype(F32)
    return y.reshape((B, W.shape[0]) + x.shape[2:])


def leaky_relu(x, slope=0.2):
    return np.where(x >= 0, x, x * F32(slope)).astype(F32)


def relu(x):
    return np.maximum(x, F32(0)).astype(F32)


def edge_conv(x, idx, W, gamma, beta, G):
    """One EdgeConv block: graph feature -> Conv2d 1x1 (no bias) -> GroupNorm(G)
    -> LeakyReLU(0.2) -> max over k (SEDNet.py:80-82 with :37-45)."""
    feat = graph.graph_feature_from_idx(x, idx)             # [B,2C,N,k]
    y = conv1x1(feat, W)
    y = leaky_relu(group_norm(y, G, gamma, beta))
    return y.max(axis=-1)


def encoder_forward(params, x, k, normal_metric_W=1.0, return_idx=False):
    """DGCNNEncoderGn.forward, mode 5 (SEDNet.py:78-98). x [B,6,N] -> (x4 [B,1024], feats [B,256,N])."""
    p = params
    x = np.asarray(x, F32)
    idx1 = graph.knn_points_normals(x, k, k, normal_metric_W)
    x1 = edge_conv(x, idx1, p["encoder.conv1.0.weight"], p["encoder.bn1.weight"], p["encoder.bn1.bias"], 2)
    idx2 = graph.knn(x1, k, k)
    x2 = edge_conv(x1, idx2, p["encoder.conv2.0.weight"], p["encoder.bn2.weight"], p["encoder.bn2.bias"], 2)
    idx3 = graph.knn(x2, k, k)
    x3 = edge_conv(x2, idx3, p["encoder.conv3.0.weight"], p["encoder.bn3.weight"], p["encoder.bn3.bias"], 2)
    feats = np.concatenate([x1, x2, x3], axis=1)            # [B,256,N]
    y = conv1x1(feats, p["encoder.mlp1.weight"], p["encoder.mlp1.bias"])
    y = relu(group_norm(y, 8, p["encoder.bnmlp1.weight"], p["encoder.bnmlp1.bias"]))
    x4 = y.max(axis=2)
    if return_idx:
        return x4, feats, (idx1, idx2, idx3), (x1, x2, x3)
    return x4, feats


def log_softmax(x, axis):
    m = x.max(axis=axis, keepdims=True)
    z = x - m
    return (z - np.log(np.sum(np.exp(z.astype(np.float64)), axis=axis, keepdims=True))).astype(F32)


def sednet_forward(params, points, k, w_pos_enc=0.2, normal_metric_W=1.0):
    """SEDNet.forward(points, None, False) with embedding, primitives, edge_module,
    combine_label_prim and late_fusion all on (generate_predictions_aug.py:142-167;
    SEDNet.py:292-342). points [B,6,N] -> (embedding [B,128,N], log_prob [B,6,N], edges [B,2,N])."""
    p = params
    x4, feats = encoder_forward(p, points, k, normal_metric_W)
    B, _, N = feats.shape
    x = np.concatenate([np.repeat(x4[:, :, None], N, axis=2), feats], axis=1)      # :300-301
    x = relu(group_norm(conv1x1(x, p["conv1.weight"], p["conv1.bias"]), 8, p["bn1.weight"], p["bn1.bias"]))
    x_all = relu(group_norm(conv1x1(x, p["conv2.weight"], p["conv2.bias"]), 4, p["bn2.weight"], p["bn2.bias"]))

    x_type = relu(group_norm(conv1x1(x_all, p["mlp_prim_prob1.weight"], p["mlp_prim_prob1.bias"]), 4,
                             p["bn_prim_prob1.weight"], p["bn_prim_prob1.bias"]))   # :311
    type_logit = conv1x1(x_type, p["mlp_prim_prob2.weight"], p["mlp_prim_prob2.bias"])
    log_prob = log_softmax(type_logit, 1)

    e = conv1x1(x_type, p["edge_module.0.weight"], p["edge_module.0.bias"])        # :249-253, :316-317
    e = group_norm(e, 4, p["edge_module.1.weight"], p["edge_module.1.bias"])
    edges = conv1x1(e, p["edge_module.2.weight"], p["edge_module.2.bias"])

    x = relu(group_norm(conv1x1(x_all, p["mlp_seg_prob1.weight"], p["mlp_seg_prob1.bias"]), 4,
                        p["bn_seg_prob1.weight"], p["bn_seg_prob1.bias"]))          # :320
    a = relu(group_norm(conv1x1(x_type, p["asis.0.weight"], p["asis.0.bias"]), 4,
                        p["asis.1.weight"], p["asis.1.bias"]))
    x = F32(w_pos_enc) * a + x                                                      # :322
    pe = relu(conv1x1(np.concatenate([type_logit, edges], axis=1),
                      p["prim_encoding.0.weight"], p["prim_encoding.0.bias"]))
    x = x + F32(w_pos_enc) * pe                                                     # :326
    embedding = conv1x1(x, p["mlp_seg_prob2.weight"], p["mlp_seg_prob2.bias"])     # :329
    return embedding, log_prob, edges
