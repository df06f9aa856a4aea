"""Oracle for the training step: a plain-torch, CPU, fp32 functional restatement of the
reference forward and of its training losses, differentiated by torch.autograd exactly as the reference is
(train_sed_net.py:233-285 -> loss.backward()).

Test infrastructure only -- see oracle/__init__.py. Pinned by tests/golden/f_train.npz, which holds losses and
gradients of the reference model itself (tests/golden/make_golden.py gen_train, run under ref_shim).

Follows /root/reference/src/PointNet.py:62-208 (kNN graph + edge features), /root/reference/src/SEDNet.py:78-98 and
:292-342 (encoder + heads), /root/reference/src/segment_loss.py:33-126, :209-226 (triplet loss, label smoothing) and
/root/reference/src/My_edge_loss.py:14-105 (edge losses).
"""
import numpy as np
import torch
import torch.nn.functional as F


# ---- graph ------------------------------------------------------------------------------------------------------
def knn_l2(x, k):
    """x [B,C,N] -> idx [B,N,k] (PointNet.py:62-87): topk of -|x_i - x_j|^2 in the reference's algebraic form."""
    with torch.no_grad():
        inner = -2 * torch.matmul(x.transpose(2, 1), x)
        xx = torch.sum(x ** 2, dim=1, keepdim=True)
        score = -xx - inner - xx.transpose(2, 1)
        return score.topk(k=k, dim=-1)[1]


def knn_points_normals(x, k, W=1.0):
    """x [B,6,N] -> idx [B,N,k] (PointNet.py:90-137)."""
    with torch.no_grad():
        p, n = x[:, 0:3], x[:, 3:6]
        inner = 2 * torch.matmul(p.transpose(2, 1), p)
        xx = torch.sum(p ** 2, dim=1, keepdim=True)
        dp = xx - inner + xx.transpose(2, 1)
        dn = 2 - 2 * torch.matmul(n.transpose(2, 1), n)
        score = -(dp * (1 + W * dn))
        return score.topk(k=k, dim=-1)[1]


def graph_feature(x, idx):
    """x [B,C,N], idx [B,N,k] -> [B,2C,N,k] = cat(x_j - x_i, x_i) (PointNet.py:150-171)."""
    B, C, N = x.shape
    k = idx.shape[2]
    xt = x.transpose(2, 1).contiguous()                                      # [B,N,C]
    flat = (idx + torch.arange(B).view(B, 1, 1) * N).reshape(-1)
    nbr = xt.reshape(B * N, C)[flat].reshape(B, N, k, C)
    ctr = xt.unsqueeze(2).expand(B, N, k, C)
    return torch.cat([nbr - ctr, ctr], dim=3).permute(0, 3, 1, 2)


# ---- model ------------------------------------------------------------------------------------------------------
def _edge_block(x, idx, W, gamma, beta, G):
    y = F.conv2d(graph_feature(x, idx), W)
    y = F.leaky_relu(F.group_norm(y, G, gamma, beta, 1e-5), 0.2)
    return y.max(dim=-1)[0]


def _cgr(x, p, conv, bn, G, act=True):
    y = F.group_norm(F.conv1d(x, p[conv + ".weight"], p[conv + ".bias"]), G, p[bn + ".weight"], p[bn + ".bias"], 1e-5)
    return F.relu(y) if act else y


def sednet_forward(p, points, k, idx=None, w_pos_enc=0.2, normal_metric_W=1.0):
    """p: dict name -> torch tensor (reference state-dict keys); points [B,6,N].
    idx: optional (idx1, idx2, idx3) to pin the neighbour sets (parity runs feed the device's graphs).
    -> (embedding [B,128,N], log_prob [B,6,N], edges [B,2,N], (idx1, idx2, idx3))."""
    x = points
    i1 = knn_points_normals(x, k, normal_metric_W) if idx is None else idx[0]
    x1 = _edge_block(x, i1, p["encoder.conv1.0.weight"], p["encoder.bn1.weight"], p["encoder.bn1.bias"], 2)
    i2 = knn_l2(x1, k) if idx is None else idx[1]
    x2 = _edge_block(x1, i2, p["encoder.conv2.0.weight"], p["encoder.bn2.weight"], p["encoder.bn2.bias"], 2)
    i3 = knn_l2(x2, k) if idx is None else idx[2]
    x3 = _edge_block(x2, i3, p["encoder.conv3.0.weight"], p["encoder.bn3.weight"], p["encoder.bn3.bias"], 2)
    feats = torch.cat([x1, x2, x3], dim=1)
    y = F.relu(F.group_norm(F.conv1d(feats, p["encoder.mlp1.weight"], p["encoder.mlp1.bias"]), 8,
                            p["encoder.bnmlp1.weight"], p["encoder.bnmlp1.bias"], 1e-5))
    x4 = y.max(dim=2)[0]
    N = points.shape[2]
    x = torch.cat([x4.unsqueeze(2).expand(-1, -1, N), feats], dim=1)                       # SEDNet.py:300-301
    x = _cgr(x, p, "conv1", "bn1", 8)
    x_all = _cgr(x, p, "conv2", "bn2", 4)
    x_type = _cgr(x_all, p, "mlp_prim_prob1", "bn_prim_prob1", 4)                           # :311
    type_logit = F.conv1d(x_type, p["mlp_prim_prob2.weight"], p["mlp_prim_prob2.bias"])
    log_prob = F.log_softmax(type_logit, dim=1)
    e = _cgr(x_type, p, "edge_module.0", "edge_module.1", 4, act=False)                     # :249-253
    edges = F.conv1d(e, p["edge_module.2.weight"], p["edge_module.2.bias"])
    xs = _cgr(x_all, p, "mlp_seg_prob1", "bn_seg_prob1", 4)                                 # :320
    xs = w_pos_enc * _cgr(x_type, p, "asis.0", "asis.1", 4) + xs                            # :322
    pe = F.relu(F.conv1d(torch.cat([type_logit.detach(), edges.detach()], dim=1), p["prim_encoding.0.weight"],
                         p["prim_encoding.0.bias"]))                                        # :326 detaches both inputs
    xs = xs + w_pos_enc * pe
    emb = F.conv1d(xs, p["mlp_seg_prob2.weight"], p["mlp_seg_prob2.bias"])                 # :329
    return emb, log_prob, edges, (i1, i2, i3)


# ---- losses -----------------------------------------------------------------------------------------------------
def triplet_loss(output, labels, margin=1.0, max_segments=5, rng=np.random):
    """segment_loss.py:33-126 (if_mean_shift=False). output [B,D,N], labels [B,N] numpy.
    Consumes `rng` exactly like the reference consumes np.random (choice per segment, then 2 draws per pair)."""
    B, _, N = output.shape
    out = F.normalize(output.permute(0, 2, 1), p=2, dim=2)
    sampled = []
    for i in range(B):
        uniq = np.unique(labels[i])
        ns = min(N // uniq.shape[0] + 1, 30)
        sampled.append({l: rng.choice(list(np.where(np.isin(labels[i], l))[0]), ns, replace=True) for l in uniq})
    total = torch.zeros(1)
    only_one = 0
    for i in range(B):
        keys = sorted(sampled[i].keys())
        nk = len(keys)
        if nk == 1:
            only_one += 1
            continue
        shape_loss = torch.zeros(1)
        norm = 0
        for _ in range(min(max_segments * max_segments, nk * nk)):
            k1 = rng.choice(nk, 1)[0]
            k2 = rng.choice(nk, 1)[0]
            if k1 == k2:
                continue
            norm += 1
            a = out[i, sampled[i][keys[k1]], :]
            b = out[i, sampled[i][keys[k2]], :]
            d_pos = ((a.unsqueeze(1) - a.unsqueeze(0)) ** 2).sum(2)
            d_neg = ((a.unsqueeze(1) - b.unsqueeze(0)) ** 2).sum(2)
            c = F.relu(d_pos - d_neg + margin)
            loss = c.sum() - c.trace()
            satisfied = ((c > 0).sum() + 1.0).float()
            shape_loss = shape_loss + loss / satisfied.detach()
        total = total + shape_loss / (norm + 1e-8)
    return total / (B - only_one + 1e-8)


def label_smoothing_nll(logprobs, target, smoothing):
    """segment_loss.py:209-226. logprobs [M,C], target [M]."""
    nll = -logprobs.gather(dim=-1, index=target.unsqueeze(1)).squeeze(1)
    return ((1.0 - smoothing) * nll + smoothing * (-logprobs.mean(dim=-1))).mean()


def edge_cls_loss(edges_pred, edges_label, w):
    """My_edge_loss.py:14-25. edges_pred [B,2,N] logits, edges_label [B,N] int64, w [B,N]."""
    l = (F.cross_entropy(edges_pred, edges_label, reduction="none") * w).mean(-1)
    l = torch.where(w.sum(-1) == 0, torch.zeros_like(l), l)
    return l.mean()


def pull_push_loss(feat, label, t_pull=0.5, t_push=1.5):
    """My_edge_loss.py:29-85. feat [B,M,K], label [B,M] (values >= -1)."""
    B = feat.shape[0]
    pull = torch.zeros(1)
    push = torch.zeros(1)
    for i in range(B):
        groups = [feat[i][label[i] == v] for v in range(-1, int(label[i].max()) + 1)]
        groups = [g for g in groups if len(g) > 0]
        centres = [g.mean(0, keepdim=True) for g in groups]
        pl = torch.zeros(1)
        for g, c in zip(groups, centres):
            pl = pl + F.relu(torch.norm(g - c, 2, dim=1) - t_pull).mean()
        pull = pull + pl / len(groups)
        C = torch.cat(centres, 0)
        if C.shape[0] == 1:
            continue
        dst = torch.norm(C[:, None, :] - C[None, :, :], 2, dim=2)
        off = dst[~torch.eye(C.shape[0], dtype=torch.bool)]
        push = push + F.relu(t_push - off).mean()
    return pull / B + push / B


def edge_embedding_loss(edges_pred, emb, labels, primitives, log_prob, edges_num=2000):
    """My_edge_loss.py:88-105 with use_type=True."""
    order = torch.argsort(edges_pred[:, 1, :], dim=-1, descending=True)[:, :edges_num]
    f = torch.gather(emb.transpose(1, 2), 1, order.unsqueeze(-1).expand(-1, -1, emb.shape[1]))
    l = torch.gather(labels, 1, order)
    lp = torch.gather(log_prob.transpose(1, 2), 1, order.unsqueeze(-1).expand(-1, -1, log_prob.shape[1]))
    pr = torch.gather(primitives, 1, order)
    return F.nll_loss(lp.transpose(1, 2), pr) + pull_push_loss(f, l).mean()


def training_loss(emb, log_prob, edges, labels, primitives, edge_labels, edge_w, smoothing=0.025, rng=np.random):
    """train_sed_net.py:250-271: embed + type + edge + 0.25 * edge-embedding. -> (loss, dict of parts)."""
    embed = triplet_loss(emb, labels.numpy(), rng=rng).mean()
    prim = primitives.clone()
    prim[(prim == 9) | (prim == 6) | (prim == 7)] = 0                                  # :253-254
    prim[prim == 8] = 2
    e_loss = edge_cls_loss(edges, edge_labels, edge_w)
    p_loss = label_smoothing_nll(log_prob.transpose(1, 2).reshape(-1, log_prob.shape[1]), prim.reshape(-1), smoothing)
    ee = edge_embedding_loss(edges, emb, labels, prim, log_prob)
    loss = embed + p_loss + e_loss + 0.25 * ee
    return loss, {"embed": embed, "type": p_loss, "edge": e_loss, "edge_embed": ee}
