"""Edge-head losses -- same names and call contract as /root/reference/src/My_edge_loss.py:14-105."""
import torch
import torch.nn.functional as F


def edge_cls_loss(edges_pred, edges_label, bce_W):
    """Weighted 2-class cross entropy per cloud; clouds whose weights sum to 0 contribute 0.
    edges_pred [B,2,N] logits, edges_label [B,N], bce_W [B,N]."""
    per_cloud = (F.cross_entropy(edges_pred, edges_label, reduction="none") * bce_W).mean(-1)
    per_cloud = torch.where(bce_W.sum(-1) == 0, torch.zeros_like(per_cloud), per_cloud)
    return per_cloud.mean()


def compute_embedding_loss(pred_feat, gt_label, t_pull=0.5, t_push=1.5):
    """Pull every feature to within t_pull of its segment centre, push centres t_push apart.
    pred_feat [B,M,K], gt_label [B,M] (labels >= -1) -> (loss [1], pull [1], push [1])."""
    B = pred_feat.shape[0]
    dev = pred_feat.device
    pull = torch.zeros(1, device=dev)
    push = torch.zeros(1, device=dev)
    for i in range(B):
        # segment statistics by one-hot products instead of the reference's per-segment Python loop (same terms; a
        # scatter-add / index_add_ would do too, but its fp32 atomics -- forward and in the backward of C[inv] -- make the
        # gradients differ from run to run, and the training step is otherwise bit-reproducible)
        _, inv, cnt = torch.unique(gt_label[i], return_inverse=True, return_counts=True)
        S = cnt.shape[0]
        cntf = cnt.to(pred_feat.dtype)
        onehot = (inv[None, :] == torch.arange(S, device=dev)[:, None]).to(pred_feat.dtype)        # [S, M]
        C = (onehot @ pred_feat[i]) / cntf[:, None]
        excess = F.relu(torch.norm(pred_feat[i] - onehot.t() @ C, 2, dim=1) - t_pull)
        per_seg = (onehot @ excess) / cntf
        pull = pull + per_seg.sum() / S
        if S == 1:
            continue
        dist = torch.norm(C[:, None, :] - C[None, :, :], 2, dim=2)
        off = dist[~torch.eye(S, dtype=torch.bool, device=dev)]
        push = push + F.relu(t_push - off).mean()
    pull, push = pull / B, push / B
    return pull + push, pull, push


def compute_edge_embedding_loss(edges_pred, pred_feat, gt_label, edges_num=2000, use_type=False, primitives=None,
                                primitives_log_prob=None):
    """Embedding (and optionally type NLL) loss restricted to the edges_num points with the highest edge logit.
    edges_pred [B,2,N], pred_feat [B,K,N], gt_label [B,N], primitives [B,N], primitives_log_prob [B,C,N]."""
    order = torch.argsort(edges_pred[:, 1, :], dim=-1, descending=True)[:, :edges_num]
    feat = torch.gather(pred_feat.transpose(1, 2), 1, order.unsqueeze(-1).expand(-1, -1, pred_feat.shape[1]))
    lab = torch.gather(gt_label, 1, order)
    emb = torch.mean(compute_embedding_loss(feat.contiguous(), lab.contiguous())[0])
    if not use_type:
        return emb
    C = primitives_log_prob.shape[1]
    lp = torch.gather(primitives_log_prob.transpose(1, 2), 1, order.unsqueeze(-1).expand(-1, -1, C))
    return F.nll_loss(lp.transpose(1, 2), torch.gather(primitives, 1, order)) + emb
