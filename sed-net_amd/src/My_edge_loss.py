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
    # all clouds in one set of launches: segment statistics by one-hot products over the label range of the batch (the reference
    # loops over clouds and segments in Python; round 2 looped over clouds -- torch.unique + ~15 small launches each, forward and
    # backward). Segments a cloud does not have are masked out; the terms are the reference's. (A scatter-add would do too, but
    # its fp32 atomics -- forward and in the backward of the gather -- make the gradients differ from run to run, and the training
    # step is otherwise bit-reproducible.)
    B = pred_feat.shape[0]
    dev, dt = pred_feat.device, pred_feat.dtype
    # label ids are compacted to their rank among the ids present in the batch first: L is then the number of distinct
    # labels, not the label range -- sparse or large ids (500 apart) would otherwise size the [B, L, L, K] centre differences by the
    # range (4 GB at L = 500, B = 32, K = 128). torch.unique returns sorted ids: deterministic.
    uniq, lab = torch.unique(gt_label, return_inverse=True)
    L = int(uniq.shape[0])
    onehot = (lab[:, None, :] == torch.arange(L, device=dev)[None, :, None]).to(dt)                    # [B, L, M]
    cnt = onehot.sum(2)                                                                                # [B, L]
    present = cnt > 0
    S = present.sum(1).to(dt)                                                                          # segments per cloud
    cnt1 = cnt.clamp(min=1.0)
    C = torch.bmm(onehot, pred_feat) / cnt1[:, :, None]                                                # [B, L, K] centres
    excess = F.relu(torch.norm(pred_feat - torch.bmm(onehot.transpose(1, 2), C), 2, dim=2) - t_pull)   # [B, M]
    per_seg = torch.bmm(onehot, excess[:, :, None]).squeeze(2) / cnt1                                  # 0 for absent segments
    pull = (per_seg.sum(1) / S).sum().reshape(1) / B
    dist = torch.norm(C[:, :, None, :] - C[:, None, :, :], 2, dim=3)                                   # [B, L, L]
    pair = present[:, :, None] & present[:, None, :] & ~torch.eye(L, dtype=torch.bool, device=dev)[None]
    npair = (S * (S - 1.0)).clamp(min=1.0)                                                             # S = 1: no pairs, term 0
    push = ((F.relu(t_push - dist) * pair.to(dt)).sum((1, 2)) / npair).sum().reshape(1) / B
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
