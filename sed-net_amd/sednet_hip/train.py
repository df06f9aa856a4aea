"""One training step of the reference's train_sed_net.py:233-285 on the HIP path.

    embedding, log_prob, _, edges = model(points)            fused HIP forward (autograd.py)
    loss = triplet + label-smoothed type NLL + weighted edge CE + 0.25 * edge-embedding loss
    loss.backward()                                          HIP backward (edgeconv_bwd.hip) + library GEMMs
    [all-reduce of the 5.4 MB of gradients over RCCL when world_size > 1]
    optimizer.step()

The reference wraps the model in torch.nn.DataParallel (train_sed_net.py:149-150); here every rank owns its shard of
the batch (one process per GPU) and gradients are averaged with one flat all-reduce (shard.allreduce_gradients).
"""
import torch

from src.My_edge_loss import compute_edge_embedding_loss, edge_cls_loss
from src.segment_loss import EmbeddingLoss, LabelSmoothingLoss

_EMB = EmbeddingLoss(margin=1.0)


def remap_primitive_types(primitives):
    """train_sed_net.py:253-254: spline types {6,7,9} -> 0, 8 -> 2 (returns a new tensor)."""
    p = primitives.clone()
    p[(p == 9) | (p == 6) | (p == 7)] = 0
    p[p == 8] = 2
    return p


def training_loss(model, points, labels, primitives, edges, edges_W, smoothing=0.025):
    """points [B,6,N] on the device -> (loss tensor, dict of python floats)."""
    embedding, log_prob, _, edges_pred = model(points=points)
    embed = torch.mean(_EMB.triplet_loss(embedding, labels.cpu().numpy()))
    prim = remap_primitive_types(primitives)
    e_loss = edge_cls_loss(edges_pred, edges, edges_W)
    p_loss = LabelSmoothingLoss(smoothing)(log_prob.transpose(1, 2).contiguous().view(-1, log_prob.shape[1]),
                                           prim.contiguous().view(-1))
    ee = compute_edge_embedding_loss(edges_pred=edges_pred, pred_feat=embedding, gt_label=labels, use_type=True,
                                     primitives=prim, primitives_log_prob=log_prob)
    loss = embed + p_loss + e_loss + 0.25 * ee
    return loss, {"embed": embed.item(), "type": p_loss.item(), "edge": e_loss.item(), "edge_embed": ee.item()}


def train_step(model, optimizer, batch, smoothing=0.025, dist=None):
    """batch = (points [B,6,N], labels [B,N], primitives [B,N], edges [B,N], edges_W [B,N]) on the device."""
    from .shard import allreduce_gradients
    points, labels, primitives, edges, edges_W = batch
    optimizer.zero_grad()
    loss, parts = training_loss(model, points, labels, primitives, edges, edges_W, smoothing)
    loss.backward()
    if dist is not None:
        allreduce_gradients(model, dist)
    optimizer.step()
    parts["loss"] = loss.item()
    return parts
