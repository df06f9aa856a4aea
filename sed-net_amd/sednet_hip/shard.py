"""Multi-GPU layout: clouds are independent (the reference loops over them one by one,
generate_predictions_aug.py:213; GroupNorm and mean-shift are per cloud), so a batch shards by cloud index with no
data-path collective. One process per GPU; the only exchange is the final gather of the per-cloud results
(RCCL all_gather over xGMI on the GPU box, gloo in the CPU tests)."""
import torch


def shard_range(n_items, rank, world):
    """contiguous block of cloud indices owned by `rank` (first `n_items % world` ranks get one extra)."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


GATHER_KEYS = ("labels", "types", "params", "valid", "seg_type")


def gather_results(out, dist, keys=GATHER_KEYS):
    """all_gather the per-cloud result tensors of every rank (equal shard sizes) -> dict with the leading
    dimension world * B, ordered by rank (= by global cloud index for contiguous shards)."""
    world = dist.get_world_size()
    host_staged = dist.get_backend() != "nccl"        # gloo (tests): collectives on host copies
    res = dict(out)
    for k in keys:
        if k not in out:
            continue
        t = out[k].contiguous()
        src = t.cpu() if (host_staged and t.is_cuda) else t
        full = torch.empty((world * src.shape[0],) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
        dist.all_gather_into_tensor(full, src)
        res[k] = full.to(t.device) if host_staged else full
    return res


def gather_ragged(t, dist):
    """all_gather for unequal shard sizes along dim 0 (pads to the maximum, trims after)."""
    world = dist.get_world_size()
    n = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    m = max(sizes)
    pad = torch.zeros((m,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[:t.shape[0]] = t
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    return torch.cat([p[:s] for p, s in zip(parts, sizes)], 0)


def allreduce_gradients(model, dist, average=True):
    """Data-parallel training: average the gradients of all ranks with ONE flat all-reduce (the model has 1.35 M fp32
    parameters = 5.4 MB, far below the size where bucketing would pay on xGMI; the reference's DataParallel,
    train_sed_net.py:149-150, reduces per parameter). Parameters without a gradient on this rank contribute zeros, so
    every rank reduces the same layout."""
    params = [p for p in model.parameters() if p.requires_grad]
    if not params:
        return
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params])
    staged = flat if dist.get_backend() != "gloo" else flat.cpu()
    dist.all_reduce(staged, op=dist.ReduceOp.SUM)
    if average:
        staged /= dist.get_world_size()
    flat = staged.to(flat.device)
    o = 0
    for p in params:
        n = p.numel()
        p.grad = flat[o:o + n].view_as(p).clone()
        o += n
