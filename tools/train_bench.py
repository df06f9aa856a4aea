#!/usr/bin/env python
"""Time the training step: reference configuration batch_size = 4, 10 000 points, k = 64
(configs/config_SEDNet_normal.yml:30,37,46). Prints ms per step and the forward / backward split.
    python tools/train_bench.py [B] [N] [k] [steps]
    python -m torch.distributed.run --nproc-per-node G --master-addr 127.0.0.1 tools/train_bench.py B N k steps
Data-parallel: B clouds PER RANK, gradients averaged with one flat all-reduce per step (RCCL; SED_BENCH_BACKEND=gloo
lets several ranks share one GPU for a functional check)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "sed-net_amd"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from sednet_hip import synth  # noqa: E402
from sednet_hip.train import train_step, training_loss  # noqa: E402
from src.SEDNet import SEDNet  # noqa: E402
from train_case import train_case  # noqa: E402

from sednet_hip import ops  # noqa: E402

argv = [a for a in sys.argv[1:] if a != "--bf16"]
ops.TRAIN_BF16 = "--bf16" in sys.argv                    # bf16 products (BASELINE configs[4]); default: fp32 like the reference
B, N, k, steps = (int(a) for a in (argv + ["4", "10000", "64", "5"][len(argv):]))
world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
dist = None
torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
if world > 1:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    backend = os.environ.get("SED_BENCH_BACKEND", "nccl")
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", torch.cuda.current_device()))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
x, labels, types, edges, edges_w, _ = train_case(synth, N, B, seed0=700 + 10 * rank)       # this rank's shard
m = SEDNet(embedding=True, emb_size=128, primitives=True, num_primitives=6, mode=5, num_channels=6,
           combine_label_prim=True, edge_module=True, late_fusion=True, nn_nb=k)
m.load_state_dict({n: torch.from_numpy(v) for n, v in synth.closed_form_state_dict(4).items()})
m = m.cuda().train()
opt = torch.optim.AdamW(m.parameters(), lr=1e-4, weight_decay=0.0)
batch = tuple(torch.from_numpy(a).cuda() for a in (x, labels, types, edges, edges_w))
for _ in range(2):
    train_step(m, opt, batch, dist=dist)
torch.cuda.synchronize()
if dist is not None:
    dist.barrier()
t0 = time.perf_counter()
for _ in range(steps):
    out = train_step(m, opt, batch, dist=dist)
torch.cuda.synchronize()
if dist is not None:
    dist.barrier()
ms = (time.perf_counter() - t0) / steps * 1e3
# forward-only share
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    with torch.enable_grad():
        loss, _ = training_loss(m, *batch)
torch.cuda.synchronize()
fwd = (time.perf_counter() - t0) / steps * 1e3
if rank == 0:
    print(f"train step ranks={world} B={B}/rank N={N} k={k}: {ms:.1f} ms/step ({B * world / ms * 1e3:.1f} clouds/s), "
          f"forward+loss {fwd:.1f} ms, backward+all-reduce+optimizer {ms - fwd:.1f} ms, loss {out['loss']:.4f}, "
          f"peak mem {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB")
if dist is not None:
    # replicas must stay identical: same parameters on every rank after the averaged update
    chk = torch.cat([p.detach().reshape(-1)[:64] for p in m.parameters()]).cpu()
    ref = chk.clone()
    dist.broadcast(ref, 0)
    assert torch.equal(chk, ref), "replicas diverged"
    dist.destroy_process_group()
