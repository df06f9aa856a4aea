"""Time the d = 128 mean-shift iteration schedules on the GPU: python tools/ms_f16_bench.py [B] [N] [iters] [variants]
variants: comma list of schedule[/digits][@wave_queries], e.g. f16@32,f16@64,f16/2@64,f16c@64,batched"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sed-net_amd")]
import numpy as np, torch
from sednet_hip import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 50
variants = sys.argv[4].split(",") if len(sys.argv) > 4 else ["f16", "f16c", "batched"]
g = torch.Generator().manual_seed(0)
cent = torch.nn.functional.normalize(torch.randn(B, 14, 128, generator=g), dim=2)
X = cent[:, torch.arange(N) % 14] + 0.02 * torch.randn(B, N, 128, generator=g)
X = torch.nn.functional.normalize(X, dim=2).cuda().contiguous()
if os.environ.get("BLOB"):                # one blob (what the bench's closed-form weights produce): nothing to skip, little toggling
    X = torch.nn.functional.normalize(torch.tensor([1.0] + [0.0] * 127) + 0.03 * torch.randn(B, N, 128, generator=g), dim=2).cuda().contiguous()
if os.environ.get("CONST_ROWS"):          # identical rows: same instruction stream, (almost) no operand toggling -> DVFS probe
    X = torch.zeros_like(X); X[:, :, 0] = 1.0
bw = ops.ms_bandwidth(X, 150, 0.003)
print("bw", bw[:4].tolist())
res = {}
for v in variants:
    name, _, wq = v.partition("@")
    name, _, dg = name.partition("/")
    ops.ms_set_variant(name)
    ops.ms_set_weight_digits(int(dg) if dg else 1)
    ops.MS_WAVE_QUERIES = int(wq) if wq else 0
    out = ops.ms_iterate(X, bw, iters); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); out = ops.ms_iterate(X, bw, iters); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    fl = 4.0 * N * N * 128 * iters * B
    res[v] = out
    print(f"{v:8s} {ms:9.2f} ms  {fl / ms / 1e9:8.1f} TFLOP/s fp32-equivalent  (x3 = {3 * fl / ms / 1e9:8.1f} TF/s of fp16 MFMA work)", flush=True)
ops.ms_set_variant("auto"); ops.ms_set_weight_digits(1); ops.MS_WAVE_QUERIES = 0
ks = list(res)
for k in ks[1:]:
    print(f"max |{ks[0]} - {k}| = {(res[ks[0]] - res[k]).abs().max().item():.3e}")
