import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sed-net_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from sednet_hip import synth, ops, _lib
B, iters, N, d = 32, 6, 10000, 128
X = np.stack([synth.clustered_embedding(N=N, d=d, n_clusters=12 + b % 8, sigma=0.01, seed=b)[0] for b in range(B)])
X = torch.from_numpy(X).cuda(); out = torch.empty_like(X)
bw = torch.full((B,), 0.16, device="cuda")
fn = _lib.lib.sed_ms_iterate_variant
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p] * 4
names = {0: "full", 1: "noDMA", 2: "noExp", 3: "noDMA+noExp", 4: "noBarrier", 5: "noDMA+noBar", 6: "noExp+noBar", 7: "none", 8: "generic (non-pipelined)", 9: "generic noStage", 10: "generic noExp", 12: "generic noBarrier", 15: "generic none"}
for v in [int(a) for a in sys.argv[1:]] or list(names):
    for rep in range(2):
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record(); rc = fn(v, B, N, iters, bw.data_ptr(), X.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream); e.record()
        torch.cuda.synchronize()
    ms = s.elapsed_time(e)
    print(f"variant {v} {names[v]:24s} rc {rc} ms {ms:8.2f} TFLOP/s {4*N*N*d*iters*B/ms/1e9:7.1f}", flush=True)
