"""A/B of two builds of the head GEMMs (ops.pointwise, 3-way bf16 split): digests of every output (Y, GroupNorm statistics,
column extrema) over shapes that exercise ragged tiles, and the time of the nine wide layers of one forward at B x N.
    SEDHIP_LIB=<build>.so python tools/pointwise_ab.py [B N] > a.txt   (once per build; the digest lines must be identical)"""
import hashlib, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "sed-net_amd"))
import torch
from sednet_hip import ops, _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10000


def dig(t):
    return "-" if t is None else hashlib.sha256(t.detach().contiguous().cpu().numpy().tobytes()).hexdigest()[:16]


g = torch.Generator().manual_seed(0)
print("lib", os.path.basename(_lib.LIB_PATH))
CASES = [(2, 10000, 256, 1024, ops.F_STATS | ops.F_COLEXT, 8, False), (2, 10000, 256, 512, ops.F_STORE | ops.F_STATS, 8, True),
         (3, 900, 512, 256, ops.F_STORE | ops.F_STATS, 4, False), (2, 257, 256, 256, ops.F_STORE | ops.F_STATS, 4, False),
         (1, 128, 256, 128, ops.F_STORE | ops.F_STATS, 4, False), (2, 129, 32, 256, ops.F_STORE | ops.F_RELU, 0, False),
         (1, 31, 256, 128, ops.F_STORE, 0, False), (2, 1500, 256, 50, ops.F_STORE, 0, False),
         (1, 385, 256, 1024, ops.F_STATS | ops.F_COLEXT, 8, False), (2, 2000, 256, 256, ops.F_STORE | ops.F_STATS | ops.F_COLEXT, 4, True)]
for b, n, K, Cout, flags, G, use_cb in ([] if os.environ.get('PW_TIME_ONLY') else CASES):
    Coutp = (Cout + 63) // 64 * 64
    X = torch.randn(b, n, K, generator=g).cuda()
    Wt = torch.zeros(K, Coutp)
    Wt[:, :Cout] = torch.randn(K, Cout, generator=g) / K ** 0.5
    Wt = Wt.cuda()
    bias = torch.zeros(Coutp)
    bias[:Cout] = torch.randn(Cout, generator=g)
    bias = bias.cuda()
    cb = torch.randn(b, Coutp, generator=g).cuda() if use_cb else None
    out = torch.full((b, n, Cout), 7.0, device="cuda") if flags & ops.F_STORE else None
    Y, stats, colext = ops.pointwise(X, Wt, Cout, bias=bias, cbias=cb, out=out, flags=flags, G=G, split=True)
    torch.cuda.synchronize()
    ce = None
    if colext is not None:
        nblk = (n + 127) // 128
        ce = colext.view(torch.float32)[: b * nblk * Coutp * 2]
    ref = None
    if Y is not None:      # fp64 check of the values themselves (both builds must also be RIGHT)
        ref = (X.double() @ Wt.double()[:, :Cout] + bias.double()[:Cout] + (cb.double()[:, None, :Cout] if cb is not None else 0))
        if flags & ops.F_RELU:
            ref = ref.clamp_min(0)
        err = float((Y.double() - ref).abs().max())
    print(f"case B={b} N={n} K={K} Cout={Cout} flags={flags}: Y {dig(Y)} stats {dig(stats)} colext {dig(ce)}"
          + (f" maxerr {err:.2e}" if ref is not None else ""))

LAYERS = [("mlp1 256->1024 stats+colext", 256, 1024, ops.F_STATS | ops.F_COLEXT, 8, 1),
          ("conv1 256->512", 256, 512, ops.F_STORE | ops.F_STATS, 8, 1),
          ("conv2 512->256", 512, 256, ops.F_STORE | ops.F_STATS, 4, 1),
          ("prim1/seg1/asis 256->256", 256, 256, ops.F_STORE | ops.F_STATS, 4, 3),
          ("edge0/seg2 256->128", 256, 128, ops.F_STORE | ops.F_STATS, 4, 2),
          ("penc 32->256 relu", 32, 256, ops.F_STORE | ops.F_RELU, 0, 1)]


def t_ms(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


tot = 0.0
for name, K, Cout, flags, G, mult in LAYERS:
    X = torch.randn(B, N, K, generator=g).cuda()
    Wt = (torch.randn(K, Cout, generator=g) / K ** 0.5).cuda()
    bias = torch.randn(Cout, generator=g).cuda()
    out = torch.empty(B, N, Cout, device="cuda") if flags & ops.F_STORE else None
    t = t_ms(lambda: ops.pointwise(X, Wt, Cout, bias=bias, out=out, flags=flags, G=G, split=True))
    tot += mult * t
    print(f"time {name}: {t:.3f} ms x{mult}  ({2.0 * B * N * K * Cout / t / 1e9:.0f} TF/s logical)")
print(f"time sum of one forward's wide layers: {tot:.3f} ms")
