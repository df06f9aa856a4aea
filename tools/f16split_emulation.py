"""CPU emulation of the fp16x2-split mean-shift arithmetic (tools only; imports the oracle as the checker).
x = h + l with h = fp16(x * 2^SX), l = fp16(x * 2^SX - h); products h.h + h.l + l.h accumulated in fp32/fp64;
the l.l term (<= 2^-24 relative per product) is dropped.  Compared against the reference snapshots of f_ms.npz."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
F32 = np.float32

def split16(a, scale):
    s = (a.astype(F32) * F32(scale)).astype(F32)
    h = s.astype(np.float16)
    l = (s - h.astype(F32)).astype(F32).astype(np.float16)
    return h.astype(F32), l.astype(F32)

def iterate(X, b, iters, mode, snaps):
    X = X.astype(F32); Q = X.copy()
    SX, SP = 2.0 ** 11, 2.0 ** 14
    xh, xl = split16(X, SX)
    c = F32(-0.5) / (F32(b) * F32(b))
    out = {}
    for it in range(iters):
        if mode == "f32":
            s = (Q @ X.T).astype(F32)
        elif mode == "f64":
            s = (Q.astype(np.float64) @ X.T.astype(np.float64))
        else:
            qh, ql = split16(Q, SX)
            acc = (ql.astype(np.float64) @ xh.T) + (qh.astype(np.float64) @ xl.T)
            acc = acc.astype(F32) if mode.startswith("f16x2") else acc
            acc = acc + qh.astype(np.float64) @ xh.T
            s = (acc.astype(F32) * F32(2.0 ** -22)).astype(F32)
        if mode == "f64":
            a = np.clip((2.0 - 2.0 * s) * (-0.5 / (float(b) ** 2)), -75, 75); p = np.exp(a)
            o = p @ X.astype(np.float64); rs = p.sum(1, keepdims=True)
        else:
            a = np.clip(((F32(2) - F32(2) * s) * c).astype(F32), F32(-75), F32(75))
            p = np.exp(a.astype(np.float64)).astype(F32)
            if mode == "f32":
                o = (p @ X).astype(F32); rs = p.sum(1, keepdims=True, dtype=F32)
            else:
                ph, pl = split16(p, SP)
                if mode.endswith("ph"):
                    # 5-MFMA form: the weights enter both sums as their fp16 heads only (numerator and denominator
                    # consistently), X keeps both digits
                    o = (ph.astype(np.float64) @ xl) + (ph.astype(np.float64) @ xh)
                    rs = (ph.astype(np.float64).sum(1, keepdims=True) * 2.0 ** -14).astype(F32)
                else:
                    o = (pl.astype(np.float64) @ xh) + (ph.astype(np.float64) @ xl) + (ph.astype(np.float64) @ xh)
                    rs = ((ph.astype(np.float64) + pl).sum(1, keepdims=True) * 2.0 ** -14).astype(F32)
                o = (o * 2.0 ** -25).astype(F32)
        if mode == "f64":
            nq = o / rs
            Q = (nq / np.linalg.norm(nq, axis=1, keepdims=True)).astype(F32)
        else:
            m = (o * (F32(1) / rs) - Q).astype(F32); nq = (Q + m).astype(F32)
            Q = (nq / np.sqrt((nq * nq).sum(1, keepdims=True, dtype=F32))).astype(F32)
        if it + 1 in snaps: out[it + 1] = Q.copy()
    return out

if __name__ == "__main__":
    g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", sys.argv[1] if len(sys.argv) > 1 else "f_ms.npz"))
    X = g["X"]; b = max(float(g["bw_q05_ns2000"]) if "bw_q05_ns2000" in g.files else float(g["bw"]), 0.003)
    ref = {1: g["newX_it1"], 5: g["newX_it5"], 50: g["newX_it50"]}
    res = {m: iterate(X, b, 50, m, ref) for m in ("f64", "f32", "f16x2", "f16x2ph")}
    for it in (1, 5, 50):
        n = ref[it].shape[0]
        print(it, {m: float(np.abs(res[m][it][:n] - ref[it]).max()) for m in res},
              "f16x2 vs f64", float(np.abs(res["f16x2"][it] - res["f64"][it]).max()),
              "f16x2ph vs f64", float(np.abs(res["f16x2ph"][it] - res["f64"][it]).max()),
              "f32 vs f64", float(np.abs(res["f32"][it] - res["f64"][it]).max()))
