"""CPU emulation of ms_iterate_f16.hip's index arithmetic (stage image layout, operand slots, accumulator rows) under the
documented v_mfma_f32_32x32x16_f16 lane layout:  A[m = lane & 31][k = 8 (lane >> 5) + i], B[k][n = lane & 31],
D[row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)][col = lane & 31].  One wave, one iteration, compared with a direct
evaluation. Checks the kernel's bookkeeping, not the hardware (tests/test_gpu_mean_shift.py does that)."""
import numpy as np

KT, XROW = 64, 272
TROW = 2 * KT + 16
XPLANE, TPLANE = KT * XROW, 128 * TROW
OFF_XH, OFF_XL, OFF_TH, OFF_TL = 0, XPLANE, 2 * XPLANE, 2 * XPLANE + TPLANE
STAGE = 2 * XPLANE + 2 * TPLANE
SX = 2048.0

def slot_pos(m): return (m >> 4) * 16 + ((m >> 2) & 1) * 8 + ((m >> 3) & 1) * 4 + (m & 3)
def row(r, hi): return (r & 3) + 8 * (r >> 2) + 4 * hi

def split(v):
    h = np.float16(v); l = np.float16(np.float32(v) - np.float32(h)); return h, l

def build_stage(X, stage):
    img = np.zeros(STAGE // 2, np.float16)          # index in halves
    N = X.shape[0]
    for kk in range(KT):
        key = stage * KT + kk
        for d in range(128):
            v = np.float32(X[key, d] * SX) if key < N else np.float32(0)
            h, l = split(v)
            c, m, sub, km = d >> 5, d & 31, kk >> 5, kk & 31
            xo = (kk * XROW) // 2 + c * 32 + slot_pos(m)
            to = (d * TROW) // 2 + sub * 32 + slot_pos(km)
            img[OFF_XH // 2 + xo] = h; img[OFF_XL // 2 + xo] = l
            img[OFF_TH // 2 + to] = h; img[OFF_TL // 2 + to] = l
    return img

def mfma(A, B, C):
    """A, B: [64 lanes][8]; C: [64][16] -> D"""
    Am = np.zeros((32, 16)); Bm = np.zeros((16, 32))
    for lane in range(64):
        for i in range(8):
            Am[lane & 31, 8 * (lane >> 5) + i] = A[lane, i]
            Bm[8 * (lane >> 5) + i, lane & 31] = B[lane, i]
    Dm = Am @ Bm
    D = C.copy()
    for lane in range(64):
        for r in range(16):
            D[lane, r] += Dm[row(r, lane >> 5), lane & 31]
    return D

def rd8(img, byte_off): return img[byte_off // 2: byte_off // 2 + 8].astype(np.float64)

def main():
    rng = np.random.default_rng(0)
    N = 70
    X = rng.normal(size=(N, 128)); X /= np.linalg.norm(X, axis=1, keepdims=True); X = X.astype(np.float32)
    b = 0.9
    nst = (N + KT - 1) // KT
    imgs = [build_stage(X, s) for s in range(nst)]
    # Q planes of the 32 query rows 0..31 (one wave)
    qh = np.zeros((8, 64, 8)); ql = np.zeros((8, 64, 8))
    for lane in range(64):
        li, hi = lane & 31, lane >> 5
        for c in range(4):
            for j in range(2):
                for i in range(8):
                    d = 32 * c + 8 * (2 * j + (i >> 2)) + 4 * hi + (i & 3)
                    h, l = split(np.float32(X[li, d] * SX))
                    qh[2 * c + j, lane, i], ql[2 * c + j, lane, i] = h, l
    o = [np.zeros((64, 16)) for _ in range(4)]
    rsum = np.zeros(64)
    K1 = 1.4426950408889634 / (b * b) / 4194304.0; K0 = 14.0 - 1.4426950408889634 / (b * b)
    for st in range(nst):
        img = imgs[st]
        for sub in range(2):
            key0 = st * KT + sub * 32
            if key0 >= N: continue
            s = np.zeros((64, 16))
            for ks in range(8):
                xh = np.stack([rd8(img, OFF_XH + sub * 32 * XROW + (l & 31) * XROW + (l >> 5) * 16 + ks * 32) for l in range(64)])
                xl = np.stack([rd8(img, OFF_XL + sub * 32 * XROW + (l & 31) * XROW + (l >> 5) * 16 + ks * 32) for l in range(64)])
                s = mfma(xl, qh[ks], s); s = mfma(xh, ql[ks], s); s = mfma(xh, qh[ks], s)
            p = np.exp2(np.maximum(s * K1 + K0, 14 - 75 * 1.4426950408889634))
            for lane in range(64):
                for r in range(16):
                    if key0 + row(r, lane >> 5) >= N: p[lane, r] = 0
            rsum += p.sum(1)
            ph = np.float16(p).astype(np.float64); pl = np.float16(p - ph).astype(np.float64)
            for c in range(4):
                for j in range(2):
                    th = np.stack([rd8(img, OFF_TH + (l & 31) * TROW + (l >> 5) * 16 + sub * 64 + c * 32 * TROW + j * 32) for l in range(64)])
                    tl = np.stack([rd8(img, OFF_TL + (l & 31) * TROW + (l >> 5) * 16 + sub * 64 + c * 32 * TROW + j * 32) for l in range(64)])
                    o[c] = mfma(tl, ph[:, 8 * j:8 * j + 8], o[c]); o[c] = mfma(th, pl[:, 8 * j:8 * j + 8], o[c])
                    o[c] = mfma(th, ph[:, 8 * j:8 * j + 8], o[c])
    rs = rsum[:32] + rsum[32:]
    newq = np.zeros((32, 128))
    for lane in range(64):
        li, hi = lane & 31, lane >> 5
        for c in range(4):
            for r in range(16):
                newq[li, 32 * c + row(r, hi)] = o[c][lane, r] / 2048.0 / rs[li]
    newq /= np.linalg.norm(newq, axis=1, keepdims=True)
    # direct
    Xd = X.astype(np.float64); Q = Xd[:32]
    P = np.exp(np.clip(-(2 - 2 * Q @ Xd.T) / b / b / 2, -75, 75))
    ref = (P @ Xd) / P.sum(1, keepdims=True); ref /= np.linalg.norm(ref, axis=1, keepdims=True)
    err = np.abs(newq - ref).max()
    print("max abs err vs direct fp64:", err)
    assert err < 2e-7
    print("layout OK")

if __name__ == "__main__":
    main()
