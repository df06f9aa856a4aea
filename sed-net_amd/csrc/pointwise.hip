// Point-wise (kernel-1) convolutions of the SED-Net encoder tail and heads as fp32-MFMA GEMMs with fused
// bias / per-cloud bias / ReLU / GroupNorm statistics / column extrema, plus the small kernels around them.
//
// Replaces /root/reference/src/SEDNet.py:94-96 (mlp1 -> GroupNorm(8) -> ReLU -> max over N) and :300-329
// (conv1, conv2, type / edge / embedding heads: Conv1d -> GroupNorm -> ReLU chains).
//
//   Y[b, p, o] = sum_c X[b, p, c] Wt[c, o] + bias[o] + cbias[b, o]
// X is point-major [B,N,ldx]; a workgroup owns 128 points x BN channels (BN = 128 or 64), 4 waves x 32
// points, K streamed in chunks of 32 through a double-buffered LDS ring (A tile row stride 36 floats ->
// conflict-free ds_read_b128; B tile rows read as consecutive floats). v_mfma_f32_32x32x2_f32 keeps fp32
// exactness (fma chains), which the parity bar needs.
// GroupNorm over (C/G, N) needs global statistics: the epilogue emits deterministic per-workgroup fp64
// partial sums per 32-channel tile; gn_finalize reduces them in fixed order; gn_apply applies
// scale/shift + activation (+ optional `out = scale * act(.) + addend` for the feature-fusion adds at
// SEDNet.py:322,326). For mlp1 only max_N relu(GN(y)) is needed, and relu(affine) is monotone per channel,
// so the epilogue keeps per-channel max/min over the tile instead of writing the [N,1024] tensor.
#include "common.h"
#include <type_traits>
#include "split16.h"

namespace {

enum { F_RELU = 1, F_STORE = 2, F_STATS = 4, F_COLEXT = 8 };

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// BF16 (training, BASELINE configs[4]): both operands are rounded to bf16 (round to nearest even) while they are staged
// into LDS and the products run on v_mfma_f32_32x32x16_bf16 with fp32 accumulation; bias, statistics (fp64 partials) and
// the stored output stay fp32. LDS then holds A as [128 points][32 k] bf16 (row stride 80 B) and B TRANSPOSED as
// [BN channels][32 k] bf16 (row stride 80 B), so that every MFMA operand is one ds_read_b128.
template <int TN, bool BF16>
__global__ __launch_bounds__(256, 2) void pointwise_kernel(const float* __restrict__ X, int ldx, int K,
                                                           const float* __restrict__ Wt, int ldw,
                                                           const float* __restrict__ bias,
                                                           const float* __restrict__ cbias, float* __restrict__ Y,
                                                           int ldy, int Cout, double* __restrict__ part,
                                                           float* __restrict__ colext, int N, int flags) {
    constexpr int BN = 32 * TN;
    constexpr int LDA = 36;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                       // [2][128][LDA]
    float* Bs = smem + 2 * 128 * LDA;       // [2][32][BN]
    double* red = (double*)(Bs + 2 * 32 * BN);   // [4][TN][2]
    float* ext = (float*)(red + 4 * TN * 2);      // [4][BN][2]

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, hi = lane >> 5;
    const int cloud = blockIdx.y, o0 = blockIdx.z * BN;
    const int p0 = blockIdx.x * 128;
    const float* Xc = X + (size_t)cloud * N * ldx;
    const int nchunk = K / 32;

    f32x4 sa[4], sb[TN];
    auto stage_load = [&](int ch) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = tid + 256 * u, row = i >> 3, c4 = i & 7;
            int p = p0 + row;
            if (p >= N) p = N - 1;
            sa[u] = *(const f32x4*)(Xc + (size_t)p * ldx + ch * 32 + 4 * c4);
        }
#pragma unroll
        for (int u = 0; u < TN; ++u) {
            const int i = tid + 256 * u, row = i / (BN / 4), c4 = i % (BN / 4);
            sb[u] = *(const f32x4*)(Wt + (size_t)(ch * 32 + row) * ldw + o0 + 4 * c4);
        }
    };
    // bf16 images inside the same LDS regions: A [buf][128][40 halves], B^T [buf][BN][40 halves]
    __bf16* Ab = (__bf16*)As;
    __bf16* Bb = (__bf16*)Bs;
    auto stage_store = [&](int buf) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = tid + 256 * u, row = i >> 3, c4 = i & 7;
            if (BF16) {
                bf16x4 h;
#pragma unroll
                for (int e = 0; e < 4; ++e) h[e] = (__bf16)sa[u][e];
                *(bf16x4*)(Ab + (buf * 128 + row) * 40 + 4 * c4) = h;
            } else {
                *(f32x4*)(As + (buf * 128 + row) * LDA + 4 * c4) = sa[u];
            }
        }
#pragma unroll
        for (int u = 0; u < TN; ++u) {
            const int i = tid + 256 * u, row = i / (BN / 4), c4 = i % (BN / 4);
            if (BF16) {
#pragma unroll
                for (int e = 0; e < 4; ++e) Bb[(buf * BN + 4 * c4 + e) * 40 + row] = (__bf16)sb[u][e];   // transpose
            } else {
                *(f32x4*)(Bs + (buf * 32 + row) * BN + 4 * c4) = sb[u];
            }
        }
    };

    f32x16 acc[TN];
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    stage_load(0);
    stage_store(0);
    __syncthreads();
    int cur = 0;
    for (int ch = 0; ch < nchunk; ++ch) {
        if (ch + 1 < nchunk) stage_load(ch + 1);
        if (BF16) {
            const __bf16* a = Ab + (cur * 128 + wave * 32 + li) * 40 + hi * 8;
            const __bf16* b = Bb + (cur * BN + li) * 40 + hi * 8;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 av = *(const bf16x8*)(a + 16 * s2);
#pragma unroll
                for (int t = 0; t < TN; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, *(const bf16x8*)(b + 32 * t * 40 + 16 * s2), acc[t],
                                                                     0, 0, 0);
            }
        } else {
            const float* a = As + (cur * 128 + wave * 32 + li) * LDA + hi * 16;
            const float* b = Bs + (cur * 32 + hi * 16) * BN + li;
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const f32x4 av = *(const f32x4*)(a + 4 * s4);
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int t = 0; t < TN; ++t) acc[t] = mfma32(av[c], b[(4 * s4 + c) * BN + 32 * t], acc[t]);
            }
        }
        if (ch + 1 < nchunk) stage_store(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue
    const int pw = p0 + wave * 32;
    unsigned vmask = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) vmask |= (pw + mfma_row(r, hi) < N ? 1u : 0u) << r;
#pragma unroll
    for (int t = 0; t < TN; ++t) {
        const int o = o0 + 32 * t + li;
        float add = bias ? bias[o] : 0.f;
        if (cbias) add += cbias[(size_t)cloud * ldw + o];
        float ps = 0.f, pq = 0.f, mx = -3.0e38f, mn = 3.0e38f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = acc[t][r] + add;
            if (flags & F_RELU) v = fmaxf(v, 0.f);
            const bool ok = (vmask >> r) & 1u;
            if ((flags & F_STORE) && ok && o < Cout)
                Y[((size_t)cloud * N + pw + mfma_row(r, hi)) * ldy + o] = v;
            if (ok) { ps += v; pq = fmaf(v, v, pq); mx = fmaxf(mx, v); mn = fminf(mn, v); }
        }
        if (flags & F_STATS) {
            double d1 = (double)ps, d2 = (double)pq;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { d1 += __shfl_xor(d1, off, 64); d2 += __shfl_xor(d2, off, 64); }
            if (lane == 0) { red[(wave * TN + t) * 2] = d1; red[(wave * TN + t) * 2 + 1] = d2; }
        }
        if (flags & F_COLEXT) {
            mx = fmaxf(mx, xor32(mx));
            mn = fminf(mn, xor32(mn));
            if (hi == 0) { ext[(wave * BN + 32 * t + li) * 2] = mx; ext[(wave * BN + 32 * t + li) * 2 + 1] = mn; }
        }
    }
    if (flags & (F_STATS | F_COLEXT)) __syncthreads();
    if ((flags & F_STATS) && tid < TN * 2) {
        const int t = tid >> 1, which = tid & 1;
        double s = 0.0;
        for (int w = 0; w < 4; ++w) s += red[(w * TN + t) * 2 + which];
        const int ntile = gridDim.z * TN;
        part[(((size_t)cloud * gridDim.x + blockIdx.x) * ntile + blockIdx.z * TN + t) * 2 + which] = s;
    }
    if ((flags & F_COLEXT) && tid < BN) {
        float mx = -3.0e38f, mn = 3.0e38f;
        for (int w = 0; w < 4; ++w) { mx = fmaxf(mx, ext[(w * BN + tid) * 2]); mn = fminf(mn, ext[(w * BN + tid) * 2 + 1]); }
        float* dst = colext + (((size_t)cloud * gridDim.x + blockIdx.x) * ldw + o0 + tid) * 2;
        dst[0] = mx;
        dst[1] = mn;
    }
}

// ------------------------------------------------------------------------------------------------------------
// Inference GEMMs on the bf16 matrix pipe with fp32-equivalent results (round 2): three-way bf16 splits.
//     v = b1 + b2 + b3 + r,   b1 = bf16(v), b2 = bf16(v - b1), b3 = bf16(v - b1 - b2),   |r| <= 2^-27 |v|
// (bf16 keeps fp32's exponent, so no scaling and no bound on the activations is needed -- the split-fp16 scheme of the
// mean-shift kernels needs a per-row scale, and measured here its max / scale / rescale work made the kernel VALU-bound:
// 16 VALU instructions per MFMA, no faster than fp32). A product x w is evaluated as
//     x1 w3 + x3 w1 + x2 w2 + x1 w2 + x2 w1 + x1 w1
// -- six bf16 MFMAs per 16 k (products of two 8-bit significands are exact in the fp32 accumulator); dropped: x2 w3 + x3 w2
// + x3 w3 + the representation error, <= 2^-25 |x w|: below ONE fp32 rounding, where the fp32 fma chain rounds K times.
// Matrix time: 6 x 32 cycles per 16 k against 8 x 64 cycles for v_mfma_f32_32x32x2_f32 = 0.375 x.
// The weights arrive pre-split (three planes [Coutp][K] bf16, k-contiguous) and are staged through a double-buffered LDS
// ring (row stride 80 B: one conflict-free ds_read_b128 per operand). A wave owns 32 points x BN channels, so an
// activation is needed by ONE wave only: every lane reads the 8 consecutive k of its MFMA operand straight from global
// memory (two 16-byte loads per 16 k, prefetched one chunk ahead) and splits them in registers -- no LDS round trip.
// Workgroups that share an X tile (same points, different channel block) get consecutive slots on the SAME XCD, so that X
// is read from HBM once and from that XCD's L2 afterwards. Epilogue identical to pointwise_kernel.
__device__ __forceinline__ void split3(const f32x4& lo, const f32x4& hi4, bf16x8& b1, bf16x8& b2, bf16x8& b3) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float v = e < 4 ? lo[e & 3] : hi4[e & 3];
        const __bf16 p1 = (__bf16)v;
        const float r1 = v - (float)p1;
        const __bf16 p2 = (__bf16)r1;
        const float r2 = r1 - (float)p2;
        b1[e] = p1;
        b2[e] = p2;
        b3[e] = (__bf16)r2;
    }
}

__device__ __forceinline__ void split2(const f32x4& lo, const f32x4& hi4, float scale, h16x8& h, h16x8& l) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { h16 a, b; split_pair(e < 4 ? lo[e & 3] : hi4[e & 3], scale, a, b); h[e] = a; l[e] = b; }
}

// GNIN (round 6): X is a layer's PRE-normalisation output and the kernel applies that layer's GroupNorm + activation to every value as
// it arrives -- x = act(fmaf(X, a_k, b_k)), a_k = rstd_g gamma_k, b_k = fmaf(-a_k, mean_g, beta_k): gn_apply_kernel's arithmetic, the
// same bits -- so the normalised tensor is never written or read (a_k, b_k of the cloud sit in LDS: 4 KiB for K <= 512).
struct GnIn { const float *stats, *gamma, *beta; int G, act; };
__device__ __forceinline__ void gn_in_table(const GnIn& gi, int cloud, int K, float* coef /* [2][K] */, int tid) {
    const int cpg = K / gi.G;
    for (int k = tid; k < K; k += 256) {
        const int g = k / cpg;
        const float mean = gi.stats[((size_t)cloud * gi.G + g) * 2], rstd = gi.stats[((size_t)cloud * gi.G + g) * 2 + 1];
        const float a = rstd * gi.gamma[k];
        coef[k] = a;
        coef[K + k] = fmaf(-a, mean, gi.beta[k]);
    }
}
__device__ __forceinline__ f32x4 gn_in_apply(const f32x4& x, const f32x4& a, const f32x4& b, int act) {
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float v = fmaf(x[e], a[e], b[e]);
        if (act == 1) v = fmaxf(v, 0.f);
        o[e] = v;
    }
    return o;
}

// H16: the two-plane split-fp16 form (split16.h) for inputs whose per-row magnitude bound the producer has already written
// (`rowmax`, sed_gn_apply_f32): 3 fp16 MFMAs per 16 k instead of 6 bf16 ones, 2 weight planes instead of 3, a 5-instruction
// split per activation instead of 9. The rows are scaled by 2^e (row bound in [2^13, 2^14)), the weights per output channel
// (winv = 2^-e behind the planes); the epilogue unscales by the exact product of the two powers of two.
template <int TN, bool H16, bool GNIN = false>
__global__ __launch_bounds__(256, 2) void pointwise_split_kernel(const float* __restrict__ X, int ldx, int K,
                                                                 const __bf16* __restrict__ Wp /* [3 | 2][Coutp][K] */,
                                                                 int Coutp, const float* __restrict__ bias,
                                                                 const float* __restrict__ cbias, float* __restrict__ Y,
                                                                 int ldy, int Cout, double* __restrict__ part,
                                                                 float* __restrict__ colext, int N, int nblk, int flags,
                                                                 const unsigned* __restrict__ rowmax, GnIn gi = GnIn{}) {
    constexpr int BN = 32 * TN;
    constexpr int LDH = 40;                                  // 16-bit values per LDS row (32 k + 8 pad = 80 B)
    constexpr int NP = H16 ? 2 : 3;                          // weight planes
    constexpr int NB = NP * BN * 4 / 256;                    // 16-byte pieces of one weight stage per thread
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __bf16* Bs = (__bf16*)smem;                              // [2][NP][BN][LDH]
    double* red = (double*)(Bs + 2 * NP * BN * LDH);         // [4][TN][2]
    float* ext = (float*)(red + 4 * TN * 2);                 // [4][BN][2]
    float* ainv_s = ext + 4 * BN * 2;                        // [4][32] (H16): 2^-e of the waves' rows
    float* coef = ainv_s + 4 * 32;                           // (GNIN) [2][K]: a_k | b_k of this cloud (pointwise_wide_kernel)

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, hi = lane >> 5;
    // slot -> (point tile, channel block): slots L, L + 8, L + 16, ... run on XCD L % 8; consecutive slots of one XCD walk
    // the channel blocks of one point tile first
    const int nz = Coutp / BN;
    const int L = blockIdx.x, xcd = L & 7, j = L >> 3;
    const int zb = j % nz, pt = (j / nz) * 8 + xcd;
    if (pt >= nblk) return;
    const int cloud = blockIdx.y, o0 = zb * BN;
    const int p0 = pt * 128;
    const float* Xc = X + (size_t)cloud * N * ldx;
    const int nchunk = K / 32;
    const size_t plane = (size_t)Coutp * K;

    if (GNIN) gn_in_table(gi, cloud, K, coef, tid);          // visible after the barrier in front of the chunk loop
    int prow = p0 + wave * 32 + li;
    if (prow >= N) prow = N - 1;
    const float* xrow = Xc + (size_t)prow * ldx + 8 * hi;
    // activations are prefetched TWO chunks ahead (the X stream comes from HBM: one chunk of matrix work, ~0.7 us, does not
    // cover its latency); the weight planes (L2-resident) one chunk ahead
    f32x4 xa0[4], xa1[4];                                    // [k-step][first / second float4]
    auto load_a = [&](int ch, f32x4* xa) {
#pragma unroll
        for (int u = 0; u < 4; ++u) xa[u] = *(const f32x4*)(xrow + ch * 32 + 16 * (u >> 1) + 4 * (u & 1));
    };
    u32x4 sb[NB];
    auto load_b = [&](int ch) {
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int i = tid + 256 * u, pl = i / (BN * 4), row = (i / 4) % BN, seg = i & 3;
            sb[u] = *(const u32x4*)(Wp + pl * plane + (size_t)(o0 + row) * K + ch * 32 + 8 * seg);
        }
    };
    auto store_b = [&](int buf) {
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int i = tid + 256 * u, pl = i / (BN * 4), row = (i / 4) % BN, seg = i & 3;
            *(u32x4*)(Bs + ((buf * NP + pl) * BN + row) * LDH + 8 * seg) = sb[u];
        }
    };
    float ascale = 1.f;
    if (H16) {
        ascale = split_row_scale(__uint_as_float(rowmax[(size_t)cloud * N + prow]));
        if (hi == 0) ainv_s[wave * 32 + li] = 1.0f / ascale;      // visible after the first barrier below
    }

    f32x16 acc[TN];
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    auto step = [&](int ch, f32x4* xa, int cur) {
        bf16x8 a1[2], a2[2], a3[2];
        h16x8 ah[2], al[2];
        if (GNIN) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = ch * 32 + 16 * (u >> 1) + 8 * hi + 4 * (u & 1);
                xa[u] = gn_in_apply(xa[u], *(const f32x4*)(coef + k), *(const f32x4*)(coef + K + k), gi.act);
            }
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            if (H16) split2(xa[2 * s2], xa[2 * s2 + 1], ascale, ah[s2], al[s2]);
            else split3(xa[2 * s2], xa[2 * s2 + 1], a1[s2], a2[s2], a3[s2]);
        }
        if (ch + 2 < nchunk) load_a(ch + 2, xa);
        if (ch + 1 < nchunk) load_b(ch + 1);
        const __bf16* b1p = Bs + ((cur * NP + 0) * BN + li) * LDH + hi * 8;
        const __bf16* b2p = b1p + BN * LDH;
        const __bf16* b3p = b2p + BN * LDH;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
            for (int t = 0; t < TN; ++t) {
                if (H16) {
                    const h16x8 wh = *(const h16x8*)(b1p + 32 * t * LDH + 16 * s2);
                    const h16x8 wl = *(const h16x8*)(b2p + 32 * t * LDH + 16 * s2);
                    acc[t] = mfma16(al[s2], wh, acc[t]);
                    acc[t] = mfma16(ah[s2], wl, acc[t]);
                    acc[t] = mfma16(ah[s2], wh, acc[t]);
                } else {
                    const bf16x8 w1 = *(const bf16x8*)(b1p + 32 * t * LDH + 16 * s2);
                    const bf16x8 w2 = *(const bf16x8*)(b2p + 32 * t * LDH + 16 * s2);
                    const bf16x8 w3 = *(const bf16x8*)(b3p + 32 * t * LDH + 16 * s2);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[s2], w3, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[s2], w1, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[s2], w2, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[s2], w2, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[s2], w1, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[s2], w1, acc[t], 0, 0, 0);
                }
            }
        }
        if (ch + 1 < nchunk) store_b(cur ^ 1);
        __syncthreads();
    };

    load_a(0, xa0);
    if (nchunk > 1) load_a(1, xa1);
    load_b(0);
    store_b(0);
    __syncthreads();
    for (int ch = 0; ch < nchunk; ch += 2) {
        step(ch, xa0, 0);
        if (ch + 1 < nchunk) step(ch + 1, xa1, 1);
    }

    // ---- epilogue (pointwise_kernel's arithmetic and summation order). The element loop is instantiated for every flag
    // combination and for "all 32 points of the wave's tile exist" (every tile but a cloud's last): no
    // per-element flag tests, masks or 64-bit address arithmetic (the generic form was ~1900 instructions per wave, a third of
    // the kernel at K = 256); a point row's address is a uniform base + one 32-bit lane offset.
    const int pw = p0 + wave * 32;
    unsigned vmask = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) vmask |= (pw + mfma_row(r, hi) < N ? 1u : 0u) << r;
    const bool full = pw + 32 <= N;
    f32x4 ai4[4];                                            // (H16) 2^-e of this lane's 16 accumulator rows
    const float* winv = (const float*)(Wp + (size_t)NP * plane);
    if (H16) {
#pragma unroll
        for (int g = 0; g < 4; ++g) ai4[g] = *(const f32x4*)&ainv_s[wave * 32 + 8 * g + 4 * hi];
    }
    auto tiles = [&](auto relu_c, auto store_c, auto stats_c, auto ext_c, auto full_c) {
        constexpr bool RELU = decltype(relu_c)::value, STORE = decltype(store_c)::value, STATS = decltype(stats_c)::value,
                       EXT = decltype(ext_c)::value, FULL = decltype(full_c)::value;
#pragma unroll
        for (int t = 0; t < TN; ++t) {
            const int o = o0 + 32 * t + li;
            float add = bias ? bias[o] : 0.f;
            if (cbias) add += cbias[(size_t)cloud * Coutp + o];
            float ps = 0.f, pq = 0.f, mx = -3.0e38f, mn = 3.0e38f;
            const unsigned loff = (unsigned)(4 * hi * ldy + o);          // lane part of the address: row 4 hi, channel o
            const bool och = o < Cout;
            const float wi = H16 ? winv[o] : 1.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = H16 ? __fadd_rn(acc[t][r] * (ai4[r >> 2][r & 3] * wi), add) : acc[t][r] + add;
                if (RELU) v = sed_vmax(v, 0.f);
                const bool ok = FULL || ((vmask >> r) & 1u);
                if (STORE && ok && och) {
                    float* rowbase = Y + ((size_t)cloud * N + pw + ((r & 3) + 8 * (r >> 2))) * ldy;     // uniform
                    rowbase[loff] = v;
                }
                if (ok) {
                    if (STATS) { ps += v; pq = fmaf(v, v, pq); }
                    if (EXT) { mx = sed_vmax(mx, v); mn = sed_vmin(mn, v); }
                }
            }
            if (STATS) {
                double d1 = (double)ps, d2 = (double)pq;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) { d1 += __shfl_xor(d1, off, 64); d2 += __shfl_xor(d2, off, 64); }
                if (lane == 0) { red[(wave * TN + t) * 2] = d1; red[(wave * TN + t) * 2 + 1] = d2; }
            }
            if (EXT) {
                mx = fmaxf(mx, xor32(mx));
                mn = fminf(mn, xor32(mn));
                if (hi == 0) { ext[(wave * BN + 32 * t + li) * 2] = mx; ext[(wave * BN + 32 * t + li) * 2 + 1] = mn; }
            }
        }
    };
    using TT = std::true_type;
    using FF = std::false_type;
    auto with_full = [&](auto relu_c, auto store_c, auto stats_c, auto ext_c) {
        if (full) tiles(relu_c, store_c, stats_c, ext_c, TT{});
        else tiles(relu_c, store_c, stats_c, ext_c, FF{});
    };
    auto d3 = [&](auto a, auto b, auto c) { if (flags & F_COLEXT) with_full(a, b, c, TT{}); else with_full(a, b, c, FF{}); };
    auto d2 = [&](auto a, auto b) { if (flags & F_STATS) d3(a, b, TT{}); else d3(a, b, FF{}); };
    auto d1 = [&](auto a) { if (flags & F_STORE) d2(a, TT{}); else d2(a, FF{}); };
    if (flags & F_RELU) d1(TT{}); else d1(FF{});
    if (flags & (F_STATS | F_COLEXT)) __syncthreads();
    if ((flags & F_STATS) && tid < TN * 2) {
        const int t = tid >> 1, which = tid & 1;
        double s = 0.0;
        for (int w = 0; w < 4; ++w) s += red[(w * TN + t) * 2 + which];
        const int ntile = nz * TN;
        part[(((size_t)cloud * nblk + pt) * ntile + zb * TN + t) * 2 + which] = s;
    }
    if ((flags & F_COLEXT) && tid < BN) {
        float mx = -3.0e38f, mn = 3.0e38f;
        for (int w = 0; w < 4; ++w) { mx = fmaxf(mx, ext[(w * BN + tid) * 2]); mn = fminf(mn, ext[(w * BN + tid) * 2 + 1]); }
        float* dst = colext + (((size_t)cloud * nblk + pt) * Coutp + o0 + tid) * 2;
        dst[0] = mx;
        dst[1] = mn;
    }
}

// ------------------------------------------------------------------------------------------------------------
// Round 6: the WIDE form of pointwise_split_kernel<4, false> -- a wave owns 64 points x 128 channels (two 32-point subtiles), a
// workgroup 256 points x 128 channels. Why: in the 32-point form every weight operand read from LDS feeds 6 MFMAs (192 matrix
// cycles for 8 LDS cycles of a ds_read_b128, times 8 waves per CU: the LDS pipe is ~70 % as busy as the matrix pipe would be at
// its peak, plus 1.16 conflict cycles per instruction from the 16-byte staging stores into 80-byte rows) and every 128 points
// re-stage the whole weight slice through registers; MFMA-pipe busy 0.38 (profiles/r05_pmc_kernels.md), and halving the MFMAs
// (the split-fp16 form of round 4) took 18 % off, not 50 %. Here a weight operand feeds 12 MFMAs (the two subtiles), the weight
// slice is staged once per 256 points, and it travels global -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers,
// no LDS store instructions) into rows of 64 B whose four 16-byte chunks sit at slot c ^ ((row >> 2) & 3): the 16 lanes a
// ds_read_b128 serves together read 16 rows at one k offset and land in 16 different bank groups. A lane's DMA source address is
// free, so the swizzle costs nothing on the way in. The copies are inline instructions: one chunk ahead, waited for by the
// explicit vmcnt(0) in front of the chunk's barrier (which is also what the chunk's activations, loaded one chunk ahead, need).
// Same MFMA sequence per accumulator and the same epilogue arithmetic as pointwise_split_kernel => the same bits; GroupNorm
// partials and column extrema stay per 128-POINT block (a workgroup writes two sets: waves 0, 1 and waves 2, 3), each a sum of
// four 32-point wave sums in the old order, so gn_finalize's inputs are bit-identical too.
#ifndef PW_WIDE
#define PW_WIDE 1                  // 0: A/B builds with the 32-point form of rounds 2-5
#endif
#ifndef PW_EXP
#define PW_EXP 0                   // measurement builds only (results wrong): 1 no epilogue, 2 no operand split, 4 activations loaded
#endif                             // once, 8 weights staged once, 16 no MFMA  (tools/pointwise_ab.py, profiles/r06_pointwise_wide.md)
template <int TN, bool GNIN>
__global__ __launch_bounds__(256, 2) void pointwise_wide_kernel(const float* __restrict__ X, int ldx, int K,
                                                                const __bf16* __restrict__ Wp /* [3][Coutp][K] */, int Coutp,
                                                                const float* __restrict__ bias, const float* __restrict__ cbias,
                                                                float* __restrict__ Y, int ldy, int Cout,
                                                                double* __restrict__ part, float* __restrict__ colext, int N,
                                                                int nblk /* 128-point blocks */, int flags, GnIn gi) {
    constexpr int BN = 32 * TN;
    constexpr int PLANE = BN * 64;                           // bytes of one weight plane of a stage: BN rows x 32 k bf16
    constexpr int STG = 3 * PLANE;
    constexpr int PW = STG / 1024 / 4;                       // 1 KiB DMA pieces per wave and stage
    static_assert(STG % 4096 == 0, "stage pieces per wave");
    extern __shared__ __attribute__((aligned(1024))) uint8_t wsm[];
    uint8_t* Bs = wsm;                                       // [2][STG]
    double* red = (double*)(Bs + 2 * STG);                   // [2 halves][4 slots][TN][2]
    float* ext = (float*)(red + 2 * 4 * TN * 2);             // [2 halves][4 slots][BN][2]
    float* coef = ext + 2 * 4 * BN * 2;                      // (GNIN) [2][K]: a_k | b_k of this cloud

    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, li = lane & 31, hi = lane >> 5;
    const int nz = Coutp / BN, nwt = (nblk + 1) >> 1;
    const int L = blockIdx.x, xcd = L & 7, j = L >> 3;       // slot -> (point tile, channel block) as in pointwise_split_kernel
    const int zb = j % nz, pt = (j / nz) * 8 + xcd;
    if (pt >= nwt) return;
    const int cloud = blockIdx.y, o0 = zb * BN;
    const int p0 = pt * 256;
    const float* Xc = X + (size_t)cloud * N * ldx;
    const int nchunk = K / 32;
    const size_t plane = (size_t)Coutp * K;
    if (GNIN) gn_in_table(gi, cloud, K, coef, tid);          // visible after the first chunk's barrier

    const float* xrow[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        int r = p0 + wave * 64 + p * 32 + li;
        if (r >= N) r = N - 1;
        xrow[p] = Xc + (size_t)r * ldx + 8 * hi;
    }
    f32x4 xa[2][4];                                          // [subtile][k-step x first / second float4] of the NEXT chunk
    auto load_a = [&](int ch) {
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int u = 0; u < 4; ++u) xa[p][u] = *(const f32x4*)(xrow[p] + ch * 32 + 16 * (u >> 1) + 4 * (u & 1));
    };
    // stage copy: 16-byte slot q of the stage = (plane, row, slot in row); it holds chunk c = slot ^ ((row >> 2) & 3) of the row
    unsigned voff[PW];
#pragma unroll
    for (int u = 0; u < PW; ++u) {
        const int q = (wave * PW + u) * 64 + lane;
        const int pl = q / (BN * 4), row = (q >> 2) % BN, c = (q & 3) ^ ((row >> 2) & 3);
        voff[u] = (unsigned)((pl * plane + (size_t)row * K + 8 * c) * sizeof(__bf16));
    }
    const __bf16* wbase = Wp + (size_t)o0 * K;
    auto stage_dma = [&](int ch, int buf) {
        const __bf16* src = wbase + ch * 32;
#pragma unroll
        for (int u = 0; u < PW; ++u) {
            const unsigned la = __builtin_amdgcn_readfirstlane(
                (unsigned)(size_t)(__attribute__((address_space(3))) uint8_t*)(Bs + buf * STG + (wave * PW + u) * 1024));
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff[u]), "s"(src), "s"(la) : "memory");
        }
    };
    const int boff = li * 64, sw = (li >> 2) & 3;            // weight operand of (tile t, k-step s2): row 32 t + li, chunk hi + 2 s2
    const int bo0 = boff + ((hi ^ sw) << 4), bo1 = boff + (((hi + 2) ^ sw) << 4);

    f32x16 acc[2][TN];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int t = 0; t < TN; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[p][t][r] = 0.f;

    // One chunk. The activations are compiler-tracked loads, the stage copies are not (inline instructions): a compiler-placed
    // wait for "my loads" in the middle of the chunk would count wrongly and drain the copies just issued. So every value the
    // chunk needs from memory is consumed (split into its three planes) right behind the explicit wait, BEFORE the next chunk's
    // loads are issued -- the compiler's own wait lands where nothing is in flight -- and the sched_barrier keeps that order.
    auto chunk = [&](int ch, int cur) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // B(ch), X(ch) landed; everyone is out of the other buffer
        bf16x8 a1[2][2], a2[2][2], a3[2][2];                   // [k-step][subtile]
        if (GNIN) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = ch * 32 + 16 * (u >> 1) + 8 * hi + 4 * (u & 1);
                const f32x4 ca = *(const f32x4*)(coef + k), cb = *(const f32x4*)(coef + K + k);
#pragma unroll
                for (int p = 0; p < 2; ++p) xa[p][u] = gn_in_apply(xa[p][u], ca, cb, gi.act);
            }
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                if (PW_EXP & 2) {
                    a1[s2][p] = __builtin_bit_cast(bf16x8, f32x4{xa[p][2 * s2][0], xa[p][2 * s2][1], xa[p][2 * s2][2], xa[p][2 * s2][3]});
                    a2[s2][p] = __builtin_bit_cast(bf16x8, f32x4{xa[p][2 * s2 + 1][0], xa[p][2 * s2 + 1][1], xa[p][2 * s2 + 1][2], xa[p][2 * s2 + 1][3]});
                    a3[s2][p] = a1[s2][p];
                } else split3(xa[p][2 * s2], xa[p][2 * s2 + 1], a1[s2][p], a2[s2][p], a3[s2][p]);
            }
        __builtin_amdgcn_sched_barrier(0);
        if (ch + 1 < nchunk) {
            if (!(PW_EXP & 8)) stage_dma(ch + 1, cur ^ 1);
            if (!(PW_EXP & 4)) load_a(ch + 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        const uint8_t* bb = Bs + cur * STG;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const uint8_t* b = bb + (s2 ? bo1 : bo0);
#pragma unroll
            for (int t = 0; t < TN; ++t) {
                const bf16x8 w1 = *(const bf16x8*)(b + t * 2048);
                const bf16x8 w2 = *(const bf16x8*)(b + PLANE + t * 2048);
                const bf16x8 w3 = *(const bf16x8*)(b + 2 * PLANE + t * 2048);
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    if (PW_EXP & 16) {
#if PW_EXP && defined(__HIP_DEVICE_COMPILE__)
                        asm volatile("" ::"v"(w1), "v"(w2), "v"(w3), "v"(a1[s2][p]), "v"(a2[s2][p]), "v"(a3[s2][p]));
#endif
                        continue;
                    }
                    acc[p][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[s2][p], w3, acc[p][t], 0, 0, 0);
                    acc[p][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[s2][p], w1, acc[p][t], 0, 0, 0);
                    acc[p][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[s2][p], w2, acc[p][t], 0, 0, 0);
                    acc[p][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[s2][p], w2, acc[p][t], 0, 0, 0);
                    acc[p][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[s2][p], w1, acc[p][t], 0, 0, 0);
                    acc[p][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[s2][p], w1, acc[p][t], 0, 0, 0);
                }
            }
        }
    };

    stage_dma(0, 0);
    load_a(0);
    for (int ch = 0; ch < nchunk; ch += 2) {
        chunk(ch, 0);
        if (ch + 1 < nchunk) chunk(ch + 1, 1);
    }

    if (PW_EXP & 1) {
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int t = 0; t < TN; ++t) {
#if PW_EXP && defined(__HIP_DEVICE_COMPILE__)
                asm volatile("" ::"v"(acc[p][t]));
#endif
            }
        return;
    }
    // ---- epilogue: pointwise_split_kernel's, per 32-point subtile. Wave w, subtile p = slot (2 w + p) & 3 of 128-point block
    // 2 pt + (w >> 1).
    const int half = wave >> 1;
    using TT = std::true_type;
    using FF = std::false_type;
    auto subtile = [&](auto pc) {
        constexpr int p = decltype(pc)::value;
        const int slot = 2 * (wave & 1) + p;
        const int pw = p0 + wave * 64 + p * 32;
        unsigned vmask = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) vmask |= (pw + mfma_row(r, hi) < N ? 1u : 0u) << r;
        const bool full = pw + 32 <= N;
        auto tiles = [&](auto relu_c, auto store_c, auto stats_c, auto ext_c, auto full_c) {
            constexpr bool RELU = decltype(relu_c)::value, STORE = decltype(store_c)::value, STATS = decltype(stats_c)::value,
                           EXT = decltype(ext_c)::value, FULL = decltype(full_c)::value;
#pragma unroll
            for (int t = 0; t < TN; ++t) {
                const int o = o0 + 32 * t + li;
                float add = bias ? bias[o] : 0.f;
                if (cbias) add += cbias[(size_t)cloud * Coutp + o];
                float ps = 0.f, pq = 0.f, mx = -3.0e38f, mn = 3.0e38f;
                const unsigned loff = (unsigned)(4 * hi * ldy + o);
                const bool och = o < Cout;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[p][t][r] + add;
                    if (RELU) v = sed_vmax(v, 0.f);
                    const bool ok = FULL || ((vmask >> r) & 1u);
                    if (STORE && ok && och) {
                        float* rowbase = Y + ((size_t)cloud * N + pw + ((r & 3) + 8 * (r >> 2))) * ldy;     // uniform
                        rowbase[loff] = v;
                    }
                    if (ok) {
                        if (STATS) { ps += v; pq = fmaf(v, v, pq); }
                        if (EXT) { mx = sed_vmax(mx, v); mn = sed_vmin(mn, v); }
                    }
                }
                if (STATS) {
                    double d1 = (double)ps, d2 = (double)pq;
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) { d1 += __shfl_xor(d1, off, 64); d2 += __shfl_xor(d2, off, 64); }
                    if (lane == 0) { red[((half * 4 + slot) * TN + t) * 2] = d1; red[((half * 4 + slot) * TN + t) * 2 + 1] = d2; }
                }
                if (EXT) {
                    mx = fmaxf(mx, xor32(mx));
                    mn = fminf(mn, xor32(mn));
                    if (hi == 0) {
                        ext[((half * 4 + slot) * BN + 32 * t + li) * 2] = mx;
                        ext[((half * 4 + slot) * BN + 32 * t + li) * 2 + 1] = mn;
                    }
                }
            }
        };
        auto with_full = [&](auto relu_c, auto store_c, auto stats_c, auto ext_c) {
            if (full) tiles(relu_c, store_c, stats_c, ext_c, TT{});
            else tiles(relu_c, store_c, stats_c, ext_c, FF{});
        };
        auto d3 = [&](auto a, auto b, auto c) { if (flags & F_COLEXT) with_full(a, b, c, TT{}); else with_full(a, b, c, FF{}); };
        auto d2 = [&](auto a, auto b) { if (flags & F_STATS) d3(a, b, TT{}); else d3(a, b, FF{}); };
        auto d1 = [&](auto a) { if (flags & F_STORE) d2(a, TT{}); else d2(a, FF{}); };
        if (flags & F_RELU) d1(TT{}); else d1(FF{});
    };
    subtile(std::integral_constant<int, 0>{});
    subtile(std::integral_constant<int, 1>{});
    if (flags & (F_STATS | F_COLEXT)) __syncthreads();
    if ((flags & F_STATS) && tid < 2 * TN * 2) {
        const int h = tid / (TN * 2), t = (tid >> 1) % TN, which = tid & 1;
        if (2 * pt + h < nblk) {
            double s = 0.0;
            for (int w = 0; w < 4; ++w) s += red[((h * 4 + w) * TN + t) * 2 + which];
            const int ntile = nz * TN;
            part[(((size_t)cloud * nblk + 2 * pt + h) * ntile + zb * TN + t) * 2 + which] = s;
        }
    }
    if ((flags & F_COLEXT) && tid < 2 * BN) {
        const int h = tid / BN, c = tid % BN;
        if (2 * pt + h < nblk) {
            float mx = -3.0e38f, mn = 3.0e38f;
            for (int w = 0; w < 4; ++w) { mx = fmaxf(mx, ext[((h * 4 + w) * BN + c) * 2]); mn = fminf(mn, ext[((h * 4 + w) * BN + c) * 2 + 1]); }
            float* dst = colext + (((size_t)cloud * nblk + 2 * pt + h) * Coutp + o0 + c) * 2;
            dst[0] = mx;
            dst[1] = mn;
        }
    }
}

__global__ void gn_finalize_kernel(const double* __restrict__ part, int nblk, int ntile, int G, double count,
                                   float eps, float* __restrict__ stats) {
    // one wave per (cloud, group): lane l adds entries l, l + 64, ... of the group's nblk x tpg partial pairs, then a fixed
    // xor tree over the lanes (the serial loop of one thread per group was 22-35 us of dependent loads per launch)
    const int cloud = blockIdx.x, g = blockIdx.y, lane = threadIdx.x;
    const int tpg = ntile / G, n = nblk * tpg;
    double s = 0.0, q = 0.0;
    for (int e = lane; e < n; e += 64) {
        const int b = e / tpg, t = g * tpg + (e - b * tpg);
        const double* pp = part + (((size_t)cloud * nblk + b) * ntile + t) * 2;
        s += pp[0];
        q += pp[1];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { s += __shfl_xor(s, off, 64); q += __shfl_xor(q, off, 64); }
    if (lane != 0) return;
    const double mean = s / count;
    double var = q / count - mean * mean;
    if (var < 0.0) var = 0.0;
    stats[((size_t)cloud * G + g) * 2] = (float)mean;
    stats[((size_t)cloud * G + g) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
}

// out = scale * act(Y * a + b) + addend ;  a = rstd_g gamma_o ; b = beta_o - mean_g a
// A thread owns one float4 of channels and walks GN_PT points with it: the affine coefficients are formed once (vector loads
// of gamma / beta, one (mean, rstd) pair when the four channels share a group), all index arithmetic is 32-bit (the first
// version did a 64-bit division per float4 and four scalar parameter fetches per element: 3.2 TB/s on 1.3 GB layers).
constexpr int GN_PT = 4;
__global__ __launch_bounds__(256) void gn_apply_kernel(const float* __restrict__ Y, int ldy, int C, int G,
                                                       const float* __restrict__ stats,
                                                       const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, int act, float slope,
                                                       float scale, const float* __restrict__ addend, int lda,
                                                       float* __restrict__ out, int ldo, int N,
                                                       unsigned* __restrict__ rowmax) {
    const unsigned cloud = blockIdx.y;
    const unsigned c4n = (unsigned)C / 4;
    const unsigned rows_per_block = 256u / c4n > 0 ? 256u / c4n : 1u;      // c4n <= 256: whole rows per block pass
    // thread -> (row inside the pass, float4 of channels); blocks whose c4n does not divide 256 leave the tail threads idle
    const unsigned tr = threadIdx.x / c4n, c = (threadIdx.x - tr * c4n) * 4;
    if (c4n > 256u || tr >= rows_per_block) {
        if (c4n <= 256u) return;
    }
    f32x4 a4 = {1.f, 1.f, 1.f, 1.f}, b4 = {0.f, 0.f, 0.f, 0.f};
    const unsigned cpg = (unsigned)C / (unsigned)G;
    auto coeffs = [&](unsigned cc) {
        if (!stats) return;
        const f32x4 g4 = *(const f32x4*)(gamma + cc), be4 = *(const f32x4*)(beta + cc);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned g = (cc + u) / cpg;
            const float mean = stats[((size_t)cloud * G + g) * 2], rstd = stats[((size_t)cloud * G + g) * 2 + 1];
            a4[u] = rstd * g4[u];
            b4[u] = fmaf(-a4[u], mean, be4[u]);
        }
    };
    auto apply = [&](size_t row, unsigned cc) -> float {
        const f32x4 y = *(const f32x4*)(Y + row * ldy + cc);
        f32x4 ad = {0.f, 0.f, 0.f, 0.f};
        if (addend) ad = *(const f32x4*)(addend + row * lda + cc);
        f32x4 o;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float v = y[u];
            if (stats) v = fmaf(v, a4[u], b4[u]);
            if (act == 1) v = fmaxf(v, 0.f);
            else if (act == 2) v = v >= 0.f ? v : v * slope;
            o[u] = addend ? __fadd_rn(__fmul_rn(scale, v), ad[u]) : scale * v;   // (w * a) + x, SEDNet.py:322,326
        }
        *(f32x4*)(out + row * ldo + cc) = o;
        return fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3])));
    };
    // rowmax (optional): max |out| of every row, as float bits (non-negative floats order like their bit patterns), atomically
    // merged so that several calls can fill column ranges of the same rows -- the row bound of sed_pointwise_fwd_split16_f32.
    // The c4n lanes of a row that sit in one wave agree on their maximum first (power-of-two c4n only: the host checks).
    auto merge_rowmax = [&](size_t row, float m, bool valid) {
        const unsigned span = c4n < 64u ? c4n : 64u;
        for (unsigned off = span >> 1; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, (int)off, 64));
        if (valid && (threadIdx.x & (span - 1)) == 0) atomicMax(rowmax + row, __float_as_uint(m));
    };
    if (c4n <= 256u) {
        coeffs(c);
        const unsigned p0 = blockIdx.x * rows_per_block * GN_PT + tr;
#pragma unroll
        for (int k = 0; k < GN_PT; ++k) {
            const unsigned p = p0 + k * rows_per_block;
            float m = 0.f;
            if (p < (unsigned)N) m = apply((size_t)cloud * N + p, c);
            if (rowmax) merge_rowmax((size_t)cloud * N + p, m, p < (unsigned)N);
        }
    } else {                                                   // very wide layers: one row per block pass, channels strided
        for (int k = 0; k < GN_PT; ++k) {
            const unsigned p = blockIdx.x * GN_PT + k;
            if (p >= (unsigned)N) break;
            float m = 0.f;
            for (unsigned cc = threadIdx.x * 4; cc < (unsigned)C; cc += 1024) { coeffs(cc); m = fmaxf(m, apply((size_t)cloud * N + p, cc)); }
            if (rowmax) {
                for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
                if ((threadIdx.x & 63) == 0) atomicMax(rowmax + (size_t)cloud * N + p, __float_as_uint(m));
            }
        }
    }
}

// out = s3 * Y3 + (s1 * act1(GN1(Y1)) + act2(GN2(Y2)))     (round 6) -- the three elementwise passes behind the embedding head of
// SEDNet.py:320-326 in one: xs = relu(bn_seg(seg1)), x = w * relu(bn_asis(asis)) + xs, x' = w * pe + x. Each step is the
// arithmetic gn_apply_kernel compiles to for it (fmaf(y, a, b), the bare max, and ONE fused multiply-add for "scale * v + addend":
// hipcc contracts gn_apply's __fadd_rn(__fmul_rn(..)) -- DESIGN.md Appendix A -- so the fma is written out here), hence the same bits
// with 2.6 GB instead of 5.2 GB of traffic per forward. C % 4 == 0, C <= 1024, 256 % (C / 4) == 0; Y3 may be null (then x is written).
__global__ __launch_bounds__(256) void gn_apply_fused_kernel(const float* __restrict__ Y1, int ld1, const float* __restrict__ st1,
                                                             const float* __restrict__ g1, const float* __restrict__ b1, int G1,
                                                             int act1, float s1, const float* __restrict__ Y2, int ld2,
                                                             const float* __restrict__ st2, const float* __restrict__ g2,
                                                             const float* __restrict__ b2, int G2, int act2,
                                                             const float* __restrict__ Y3, int ld3, float s3,
                                                             float* __restrict__ out, int ldo, int C, int N) {
    const unsigned cloud = blockIdx.y;
    const unsigned c4n = (unsigned)C / 4, rpb = 256u / c4n;
    const unsigned tr = threadIdx.x / c4n, c = (threadIdx.x - tr * c4n) * 4;
    if (tr >= rpb) return;
    f32x4 a1, o1, a2, o2;
    auto coeffs = [&](const float* st, const float* g, const float* be, int G, f32x4& a4, f32x4& b4) {
        const unsigned cpg = (unsigned)C / (unsigned)G;
        const f32x4 g4 = *(const f32x4*)(g + c), be4 = *(const f32x4*)(be + c);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned gi = (c + u) / cpg;
            const float mean = st[((size_t)cloud * G + gi) * 2], rstd = st[((size_t)cloud * G + gi) * 2 + 1];
            a4[u] = rstd * g4[u];
            b4[u] = fmaf(-a4[u], mean, be4[u]);
        }
    };
    coeffs(st1, g1, b1, G1, a1, o1);
    coeffs(st2, g2, b2, G2, a2, o2);
    const unsigned p0 = blockIdx.x * rpb * GN_PT + tr;
#pragma unroll
    for (int k = 0; k < GN_PT; ++k) {
        const unsigned p = p0 + k * rpb;
        if (p >= (unsigned)N) continue;
        const size_t row = (size_t)cloud * N + p;
        const f32x4 y1 = *(const f32x4*)(Y1 + row * ld1 + c), y2 = *(const f32x4*)(Y2 + row * ld2 + c);
        f32x4 y3 = {0.f, 0.f, 0.f, 0.f};
        if (Y3) y3 = *(const f32x4*)(Y3 + row * ld3 + c);
        f32x4 o;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float v1 = fmaf(y1[u], a1[u], o1[u]);
            if (act1 == 1) v1 = fmaxf(v1, 0.f);
            float v2 = fmaf(y2[u], a2[u], o2[u]);
            if (act2 == 1) v2 = fmaxf(v2, 0.f);
            float x = fmaf(s1, v1, v2);
            if (Y3) x = fmaf(s3, y3[u], x);
            o[u] = x;
        }
        *(f32x4*)(out + row * ldo + c) = o;
    }
}

// x4[b][o] = relu(GN(extreme over N)) from per-block column extrema (mlp1 + bnmlp1 + max over N)
__global__ void colext_finalize_kernel(const float* __restrict__ colext, int nblk, int ldw, int C, int G,
                                       const float* __restrict__ stats, const float* __restrict__ gamma,
                                       const float* __restrict__ beta, float* __restrict__ out) {
    const int cloud = blockIdx.y, o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= C) return;
    float mx = -3.0e38f, mn = 3.0e38f;
    for (int b = 0; b < nblk; ++b) {
        const float* src = colext + (((size_t)cloud * nblk + b) * ldw + o) * 2;
        mx = fmaxf(mx, src[0]);
        mn = fminf(mn, src[1]);
    }
    const int g = o / (C / G);
    const float mean = stats[((size_t)cloud * G + g) * 2], rstd = stats[((size_t)cloud * G + g) * 2 + 1];
    const float a = rstd * gamma[o];
    const float b = fmaf(-a, mean, beta[o]);
    const float v = fmaf(a >= 0.f ? mx : mn, a, b);
    out[(size_t)cloud * C + o] = fmaxf(v, 0.f);
}

// out[b][o] = bias[o] + sum_c W[o][c] v[b][c]   (one wave per output; the repeated-global part of conv1)
__global__ __launch_bounds__(64) void gemv_bias_kernel(const float* __restrict__ W, int ldw, int K,
                                                       const float* __restrict__ bias, const float* __restrict__ v,
                                                       float* __restrict__ out, int Cout, int ldo) {
    const int cloud = blockIdx.y, o = blockIdx.x, lane = threadIdx.x;
    const float* w = W + (size_t)o * ldw;
    const float* x = v + (size_t)cloud * K;
    float acc = 0.f;
    for (int c = lane; c < K; c += 64) acc = fmaf(w[c], x[c], acc);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (lane == 0) out[(size_t)cloud * ldo + o] = acc + (bias ? bias[o] : 0.f);
}

__global__ void log_softmax_kernel(const float* __restrict__ in, int ld, int C, float* __restrict__ out, int ldo,
                                   size_t rows) {
    const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const float* x = in + r * ld;
    float m = x[0];
    for (int c = 1; c < C; ++c) m = fmaxf(m, x[c]);
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += expf(x[c] - m);
    const float ls = logf(s);
    for (int c = 0; c < C; ++c) out[r * ldo + c] = (x[c] - m) - ls;
}

}  // namespace

extern "C" size_t sed_pointwise_partials_bytes(int B, int N, int Coutp) {
    return (size_t)B * ((N + 127) / 128) * (Coutp / 32) * 2 * sizeof(double);
}
extern "C" size_t sed_pointwise_colext_bytes(int B, int N, int Coutp) {
    return (size_t)B * ((N + 127) / 128) * Coutp * 2 * sizeof(float);
}

// flags: 1 ReLU, 2 store Y, 4 statistics partials, 8 column extrema.
// Wt [K][Coutp] (K multiple of 32, Coutp multiple of 64, zero padded), bias [Coutp] or NULL,
// cbias [B][Coutp] or NULL, Y [B,N,ldy] (first Cout columns written).
static int pointwise_fwd(int B, int N, int K, int Coutp, int Cout, const float* X, int ldx, const float* Wt,
                         const float* bias, const float* cbias, float* Y, int ldy, void* partials, void* colext,
                         int flags, bool bf16, hipStream_t stream);

extern "C" int sed_pointwise_fwd_f32(int B, int N, int K, int Coutp, int Cout, const float* X, int ldx,
                                     const float* Wt, const float* bias, const float* cbias, float* Y, int ldy,
                                     void* partials, void* colext, int flags, hipStream_t stream) {
    return pointwise_fwd(B, N, K, Coutp, Cout, X, ldx, Wt, bias, cbias, Y, ldy, partials, colext, flags, false, stream);
}

// Same contract, products in bf16 (operands rounded to nearest even while staged, fp32 accumulate / epilogue): the
// training path of BASELINE configs[4]. Inputs and outputs stay fp32 in memory.
extern "C" int sed_pointwise_fwd_bf16(int B, int N, int K, int Coutp, int Cout, const float* X, int ldx,
                                      const float* Wt, const float* bias, const float* cbias, float* Y, int ldy,
                                      void* partials, void* colext, int flags, hipStream_t stream) {
    return pointwise_fwd(B, N, K, Coutp, Cout, X, ldx, Wt, bias, cbias, Y, ldy, partials, colext, flags, true, stream);
}

static int pointwise_fwd(int B, int N, int K, int Coutp, int Cout, const float* X, int ldx, const float* Wt,
                         const float* bias, const float* cbias, float* Y, int ldy, void* partials, void* colext,
                         int flags, bool bf16, hipStream_t stream) {
    if (B <= 0 || N <= 0 || !X || !Wt) return SED_EINVAL;
    if (K % 32 != 0 || Coutp % 64 != 0 || ldx % 4 != 0 || ldx < K || Cout > Coutp) return SED_EUNSUPPORTED;
    if ((flags & F_STORE) && (!Y || ldy < Cout)) return SED_EINVAL;
    if ((flags & F_STATS) && !partials) return SED_EINVAL;
    if ((flags & F_COLEXT) && !colext) return SED_EINVAL;
    const int nblk = (N + 127) / 128;
    if (Coutp % 128 == 0) {
        static std::atomic<unsigned long long> attr_set{0};      // devices whose limit has been raised (common.h)
        int attr_set_err = 0;
        if (sed_first_on_device(attr_set, &attr_set_err)) {
            hipError_t e = hipFuncSetAttribute((const void*)pointwise_kernel<4, false>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
            if (e != hipSuccess) return (int)e;
            e = hipFuncSetAttribute((const void*)pointwise_kernel<4, true>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
            if (e != hipSuccess) return (int)e;
            sed_mark_device(attr_set);
        } else if (attr_set_err) return attr_set_err;
        const size_t sm = (2 * 128 * 36 + 2 * 32 * 128) * sizeof(float) + 4 * 4 * 2 * sizeof(double) + 4 * 128 * 2 * sizeof(float);
        if (bf16)
            pointwise_kernel<4, true><<<dim3(nblk, B, Coutp / 128), 256, sm, stream>>>(
                X, ldx, K, Wt, Coutp, bias, cbias, Y, ldy, Cout, (double*)partials, (float*)colext, N, flags);
        else
            pointwise_kernel<4, false><<<dim3(nblk, B, Coutp / 128), 256, sm, stream>>>(
                X, ldx, K, Wt, Coutp, bias, cbias, Y, ldy, Cout, (double*)partials, (float*)colext, N, flags);
    } else {
        const size_t sm = (2 * 128 * 36 + 2 * 32 * 64) * sizeof(float) + 4 * 2 * 2 * sizeof(double) + 4 * 64 * 2 * sizeof(float);
        if (bf16)
            pointwise_kernel<2, true><<<dim3(nblk, B, Coutp / 64), 256, sm, stream>>>(
                X, ldx, K, Wt, Coutp, bias, cbias, Y, ldy, Cout, (double*)partials, (float*)colext, N, flags);
        else
            pointwise_kernel<2, false><<<dim3(nblk, B, Coutp / 64), 256, sm, stream>>>(
                X, ldx, K, Wt, Coutp, bias, cbias, Y, ldy, Cout, (double*)partials, (float*)colext, N, flags);
    }
    SED_LAUNCH_CHECK();
    return SED_OK;
}

// Pre-split weights for sed_pointwise_fwd_split_f32: W [Cout][K] fp32 (row stride ldwin) -> three bf16 planes [Coutp][K]
// (rows >= Cout: zeros).
__global__ __launch_bounds__(256) void split_weights_kernel(const float* __restrict__ W, int ldwin, int Cout, int Coutp,
                                                            int K, __bf16* __restrict__ Wp) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)Coutp * K) return;
    const int o = (int)(i / K), k = (int)(i % K);
    const float v = o < Cout ? W[(size_t)o * ldwin + k] : 0.f;
    const __bf16 p1 = (__bf16)v;
    const float r1 = v - (float)p1;
    const __bf16 p2 = (__bf16)r1;
    const float r2 = r1 - (float)p2;
    const size_t plane = (size_t)Coutp * K;
    Wp[i] = p1;
    Wp[plane + i] = p2;
    Wp[2 * plane + i] = (__bf16)r2;
}

// one wave per output channel: the row's magnitude, its scale, the two planes
__global__ __launch_bounds__(64) void split16_weights_kernel(const float* __restrict__ W, int ldwin, int Cout, int Coutp, int K,
                                                             h16* __restrict__ Wp) {
    const int o = blockIdx.x, lane = threadIdx.x;
    const size_t plane = (size_t)Coutp * K;
    float am = 0.f;
    if (o < Cout)
        for (int k = lane; k < K; k += 64) am = fmaxf(am, fabsf(W[(size_t)o * ldwin + k]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) am = fmaxf(am, __shfl_xor(am, off, 64));
    const float scale = split_row_scale(am);
    for (int k = lane; k < K; k += 64) {
        h16 a, b;
        split_pair(o < Cout ? W[(size_t)o * ldwin + k] : 0.f, scale, a, b);
        Wp[(size_t)o * K + k] = a;
        Wp[plane + (size_t)o * K + k] = b;
    }
    if (lane == 0) ((float*)(Wp + 2 * plane))[o] = 1.0f / scale;
}

extern "C" size_t sed_pointwise_split_weights_bytes(int Coutp, int K) {
    return (size_t)Coutp * K * 3 * sizeof(__bf16);
}

// W [Cout][K] (the Conv1d weight, row stride ldw >= K) -> split image in `wsplit` (sed_pointwise_split_weights_bytes).
// Done once per model by the caller (weights are static at inference).
extern "C" int sed_pointwise_split_weights_f32(int Cout, int Coutp, int K, const float* W, int ldw, void* wsplit,
                                               hipStream_t stream) {
    if (Cout <= 0 || Coutp < Cout || K <= 0 || K % 32 != 0 || !W || !wsplit || ldw < K) return SED_EINVAL;
    const size_t n = (size_t)Coutp * K;
    split_weights_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(W, ldw, Cout, Coutp, K, (__bf16*)wsplit);
    SED_LAUNCH_CHECK();
    return SED_OK;
}

// sed_pointwise_fwd_f32 with the products on the bf16 matrix pipe (three-way split, fp32-equivalent; see
// pointwise_split_kernel). wsplit = the image written by sed_pointwise_split_weights_f32 for the same (Coutp, K).
static int pointwise_split_launch(int B, int N, int K, int Coutp, int Cout, const float* X, int ldx, const void* wsplit,
                                  const float* bias, const float* cbias, float* Y, int ldy, void* partials, void* colext, int flags,
                                  const GnIn* gn, hipStream_t stream);
extern "C" int sed_pointwise_fwd_split_f32(int B, int N, int K, int Coutp, int Cout, const float* X, int ldx,
                                           const void* wsplit, const float* bias, const float* cbias, float* Y, int ldy,
                                           void* partials, void* colext, int flags, hipStream_t stream) {
    return pointwise_split_launch(B, N, K, Coutp, Cout, X, ldx, wsplit, bias, cbias, Y, ldy, partials, colext, flags, nullptr, stream);
}
// The same GEMM on a layer's PRE-normalisation output: every activation goes through that layer's GroupNorm + activation as it
// is loaded, x = act(X a_k + b_k) with a_k = rstd_g gamma_k, b_k = beta_k - mean_g a_k -- sed_gn_apply_f32's arithmetic (scale 1, no
// addend), the same bits -- so the normalised tensor is never written or read. in_stats [B][in_G][2] (sed_gn_finalize_f32), in_gamma /
// in_beta [K], in_act 0 none / 1 ReLU. K <= 512, K % in_G == 0.
extern "C" int sed_pointwise_fwd_split_gn_f32(int B, int N, int K, int Coutp, int Cout, const float* X, int ldx,
                                              const void* wsplit, const float* in_stats, const float* in_gamma,
                                              const float* in_beta, int in_G, int in_act, const float* bias, const float* cbias,
                                              float* Y, int ldy, void* partials, void* colext, int flags, hipStream_t stream) {
    if (!in_stats || !in_gamma || !in_beta || in_G <= 0 || (in_act != 0 && in_act != 1)) return SED_EINVAL;
    if (K > 512 || K % in_G != 0) return SED_EUNSUPPORTED;
    const GnIn gn{in_stats, in_gamma, in_beta, in_G, in_act};
    return pointwise_split_launch(B, N, K, Coutp, Cout, X, ldx, wsplit, bias, cbias, Y, ldy, partials, colext, flags, &gn, stream);
}
static int pointwise_split_launch(int B, int N, int K, int Coutp, int Cout, const float* X, int ldx, const void* wsplit,
                                  const float* bias, const float* cbias, float* Y, int ldy, void* partials, void* colext, int flags,
                                  const GnIn* gn, hipStream_t stream) {
    if (B <= 0 || N <= 0 || !X || !wsplit) return SED_EINVAL;
    if (K % 32 != 0 || Coutp % 64 != 0 || ldx % 4 != 0 || ldx < K || Cout > Coutp) return SED_EUNSUPPORTED;
    if ((flags & F_STORE) && (!Y || ldy < Cout)) return SED_EINVAL;
    if ((flags & F_STATS) && !partials) return SED_EINVAL;
    if ((flags & F_COLEXT) && !colext) return SED_EINVAL;
    const int nblk = (N + 127) / 128;
    const int nblk8 = (nblk + 7) / 8 * 8;
    auto smem = [](int BN, int TN) {
        return (size_t)(2 * 3 * BN * 40) * sizeof(__bf16) + 4 * TN * 2 * sizeof(double) + 4 * BN * 2 * sizeof(float);
    };
    if (Coutp % 128 == 0) {
        static std::atomic<unsigned long long> attr_set{0};      // devices whose limit has been raised (common.h)
        int attr_set_err = 0;
        if (sed_first_on_device(attr_set, &attr_set_err)) {
            hipError_t e = hipFuncSetAttribute((const void*)pointwise_split_kernel<4, false>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
            if (e != hipSuccess) return (int)e;
            e = hipFuncSetAttribute((const void*)pointwise_wide_kernel<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
            if (e != hipSuccess) return (int)e;
            e = hipFuncSetAttribute((const void*)pointwise_wide_kernel<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
            if (e != hipSuccess) return (int)e;
            e = hipFuncSetAttribute((const void*)pointwise_split_kernel<4, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    72 * 1024);
            if (e != hipSuccess) return (int)e;
            sed_mark_device(attr_set);
        } else if (attr_set_err) return attr_set_err;
        // both forms give the same bits: the wide one when the 32-point form's grid fills the chip's 512 workgroup slots at least
        // twice (a call with one or two clouds keeps the finer tiles: 80 workgroups of 256 points would leave CUs empty)
        if (PW_WIDE && (long)nblk * (Coutp / 128) * B >= 1024) {
            const int nwt8 = ((nblk + 1) / 2 + 7) / 8 * 8;
            const size_t sm = 2 * 3 * 128 * 64 + 2 * 4 * 4 * 2 * sizeof(double) + 2 * 4 * 128 * 2 * sizeof(float) +
                              (gn ? 2 * 512 * sizeof(float) : 0);
            if (gn)
                pointwise_wide_kernel<4, true><<<dim3(nwt8 * (Coutp / 128), B), 256, sm, stream>>>(
                    X, ldx, K, (const __bf16*)wsplit, Coutp, bias, cbias, Y, ldy, Cout, (double*)partials, (float*)colext, N, nblk,
                    flags, *gn);
            else
                pointwise_wide_kernel<4, false><<<dim3(nwt8 * (Coutp / 128), B), 256, sm, stream>>>(
                    X, ldx, K, (const __bf16*)wsplit, Coutp, bias, cbias, Y, ldy, Cout, (double*)partials, (float*)colext, N, nblk,
                    flags, GnIn{});
        } else if (gn)
            pointwise_split_kernel<4, false, true><<<dim3(nblk8 * (Coutp / 128), B), 256,
                                                     smem(128, 4) + 4 * 32 * sizeof(float) + 2 * 512 * sizeof(float), stream>>>(
                X, ldx, K, (const __bf16*)wsplit, Coutp, bias, cbias, Y, ldy, Cout, (double*)partials, (float*)colext, N, nblk,
                flags, nullptr, *gn);
        else
        pointwise_split_kernel<4, false><<<dim3(nblk8 * (Coutp / 128), B), 256, smem(128, 4), stream>>>(
            X, ldx, K, (const __bf16*)wsplit, Coutp, bias, cbias, Y, ldy, Cout, (double*)partials, (float*)colext, N, nblk,
            flags, nullptr);
    } else if (gn) {
        pointwise_split_kernel<2, false, true><<<dim3(nblk8 * (Coutp / 64), B), 256,
                                                 smem(64, 2) + 4 * 32 * sizeof(float) + 2 * 512 * sizeof(float), stream>>>(
            X, ldx, K, (const __bf16*)wsplit, Coutp, bias, cbias, Y, ldy, Cout, (double*)partials, (float*)colext, N, nblk, flags,
            nullptr, *gn);
    } else {
        pointwise_split_kernel<2, false><<<dim3(nblk8 * (Coutp / 64), B), 256, smem(64, 2), stream>>>(
            X, ldx, K, (const __bf16*)wsplit, Coutp, bias, cbias, Y, ldy, Cout, (double*)partials, (float*)colext, N, nblk,
            flags, nullptr);
    }
    SED_LAUNCH_CHECK();
    return SED_OK;
}

extern "C" size_t sed_pointwise_split16_weights_bytes(int Coutp, int K) {
    return (size_t)Coutp * K * 2 * sizeof(h16) + (size_t)Coutp * sizeof(float);
}

// W [Cout][K] -> the split-fp16 image of sed_pointwise_fwd_split16_f32: planes h, l [Coutp][K] fp16 of W 2^e (e per output
// channel, split16.h) followed by winv [Coutp] = 2^-e. Rows >= Cout: zeros.
extern "C" int sed_pointwise_split16_weights_f32(int Cout, int Coutp, int K, const float* W, int ldw, void* wsplit,
                                                 hipStream_t stream) {
    if (Cout <= 0 || Coutp < Cout || K <= 0 || K % 32 != 0 || !W || !wsplit || ldw < K) return SED_EINVAL;
    split16_weights_kernel<<<Coutp, 64, 0, stream>>>(W, ldw, Cout, Coutp, K, (h16*)wsplit);
    SED_LAUNCH_CHECK();
    return SED_OK;
}

// sed_pointwise_fwd_split_f32 in the two-plane split-fp16 form. rowmax [B*N]: the bit pattern of a float >= max_k |X[row][k]|
// for every row (what sed_gn_apply_f32 leaves in its `rowmax` argument; a bound that is too SMALL overflows fp16, one that is
// 2^j too large costs j bits). Coutp % 128 == 0 only.
extern "C" int sed_pointwise_fwd_split16_f32(int B, int N, int K, int Coutp, int Cout, const float* X, int ldx,
                                             const void* wsplit, const unsigned* rowmax, const float* bias,
                                             const float* cbias, float* Y, int ldy, void* partials, void* colext, int flags,
                                             hipStream_t stream) {
    if (B <= 0 || N <= 0 || !X || !wsplit || !rowmax) return SED_EINVAL;
    if (K % 32 != 0 || Coutp % 128 != 0 || ldx % 4 != 0 || ldx < K || Cout > Coutp) return SED_EUNSUPPORTED;
    if ((flags & F_STORE) && (!Y || ldy < Cout)) return SED_EINVAL;
    if ((flags & F_STATS) && !partials) return SED_EINVAL;
    if ((flags & F_COLEXT) && !colext) return SED_EINVAL;
    const int nblk = (N + 127) / 128;
    const int nblk8 = (nblk + 7) / 8 * 8;
    const size_t smem = (size_t)(2 * 2 * 128 * 40) * sizeof(h16) + 4 * 4 * 2 * sizeof(double) + 4 * 128 * 2 * sizeof(float) +
                        4 * 32 * sizeof(float);
    pointwise_split_kernel<4, true><<<dim3(nblk8 * (Coutp / 128), B), 256, smem, stream>>>(
        X, ldx, K, (const __bf16*)wsplit, Coutp, bias, cbias, Y, ldy, Cout, (double*)partials, (float*)colext, N, nblk, flags,
        rowmax);
    SED_LAUNCH_CHECK();
    return SED_OK;
}

// stats [B][G][2] = (mean, rstd) from partial sums over `count` values per group
extern "C" int sed_gn_finalize_f32(int B, int N, int Coutp, int G, double count, float eps, const void* partials,
                                   float* stats, hipStream_t stream) {
    if (B <= 0 || G <= 0 || G > 64 || !partials || !stats || (Coutp / 32) % G != 0) return SED_EINVAL;
    gn_finalize_kernel<<<dim3(B, G), 64, 0, stream>>>((const double*)partials, (N + 127) / 128, Coutp / 32, G, count, eps, stats);
    SED_LAUNCH_CHECK();
    return SED_OK;
}

// out = scale * act(GN(Y)) + addend ; stats NULL -> no normalisation; act 0 none / 1 ReLU / 2 LeakyReLU(slope)
extern "C" int sed_gn_apply_f32(int B, int N, int C, int G, const float* Y, int ldy, const float* stats,
                                const float* gamma, const float* beta, int act, float slope, float scale,
                                const float* addend, int lda, float* out, int ldo, unsigned* rowmax, hipStream_t stream) {
    if (B <= 0 || N <= 0 || !Y || !out || C % 4 != 0 || ldy % 4 != 0 || ldo % 4 != 0) return SED_EINVAL;
    if (rowmax && C <= 1024 && ((C / 4) & (C / 4 - 1)) != 0) return SED_EUNSUPPORTED;      // see merge_rowmax
    if (stats && (!gamma || !beta || G <= 0 || C % G != 0)) return SED_EINVAL;
    if (addend && lda % 4 != 0) return SED_EINVAL;
    const unsigned c4n = (unsigned)C / 4, rpb = c4n <= 256 ? 256 / c4n : 1;            // rows per block pass (see the kernel)
    const unsigned rows_per_block = rpb * GN_PT;
    gn_apply_kernel<<<dim3(((unsigned)N + rows_per_block - 1) / rows_per_block, B), 256, 0, stream>>>(
        Y, ldy, C, G ? G : 1, stats, gamma, beta, act, slope, scale, addend, lda, out, ldo, N, rowmax);
    SED_LAUNCH_CHECK();
    return SED_OK;
}

// out = scale3 * Y3 + (scale1 * act1(GN1(Y1)) + act2(GN2(Y2))): three sed_gn_apply_f32 passes in one (gn_apply_fused_kernel); Y3
// may be NULL. stats* [B][G*][2], gamma* / beta* [C]; act 0 none / 1 ReLU.
extern "C" int sed_gn_apply_fused_f32(int B, int N, int C, const float* Y1, int ld1, const float* stats1, const float* gamma1,
                                      const float* beta1, int G1, int act1, float scale1, const float* Y2, int ld2,
                                      const float* stats2, const float* gamma2, const float* beta2, int G2, int act2,
                                      const float* Y3, int ld3, float scale3, float* out, int ldo, hipStream_t stream) {
    if (B <= 0 || N <= 0 || !Y1 || !Y2 || !out || !stats1 || !stats2 || !gamma1 || !gamma2 || !beta1 || !beta2 || G1 <= 0 || G2 <= 0)
        return SED_EINVAL;
    if (C % 4 != 0 || C > 1024 || 256 % (C / 4) != 0 || C % G1 != 0 || C % G2 != 0 || ld1 % 4 || ld2 % 4 || ldo % 4 || (Y3 && ld3 % 4) ||
        (act1 != 0 && act1 != 1) || (act2 != 0 && act2 != 1))
        return SED_EUNSUPPORTED;
    const unsigned rows_per_block = 256u / ((unsigned)C / 4) * GN_PT;
    gn_apply_fused_kernel<<<dim3(((unsigned)N + rows_per_block - 1) / rows_per_block, B), 256, 0, stream>>>(
        Y1, ld1, stats1, gamma1, beta1, G1, act1, scale1, Y2, ld2, stats2, gamma2, beta2, G2, act2, Y3, ld3, scale3, out, ldo, C, N);
    SED_LAUNCH_CHECK();
    return SED_OK;
}

extern "C" int sed_colext_finalize_f32(int B, int N, int C, int G, const void* colext, const float* stats,
                                       const float* gamma, const float* beta, float* out, hipStream_t stream) {
    if (B <= 0 || !colext || !stats || !gamma || !beta || !out || C % G != 0) return SED_EINVAL;
    colext_finalize_kernel<<<dim3((C + 255) / 256, B), 256, 0, stream>>>((const float*)colext, (N + 127) / 128, C, C, G, stats,
                                                                        gamma, beta, out);
    SED_LAUNCH_CHECK();
    return SED_OK;
}

extern "C" int sed_gemv_bias_f32(int B, int Cout, int K, const float* W, int ldw, const float* bias, const float* v,
                                 float* out, int ldo, hipStream_t stream) {
    if (B <= 0 || Cout <= 0 || K <= 0 || !W || !v || !out || ldw < K || ldo < Cout) return SED_EINVAL;
    gemv_bias_kernel<<<dim3(Cout, B), 64, 0, stream>>>(W, ldw, K, bias, v, out, Cout, ldo);
    SED_LAUNCH_CHECK();
    return SED_OK;
}

extern "C" int sed_log_softmax_f32(size_t rows, int C, const float* in, int ld, float* out, int ldo,
                                   hipStream_t stream) {
    if (rows == 0 || C <= 0 || !in || !out) return SED_EINVAL;
    log_softmax_kernel<<<(unsigned)((rows + 255) / 256), 256, 0, stream>>>(in, ld, C, out, ldo, rows);
    SED_LAUNCH_CHECK();
    return SED_OK;
}
