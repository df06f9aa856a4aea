// Backward of the fused layers (training step).
//
//   EdgeConv layer   out[p,o] = act( gamma_o * (y[p,j*,o] - mu_g) * rstd_g + beta_o ),  y[p,j,o] = W1_o.(x_j - x_p) + W2_o.x_p
//   pointwise layer  out[p,o] = act( gamma_o * (y[p,o]   - mu_g) * rstd_g + beta_o ),  y = X W^T + b   (k = 1, j* = 0)
// (reference: /root/reference/src/SEDNet.py:37-45,78-98 and :300-329, differentiated by torch.autograd there).
//
// With dz = dout * act'(z), dyhat = gamma * dz, and the group means over all M = (C/G) N k positions
//   m1 = mean(dyhat), m2 = mean(dyhat * yhat)     (dyhat is non-zero only at the selected slot j*)
// GroupNorm's backward is  dy[p,j,o] = S[p,o] [j == j*] + alpha_g + kappa_g y[p,j,o]  with
//   S = rstd dyhat,  alpha = rstd (rstd m2 mu - m1),  kappa = -rstd^2 m2.
// gn_bwd_reduce_kernel produces S, per-(cloud, channel) sums (-> dgamma, dbeta, alpha, kappa); pointwise layers finish
// with gn_bwd_apply_kernel + two plain GEMMs (rocBLAS through torch.matmul on the host side); EdgeConv layers never
// materialise y or dy: edgeconv_bwd_weight_kernel (contraction over points, lane = output channel) and
// edgeconv_bwd_input_kernel (contraction over output channels, lane = point) each recompute y tile by tile in the
// register layout their second GEMM needs.
#include "common.h"

namespace {

constexpr int ACT_RELU = 1, ACT_LEAKY = 2;   // 0 = none

typedef __bf16 bbf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x16 mfma_bf(bbf16x8 a, bbf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}


// ---- S and per-(cloud, chunk, channel) partial sums of dz and dz * yhat ---------------------------------------
// grid (C/64, B, nchunk), block 256 = 4 row lanes x 64 channels; rows [chunk*256, +256)
__global__ __launch_bounds__(256) void gn_bwd_reduce_kernel(const float* __restrict__ dout, int ldd,
                                                            const float* __restrict__ y, int ldy,
                                                            const float* __restrict__ stats,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, int act, float slope,
                                                            int N, int C, int G, float* __restrict__ S,
                                                            double* __restrict__ part) {
    __shared__ double red[4][64][2];
    const int cloud = blockIdx.y, chunk = blockIdx.z;
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
    const int g = c / (C / G);
    const float mu = stats[((size_t)cloud * G + g) * 2], rstd = stats[((size_t)cloud * G + g) * 2 + 1];
    const float ga = gamma[c], be = beta[c];
    double s1 = 0.0, s2 = 0.0;
    const int r1 = min(N, (chunk + 1) * 256);
    for (int row = chunk * 256 + rl; row < r1; row += 4) {
        const size_t o = (size_t)cloud * N + row;
        const float yh = (y[o * ldy + c] - mu) * rstd;
        const float z = ga * yh + be;
        float dz = dout[o * ldd + c];
        if (act == ACT_RELU) dz = z > 0.f ? dz : 0.f;
        else if (act == ACT_LEAKY) dz = z > 0.f ? dz : dz * slope;
        S[o * C + c] = rstd * ga * dz;
        s1 += (double)dz;
        s2 += (double)(dz * yh);
    }
    red[rl][threadIdx.x & 63][0] = s1;
    red[rl][threadIdx.x & 63][1] = s2;
    __syncthreads();
    if (threadIdx.x < 128) {
        const int cc = threadIdx.x & 63, w = threadIdx.x >> 6;
        const double a = red[0][cc][w] + red[1][cc][w] + red[2][cc][w] + red[3][cc][w];
        part[(((size_t)cloud * gridDim.z + chunk) * C + blockIdx.x * 64 + cc) * 2 + w] = a;
    }
}

// per cloud: dbeta_b[c] = sum_n dz, dgamma_b[c] = sum_n dz yhat; alpha/kappa per group.  grid B, block 256
__global__ __launch_bounds__(256) void gn_bwd_finalize_kernel(const double* __restrict__ part, int nchunk, int C, int G,
                                                              double count, const float* __restrict__ stats,
                                                              const float* __restrict__ gamma,
                                                              float* __restrict__ dgamma_b, float* __restrict__ dbeta_b,
                                                              float* __restrict__ ak /*[B][G][2]*/) {
    const int cloud = blockIdx.x;
    const int cpg = C / G;
    for (int c = threadIdx.x; c < C; c += 256) {
        double s1 = 0.0, s2 = 0.0;
        for (int ch = 0; ch < nchunk; ++ch) {
            const double* pp = part + (((size_t)cloud * nchunk + ch) * C + c) * 2;
            s1 += pp[0];
            s2 += pp[1];
        }
        dbeta_b[(size_t)cloud * C + c] = (float)s1;
        dgamma_b[(size_t)cloud * C + c] = (float)s2;
    }
    __syncthreads();
    // group sums in fixed channel order (G <= 64 threads, each walks its channels)
    if (threadIdx.x < G) {
        const int g = threadIdx.x;
        double m1 = 0.0, m2 = 0.0;
        for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
            m1 += (double)gamma[c] * (double)dbeta_b[(size_t)cloud * C + c];
            m2 += (double)gamma[c] * (double)dgamma_b[(size_t)cloud * C + c];
        }
        m1 /= count;
        m2 /= count;
        const double mu = stats[((size_t)cloud * G + g) * 2], rstd = stats[((size_t)cloud * G + g) * 2 + 1];
        ak[((size_t)cloud * G + g) * 2] = (float)(rstd * (rstd * m2 * mu - m1));
        ak[((size_t)cloud * G + g) * 2 + 1] = (float)(-rstd * rstd * m2);
    }
}

// pointwise layers: dy = S + alpha_g + kappa_g y   (in place over S).  grid (ceil(N*C/4/256), B)
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(float* __restrict__ S, const float* __restrict__ y, int ldy,
                                                           const float* __restrict__ ak, int N, int C, int G) {
    const int cloud = blockIdx.y;
    const unsigned i4 = blockIdx.x * 256u + threadIdx.x;              // N * C / 4 < 2^32: 32-bit index arithmetic
    const unsigned c4n = (unsigned)C / 4;
    if (i4 >= (unsigned)N * c4n) return;
    const int row = (int)(i4 / c4n), c = (int)(i4 - (unsigned)row * c4n) * 4;
    const int g = c / (C / G);
    const float a = ak[((size_t)cloud * G + g) * 2], kp = ak[((size_t)cloud * G + g) * 2 + 1];
    const size_t o = (size_t)cloud * N + row;
    f32x4 s = *(f32x4*)(S + o * C + c);
    const f32x4 yy = *(const f32x4*)(y + o * ldy + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) s[e] = s[e] + a + kp * yy[e];
    *(f32x4*)(S + o * C + c) = s;
}

// ---- EdgeConv: weight gradients ------------------------------------------------------------------------------
// dW1t[c][o] = sum_{p,j} (x_j - x_p)[c] dy[p,j,o];  dW2t[c][o] = sum_p x_p[c] sum_j dy[p,j,o]
// One wave = 32 points x one 32-channel slab; y is recomputed with the forward's instruction sequence (lane = channel),
// dy is the B operand of the second MFMA (K = points), the A operand is the neighbour tile loaded channel-major.
// grid (nwg, B, Cout/32); wave w of block b handles point blocks b*4+w, +4*nwg, ...
template <int CH>
__global__ __launch_bounds__(256, 1) void edgeconv_bwd_weight_kernel(const float* __restrict__ x, int ldx,
                                                                     const int* __restrict__ idx, int k,
                                                                     const float* __restrict__ W1t,
                                                                     const float* __restrict__ W2t, int Cout, int G,
                                                                     const float* __restrict__ S,
                                                                     const uint8_t* __restrict__ jsel,
                                                                     const float* __restrict__ ak,
                                                                     float* __restrict__ part, int N) {
    constexpr int C = 2 * CH;
    constexpr int TC = (C + 31) / 32;
    __shared__ float w1[C * 32], w2[C * 32];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, hi = lane >> 5;
    const int cloud = blockIdx.y, o0 = blockIdx.z * 32;
    for (int i = tid; i < C * 32; i += 256) {
        const int c = i >> 5, o = i & 31;
        w1[i] = W1t[(size_t)c * Cout + o0 + o];
        w2[i] = W2t[(size_t)c * Cout + o0 + o];
    }
    __syncthreads();
    const int g = o0 / (Cout / G);
    const float alpha = ak[((size_t)cloud * G + g) * 2], kappa = ak[((size_t)cloud * G + g) * 2 + 1];
    const float* xb = x + (size_t)cloud * N * ldx;

    f32x16 dW1[TC], dW2[TC];
#pragma unroll
    for (int tc = 0; tc < TC; ++tc)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dW1[tc][r] = 0.f; dW2[tc][r] = 0.f; }

    const int nblk = (N + 31) / 32;
    for (int pb = blockIdx.x * 4 + wave; pb < nblk; pb += gridDim.x * 4) {
        const int p0 = pb * 32;
        const int p = p0 + li, pc = p < N ? p : N - 1;
        const int* ib = idx + ((size_t)cloud * N + pc) * k;
        float xc[CH];
#pragma unroll
        for (int s = 0; s < CH; ++s) xc[s] = xb[(size_t)pc * ldx + hi * CH + s];
        f32x16 base;
#pragma unroll
        for (int r = 0; r < 16; ++r) base[r] = 0.f;
#pragma unroll
        for (int s = 0; s < CH; ++s) base = mfma32(xc[s], w2[(hi * CH + s) * 32 + li], base);
        float Sr[16], xcT[TC][16];
        int js[16];
        unsigned vmask = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = p0 + mfma_row(r, hi);
            const bool ok = row < N;
            vmask |= (ok ? 1u : 0u) << r;
            const size_t o = ((size_t)cloud * N + (ok ? row : N - 1)) * Cout + o0 + li;
            Sr[r] = ok ? S[o] : 0.f;
            js[r] = ok ? (int)jsel[o] : -1;
#pragma unroll
            for (int tc = 0; tc < TC; ++tc) {
                const int c = 32 * tc + li;
                xcT[tc][r] = (ok && c < C) ? xb[(size_t)row * ldx + c] : 0.f;
            }
        }
        float dysum[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) dysum[r] = 0.f;

        for (int j = 0; j < k; ++j) {
            const int nb = ib[j];
            f32x16 acc = base;
#pragma unroll
            for (int s = 0; s < CH; ++s) {
                const float d = xb[(size_t)nb * ldx + hi * CH + s] - xc[s];
                acc = mfma32(d, w1[(hi * CH + s) * 32 + li], acc);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float dy = (vmask >> r) & 1u ? alpha + kappa * acc[r] : 0.f;
                dy += js[r] == j ? Sr[r] : 0.f;
                dysum[r] += dy;
                const int nbr = __shfl(nb, mfma_row(r, hi), 64);        // neighbour j of point p0 + row(r, hi)
#pragma unroll
                for (int tc = 0; tc < TC; ++tc) {
                    const int c = 32 * tc + li;
                    const float dT = ((vmask >> r) & 1u) && c < C ? xb[(size_t)nbr * ldx + c] - xcT[tc][r] : 0.f;
                    dW1[tc] = mfma32(dT, dy, dW1[tc]);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int tc = 0; tc < TC; ++tc) dW2[tc] = mfma32(xcT[tc][r], dysum[r], dW2[tc]);
    }
    // partials: part[cloud][wg][wave][2][C][Cout]
    float* pw = part + ((((size_t)cloud * gridDim.x + blockIdx.x) * 4 + wave) * 2) * C * Cout;
#pragma unroll
    for (int tc = 0; tc < TC; ++tc)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = 32 * tc + mfma_row(r, hi);
            if (c < C) {
                pw[(size_t)c * Cout + o0 + li] = dW1[tc][r];
                pw[(size_t)(C + c) * Cout + o0 + li] = dW2[tc][r];
            }
        }
}

// ---- EdgeConv weight gradients on the bf16 matrix pipe (TRAIN_BF16; C = 64 layers; round 2) -----------------------
// Same decomposition as edgeconv_bwd_weight_kernel -- lane = output channel, contraction over points, per-workgroup
// partials reduced in fixed order -- with v_mfma_f32_32x32x16_bf16 and one gather per edge block instead of one per slab:
// the NSLAB waves that own the slabs of one 32-point block share the block's difference tile d[p][c] = x_j - x_p (formed in
// fp32, rounded to bf16 like the forward does) through LDS; each of them gathers 64 / NSLAB channels of it, double
// buffered, one barrier per neighbour slot j. A wave then reads the tile twice: row-wise (its own point's 8 consecutive
// channels: the A operand of y = d W1 + base, K = channels) and column-wise (8 points' value of its channel: the A operand
// of dW1 = d^T dy, K = points, taken in the order the accumulator registers of y hold them, so dy goes into its B operand
// without leaving the registers). 8 MFMAs of 32 cycles per (edge block, slab) where the fp32 kernel issues 64 of 64 cycles
// and 32 scalar gathers. W1 / W2 columns of the wave's slab live in registers.
// grid (nwgx, B), 256 threads: wave w -> slab w % NSLAB of point block (group * PB + w / NSLAB), PB = 4 / NSLAB.
template <int NSLAB>
__global__ __launch_bounds__(256, 1) void edgeconv_bwd_weight_bf16_kernel(
    const float* __restrict__ x, int ldx, const int* __restrict__ idx, int k, const float* __restrict__ W1t,
    const float* __restrict__ W2t, int G, const float* __restrict__ S, const uint8_t* __restrict__ jsel,
    const float* __restrict__ ak, float* __restrict__ part, int N) {
    constexpr int C = 64, Cout = 32 * NSLAB, PB = 4 / NSLAB, LDT = 72 /* bf16 per tile row: 144 B */;
    constexpr int CPW = C / NSLAB;                         // channels of the tile a wave gathers
    __shared__ __attribute__((aligned(16))) __bf16 dtile[2][PB][32 * LDT];
    __shared__ __attribute__((aligned(16))) __bf16 xtile[PB][32 * LDT];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, hi = lane >> 5;
    const int slab = wave % NSLAB, sb = wave / NSLAB, o0 = 32 * slab;
    const int cloud = blockIdx.y;
    const float* xb = x + (size_t)cloud * N * ldx;
    const int g = o0 / (Cout / G);
    const float alpha = ak[((size_t)cloud * G + g) * 2], kappa = ak[((size_t)cloud * G + g) * 2 + 1];

    // B operands of y: W[c = 16 t + 8 hi + i][o = o0 + li]
    bbf16x8 w1r[4], w2r[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            w1r[t][i] = (__bf16)W1t[(size_t)(16 * t + 8 * hi + i) * Cout + o0 + li];
            w2r[t][i] = (__bf16)W2t[(size_t)(16 * t + 8 * hi + i) * Cout + o0 + li];
        }
    f32x16 dW1[2], dW2[2];
#pragma unroll
    for (int tc = 0; tc < 2; ++tc)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dW1[tc][r] = 0.f; dW2[tc][r] = 0.f; }

    // column-wise tile read: this lane's channel 32 tc + li of points row(8 u + i, hi), i = 0..7
    auto read_T = [&](const __bf16* tile, int tc, int u) {
        bbf16x8 a;
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = tile[((i & 3) + 16 * u + 8 * (i >> 2) + 4 * hi) * LDT + 32 * tc + li];
        return a;
    };

    const int nblk = (N + 31) / 32, ngrp = (nblk + PB - 1) / PB;
    for (int grp = blockIdx.x; grp < ngrp; grp += gridDim.x) {
        const int pb = grp * PB + sb;
        const int p0 = pb * 32;
        const int p = p0 + li, pc = p < N ? p : N - 1;
        const int* ib = idx + ((size_t)cloud * N + pc) * k;
        // this wave's channel slice of its point's row (fp32, kept for the differences) and of the x tile
        float xs[CPW / 2];
#pragma unroll
        for (int q = 0; q < CPW / 16; ++q) {
            const int c0 = CPW * slab + 16 * q + 8 * hi;
            const f32x4 a = *(const f32x4*)(xb + (size_t)pc * ldx + c0);
            const f32x4 b = *(const f32x4*)(xb + (size_t)pc * ldx + c0 + 4);
            bbf16x8 v;
#pragma unroll
            for (int i = 0; i < 4; ++i) { xs[8 * q + i] = a[i]; xs[8 * q + 4 + i] = b[i]; }
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = (__bf16)xs[8 * q + i];
            *(bbf16x8*)(&xtile[sb][li * LDT + c0]) = v;
        }
        f32x4 nx[CPW / 8];
        auto gather = [&](int j) {
            const int nb = ib[j];
#pragma unroll
            for (int q = 0; q < CPW / 16; ++q) {
                const int c0 = CPW * slab + 16 * q + 8 * hi;
                nx[2 * q] = *(const f32x4*)(xb + (size_t)nb * ldx + c0);
                nx[2 * q + 1] = *(const f32x4*)(xb + (size_t)nb * ldx + c0 + 4);
            }
        };
        auto put = [&](int buf) {
#pragma unroll
            for (int q = 0; q < CPW / 16; ++q) {
                bbf16x8 v;
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = (__bf16)((i < 4 ? nx[2 * q][i] : nx[2 * q + 1][i - 4]) - xs[8 * q + i]);
                *(bbf16x8*)(&dtile[buf][sb][li * LDT + CPW * slab + 16 * q + 8 * hi]) = v;
            }
        };
        gather(0);
        // per accumulator register r: point row(r, hi), channel o0 + li
        float Sr[16];
        unsigned jsp[4] = {0u, 0u, 0u, 0u};
        unsigned vmask = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = p0 + mfma_row(r, hi);
            const bool okr = row < N;
            vmask |= (okr ? 1u : 0u) << r;
            const size_t o = ((size_t)cloud * N + (okr ? row : N - 1)) * Cout + o0 + li;
            Sr[r] = okr ? S[o] : 0.f;
            jsp[r >> 2] |= (okr ? (unsigned)jsel[o] : 0xffu) << (8 * (r & 3));     // 255 never equals j (k <= 255)
        }
        put(0);
        __syncthreads();                                   // x tile and difference tile 0 complete
        f32x16 base;
#pragma unroll
        for (int r = 0; r < 16; ++r) base[r] = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
            base = mfma_bf(*(const bbf16x8*)(&xtile[sb][li * LDT + 16 * t + 8 * hi]), w2r[t], base);
        float dysum[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) dysum[r] = 0.f;

        for (int j = 0; j < k; ++j) {
            const int cur = j & 1;
            if (j + 1 < k) gather(j + 1);
            const __bf16* tile = dtile[cur][sb];
            f32x16 acc = base;
#pragma unroll
            for (int t = 0; t < 4; ++t)
                acc = mfma_bf(*(const bbf16x8*)(tile + li * LDT + 16 * t + 8 * hi), w1r[t], acc);
            bbf16x8 dyB[2];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float dy = (vmask >> r) & 1u ? fmaf(kappa, acc[r], alpha) : 0.f;
                dy += (int)((jsp[r >> 2] >> (8 * (r & 3))) & 0xffu) == j ? Sr[r] : 0.f;
                dysum[r] += dy;
                dyB[r >> 3][r & 7] = (__bf16)dy;
            }
#pragma unroll
            for (int tc = 0; tc < 2; ++tc)
#pragma unroll
                for (int u = 0; u < 2; ++u) dW1[tc] = mfma_bf(read_T(tile, tc, u), dyB[u], dW1[tc]);
            if (j + 1 < k) put(cur ^ 1);
            __syncthreads();
        }
        bbf16x8 dsB[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) dsB[r >> 3][r & 7] = (__bf16)dysum[r];
#pragma unroll
        for (int tc = 0; tc < 2; ++tc)
#pragma unroll
            for (int u = 0; u < 2; ++u) dW2[tc] = mfma_bf(read_T(xtile[sb], tc, u), dsB[u], dW2[tc]);
        __syncthreads();                                   // before the next group overwrites the tiles
    }
    // partials: part[cloud][wg][sb][2][C][Cout]; the NSLAB waves of a point block fill disjoint columns of one slot
    float* pw = part + ((((size_t)cloud * gridDim.x + blockIdx.x) * PB + sb) * 2) * C * Cout;
#pragma unroll
    for (int tc = 0; tc < 2; ++tc)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = 32 * tc + mfma_row(r, hi);
            pw[(size_t)c * Cout + o0 + li] = dW1[tc][r];
            pw[(size_t)(C + c) * Cout + o0 + li] = dW2[tc][r];
        }
}

// sum the weight partials in fixed order, two levels: grid (ceil(n / 256), nchunk) sums slots [chunk * per, +per) into
// out[chunk][n] (fp64 accumulate, fp32 store when it is the last level); the second launch (nchunk = 1) sums the chunks.
// (One level -- every thread walking all 2048 slots -- kept only n / 256 = 32 workgroups busy: 0.6 ms per call.)
__global__ void edgeconv_bwd_weight_reduce_kernel(const float* __restrict__ part, int nslot, size_t stride, int n, int per,
                                                  float* __restrict__ out, size_t ostride) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int s0 = blockIdx.y * per, s1 = min(nslot, s0 + per);
    double a = 0.0;
    for (int s = s0; s < s1; ++s) a += (double)part[(size_t)s * stride + i];
    out[(size_t)blockIdx.y * ostride + i] = (float)a;
}

// ---- EdgeConv: input gradients -------------------------------------------------------------------------------
// dx[nbr(p,j)] += W1^T dy[p,j];  dx[p] += sum_j (W2 - W1)^T dy[p,j].
// lane = point; y^T = W (x_j - x_p) is computed with the MFMA operands swapped, so that dy^T (rows = channels) is
// directly the B operand of the second MFMA (K = output channels). One workgroup = 128 points x one 32-channel output
// slab (blockIdx.z); the slabs' contributions meet in dx through fp32 atomics, like the neighbour scatter itself
// (torch's own index_select backward is atomic too).   grid (ceil(N/128), B, Cout/32)
// DET (default of the host wrapper, round 2): nothing is accumulated with atomics. Every (slab, edge) contribution is
// stored to E [cloud][slab][p k + j][C] and every (slab, point) self term to dxself [cloud][slab][p][C]; edge_gather_kernel
// then sums, per target row, the self terms and the contributions of its incoming edges in ascending edge order (reverse
// graph = the edge ids stably sorted by target) -- the same bits on every run.
template <int CH, bool DET>
__global__ __launch_bounds__(256, 1) void edgeconv_bwd_input_kernel(const float* __restrict__ x, int ldx,
                                                                    const int* __restrict__ idx, int k,
                                                                    const float* __restrict__ W1t,
                                                                    const float* __restrict__ W2t, int Cout, int G,
                                                                    const float* __restrict__ S,
                                                                    const uint8_t* __restrict__ jsel,
                                                                    const float* __restrict__ ak,
                                                                    float* __restrict__ dx, int lddx, int N,
                                                                    float* __restrict__ E, float* __restrict__ dxself) {
    constexpr int C = 2 * CH, LDW = 33, TC = C / 32, LDT = C + 1;
    __shared__ float w1[C * LDW], w2[C * LDW];
    __shared__ float tr[4][32 * LDT];        // per-wave transpose tile: df[point][channel] -> channel-major rows
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, hi = lane >> 5;
    const int cloud = blockIdx.y, o0 = blockIdx.z * 32;
    for (int i = tid; i < C * 32; i += 256) {
        const int c = i >> 5, o = i & 31;
        w1[c * LDW + o] = W1t[(size_t)c * Cout + o0 + o];
        w2[c * LDW + o] = W2t[(size_t)c * Cout + o0 + o];
    }
    __syncthreads();
    const float* xb = x + (size_t)cloud * N * ldx;
    float* dxb = dx + (size_t)cloud * N * lddx;
    const int p = blockIdx.x * 128 + wave * 32 + li;
    const bool ok = p < N;
    const int pc = ok ? p : N - 1;
    const int* ib = idx + ((size_t)cloud * N + pc) * k;
    const int g = o0 / (Cout / G);
    const float alpha = ak[((size_t)cloud * G + g) * 2], kappa = ak[((size_t)cloud * G + g) * 2 + 1];

    float xc[CH];
#pragma unroll
    for (int s = 0; s < CH; ++s) xc[s] = xb[(size_t)pc * ldx + hi * CH + s];
    f32x16 baseT;
#pragma unroll
    for (int r = 0; r < 16; ++r) baseT[r] = 0.f;
#pragma unroll
    for (int s = 0; s < CH; ++s) baseT = mfma32(w2[(hi * CH + s) * LDW + li], xc[s], baseT);
    float St[16], dysum[16];
    int js[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const size_t o = ((size_t)cloud * N + pc) * Cout + o0 + mfma_row(r, hi);
        St[r] = ok ? S[o] : 0.f;
        js[r] = ok ? (int)jsel[o] : -1;
        dysum[r] = 0.f;
    }
    f32x16 dxc[TC];
#pragma unroll
    for (int tc = 0; tc < TC; ++tc)
#pragma unroll
        for (int r = 0; r < 16; ++r) dxc[tc][r] = 0.f;

    for (int j = 0; j < k; ++j) {
        const int nb = ib[j];
        f32x16 yT = baseT;
#pragma unroll
        for (int s = 0; s < CH; ++s) {
            const float d = xb[(size_t)nb * ldx + hi * CH + s] - xc[s];
            yT = mfma32(w1[(hi * CH + s) * LDW + li], d, yT);
        }
        f32x16 df[TC];
#pragma unroll
        for (int tc = 0; tc < TC; ++tc)
#pragma unroll
            for (int r = 0; r < 16; ++r) df[tc][r] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float dy = ok ? alpha + kappa * yT[r] : 0.f;
            dy += js[r] == j ? St[r] : 0.f;
            dysum[r] += dy;
#pragma unroll
            for (int tc = 0; tc < TC; ++tc) df[tc] = mfma32(w1[(32 * tc + li) * LDW + mfma_row(r, hi)], dy, df[tc]);
        }
        // scatter: transpose through LDS so that one atomic instruction covers the C consecutive channels of ONE
        // neighbour row (a few cache lines) instead of 64 different rows
        float* mytr = tr[wave];
#pragma unroll
        for (int tc = 0; tc < TC; ++tc)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                mytr[li * LDT + 32 * tc + mfma_row(r, hi)] = df[tc][r];
                dxc[tc][r] -= df[tc][r];
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (DET) {
            float* Eb = E + ((size_t)cloud * gridDim.z + blockIdx.z) * N * k * C;
            const int pq0 = blockIdx.x * 128 + wave * 32;
            for (int q = 0; q < 32; ++q)
                if (pq0 + q < N && lane < C) Eb[((size_t)(pq0 + q) * k + j) * C + lane] = mytr[q * LDT + lane];
        } else {
            for (int q = 0; q < 32; ++q) {
                const int row = __shfl(ok ? nb : -1, q, 64);
                if (row >= 0 && lane < C) atomicAdd(dxb + (size_t)row * lddx + lane, mytr[q * LDT + lane]);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int tc = 0; tc < TC; ++tc)
            dxc[tc] = mfma32(w2[(32 * tc + li) * LDW + mfma_row(r, hi)], dysum[r], dxc[tc]);
    {
        float* mytr = tr[wave];
#pragma unroll
        for (int tc = 0; tc < TC; ++tc)
#pragma unroll
            for (int r = 0; r < 16; ++r) mytr[li * LDT + 32 * tc + mfma_row(r, hi)] = dxc[tc][r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (DET) {
            float* sb = dxself + ((size_t)cloud * gridDim.z + blockIdx.z) * N * C;
            const int pq0 = blockIdx.x * 128 + wave * 32;
            for (int q = 0; q < 32; ++q)
                if (pq0 + q < N && lane < C) sb[(size_t)(pq0 + q) * C + lane] = mytr[q * LDT + lane];
        } else {
            for (int q = 0; q < 32; ++q) {
                const int row = __shfl(ok ? p : -1, q, 64);
                if (row >= 0 && lane < C) atomicAdd(dxb + (size_t)row * lddx + lane, mytr[q * LDT + lane]);
            }
        }
    }
}

// ---- EdgeConv input gradients on the bf16 matrix pipe (TRAIN_BF16, deterministic path only; round 2) --------------
// One workgroup = 128 points (4 waves x 32, lane = point) x ALL output slabs: the neighbour row of an edge is gathered once
// (prefetched one neighbour ahead), the difference formed in fp32 and rounded to bf16 like the forward does, and per slab
//   y^T  = W1 (x_j - x_p) + W2 x_p            8 MFMAs (K = 64 input channels)
//   dy^T = alpha + kappa y^T + S [j == j*]     fp32, then rounded
//   df  += W1^T dy,   dxc += W2^T dy           4 + 4 MFMAs (K = 32 output channels)
// with v_mfma_f32_32x32x16_bf16 (fp32 accumulate) -- 16 MFMAs of 32 cycles per (edge block, slab) where the fp32 kernel
// issues 64 of 64 cycles. dy^T leaves the first product in the accumulator layout (row (r & 3) + 8 (r >> 2) + 4 hi of
// register r); the second products take their K index in exactly that order (slot 16 u + 8 hi + i <-> row of register
// 8 u + i), so dy is packed into its B operand without leaving the registers; W1 / W2 are staged in LDS once in both
// layouts ([o][c] for the first product, [c][slot(o)] per slab for the second). S sits in LDS per wave (every (j, slab)
// re-reads the point's 16 values), jsel comes from global memory. The edge contribution df (summed over slabs here, so E
// has ONE slab: a quarter of the fp32 kernel's traffic at Cout = 128) and the self term are stored straight from the
// accumulator layout, 16 bytes per lane and store, into the rows the gather kernel reads.  grid (ceil(N/128), B)
__global__ __launch_bounds__(256, 1) void edgeconv_bwd_input_bf16_kernel(
    const float* __restrict__ x, int ldx, const int* __restrict__ idx, int k, const float* __restrict__ W1t,
    const float* __restrict__ W2t, int Cout, int G, const float* __restrict__ S, const uint8_t* __restrict__ jsel,
    const float* __restrict__ ak, int N, float* __restrict__ E, float* __restrict__ dxself) {
    constexpr int C = 64, LDY = 72 /* bf16 per [o][c] row: 144 B */, LDD = 40 /* bf16 per [c][slot] row: 80 B */;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int nslab = Cout / 32, LDS_S = Cout + 4;
    __bf16* W1y = (__bf16*)smem;                          // [Cout][LDY]
    __bf16* W2y = W1y + Cout * LDY;
    __bf16* W1d = W2y + Cout * LDY;                       // [nslab][64][LDD]
    __bf16* W2d = W1d + nslab * C * LDD;
    float* Ss = (float*)(W2d + nslab * C * LDD);          // [4 waves][32][LDS_S]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, hi = lane >> 5;
    const int cloud = blockIdx.y;
    for (int i = tid; i < C * Cout; i += 256) {
        const int c = i / Cout, o = i - c * Cout;         // W*t are [c][o]
        const float a = W1t[i], b = W2t[i];
        W1y[o * LDY + c] = (__bf16)a;
        W2y[o * LDY + c] = (__bf16)b;
        const int ol = o & 31, g = ol >> 3, h2 = (ol >> 2) & 1, u3 = ol & 3;       // ol = u3 + 4 h2 + 8 g = row(r, h2), r = 4 g + u3
        const int r = 4 * g + u3, slot = (r >> 3) * 16 + h2 * 8 + (r & 7);
        W1d[((o >> 5) * C + c) * LDD + slot] = (__bf16)a;
        W2d[((o >> 5) * C + c) * LDD + slot] = (__bf16)b;
    }
    const float* xb = x + (size_t)cloud * N * ldx;
    const int p = blockIdx.x * 128 + wave * 32 + li;
    const bool ok = p < N;
    const int pc = ok ? p : N - 1;
    float* Sw = Ss + wave * 32 * LDS_S;
    for (int i = lane; i < 32 * (Cout / 4); i += 64) {    // this wave's 32 rows of S
        const int row = i / (Cout / 4), c4 = i - row * (Cout / 4);
        const int pr = blockIdx.x * 128 + wave * 32 + row;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (pr < N) v = *(const f32x4*)(S + ((size_t)cloud * N + pr) * Cout + 4 * c4);
        *(f32x4*)(Sw + row * LDS_S + 4 * c4) = v;
    }
    __syncthreads();
    const int* ib = idx + ((size_t)cloud * N + pc) * k;
    const uint8_t* jb = jsel + ((size_t)cloud * N + pc) * Cout;

    // this lane's K slots of the first product: channels 16 t + 8 hi + i
    float xp[4][8];
    bbf16x8 xpB[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const f32x4 a = *(const f32x4*)(xb + (size_t)pc * ldx + 16 * t + 8 * hi);
        const f32x4 b = *(const f32x4*)(xb + (size_t)pc * ldx + 16 * t + 8 * hi + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) { xp[t][i] = a[i]; xp[t][4 + i] = b[i]; }
#pragma unroll
        for (int i = 0; i < 8; ++i) xpB[t][i] = (__bf16)xp[t][i];
    }
    f32x16 dxc[2];
#pragma unroll
    for (int tc = 0; tc < 2; ++tc)
#pragma unroll
        for (int r = 0; r < 16; ++r) dxc[tc][r] = 0.f;

    f32x4 nx[8];
    auto gather = [&](int j) {
        const int nb = ib[j];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            nx[2 * t] = *(const f32x4*)(xb + (size_t)nb * ldx + 16 * t + 8 * hi);
            nx[2 * t + 1] = *(const f32x4*)(xb + (size_t)nb * ldx + 16 * t + 8 * hi + 4);
        }
    };
    gather(0);
    float* Eb = E + ((size_t)cloud * N + pc) * k * C;     // this point's edge rows
    for (int j = 0; j < k; ++j) {
        bbf16x8 dB[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < 8; ++i) dB[t][i] = (__bf16)((i < 4 ? nx[2 * t][i] : nx[2 * t + 1][i - 4]) - xp[t][i]);
        if (j + 1 < k) gather(j + 1);
        f32x16 df[2];
#pragma unroll
        for (int tc = 0; tc < 2; ++tc)
#pragma unroll
            for (int r = 0; r < 16; ++r) df[tc][r] = 0.f;
        for (int sl = 0; sl < nslab; ++sl) {
            f32x16 yT;
#pragma unroll
            for (int r = 0; r < 16; ++r) yT[r] = 0.f;
            const __bf16* w1 = W1y + (32 * sl + li) * LDY + 8 * hi;
            const __bf16* w2 = W2y + (32 * sl + li) * LDY + 8 * hi;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                yT = mfma_bf(*(const bbf16x8*)(w2 + 16 * t), xpB[t], yT);
                yT = mfma_bf(*(const bbf16x8*)(w1 + 16 * t), dB[t], yT);
            }
            const int g = (32 * sl) / (Cout / G);
            const float alpha = ak[((size_t)cloud * G + g) * 2], kappa = ak[((size_t)cloud * G + g) * 2 + 1];
            bbf16x8 dyB[2];
#pragma unroll
            for (int q = 0; q < 4; ++q) {                   // registers 4 q .. 4 q + 3 = channels 32 sl + 8 q + 4 hi + 0..3
                const f32x4 s4 = *(const f32x4*)(Sw + li * LDS_S + 32 * sl + 8 * q + 4 * hi);
                const unsigned js4 = *(const unsigned*)(jb + 32 * sl + 8 * q + 4 * hi);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int r = 4 * q + u;
                    float dy = ok ? fmaf(kappa, yT[r], alpha) : 0.f;
                    dy += (ok && (int)((js4 >> (8 * u)) & 0xffu) == j) ? s4[u] : 0.f;
                    dyB[r >> 3][r & 7] = (__bf16)dy;
                }
            }
#pragma unroll
            for (int tc = 0; tc < 2; ++tc) {
                const __bf16* a1 = W1d + ((sl * C) + 32 * tc + li) * LDD + 8 * hi;
                const __bf16* a2 = W2d + ((sl * C) + 32 * tc + li) * LDD + 8 * hi;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    df[tc] = mfma_bf(*(const bbf16x8*)(a1 + 16 * u), dyB[u], df[tc]);
                    dxc[tc] = mfma_bf(*(const bbf16x8*)(a2 + 16 * u), dyB[u], dxc[tc]);
                }
            }
        }
        // dx[nbr] += df (through E), dx[p] -= df
#pragma unroll
        for (int tc = 0; tc < 2; ++tc)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v = {df[tc][4 * q], df[tc][4 * q + 1], df[tc][4 * q + 2], df[tc][4 * q + 3]};
                if (ok) *(f32x4*)(Eb + (size_t)j * C + 32 * tc + 8 * q + 4 * hi) = v;
#pragma unroll
                for (int u = 0; u < 4; ++u) dxc[tc][4 * q + u] -= v[u];
            }
    }
    if (ok) {
        float* sb = dxself + ((size_t)cloud * N + p) * C;
#pragma unroll
        for (int tc = 0; tc < 2; ++tc)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v = {dxc[tc][4 * q], dxc[tc][4 * q + 1], dxc[tc][4 * q + 2], dxc[tc][4 * q + 3]};
                *(f32x4*)(sb + 32 * tc + 8 * q + 4 * hi) = v;
            }
    }
}

// dx[t] = sum_slab dxself[slab][t] + sum over the incoming edges e of t (ascending) of sum_slab E[slab][e].
// One wave per target row, lane = channel (C = 64).   grid (ceil(N / 4), B)
__global__ __launch_bounds__(256) void edge_gather_kernel(const float* __restrict__ E, const float* __restrict__ dxself,
                                                          const int* __restrict__ rptr, const int* __restrict__ redge,
                                                          int nslab, int N, int k, float* __restrict__ dx, int lddx) {
    constexpr int C = 64;
    const int cloud = blockIdx.y, lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= N) return;
    const size_t ne = (size_t)N * k;
    const float* Ec = E + (size_t)cloud * nslab * ne * C;
    const float* sc = dxself + (size_t)cloud * nslab * N * C;
    float acc = 0.f;
    for (int s = 0; s < nslab; ++s) acc += sc[((size_t)s * N + t) * C + lane];
    const int* rp = rptr + (size_t)cloud * (N + 1);
    const int* re = redge + (size_t)cloud * ne;
    const int i1 = rp[t + 1];
    int i = rp[t];
    for (; i + 4 <= i1; i += 4) {                          // four edges in flight; summed in order
        const int e0 = re[i], e1 = re[i + 1], e2 = re[i + 2], e3 = re[i + 3];
        float v[4][4];
        for (int s = 0; s < nslab && s < 4; ++s) {
            v[0][s] = Ec[((size_t)s * ne + e0) * C + lane];
            v[1][s] = Ec[((size_t)s * ne + e1) * C + lane];
            v[2][s] = Ec[((size_t)s * ne + e2) * C + lane];
            v[3][s] = Ec[((size_t)s * ne + e3) * C + lane];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            for (int s = 0; s < nslab && s < 4; ++s) acc += v[u][s];
    }
    for (; i < i1; ++i) {
        const int e = re[i];
        for (int s = 0; s < nslab; ++s) acc += Ec[((size_t)s * ne + e) * C + lane];
    }
    dx[((size_t)cloud * N + t) * lddx + lane] = acc;
}

}  // namespace

extern "C" size_t sed_gn_bwd_partials_bytes(int B, int N, int C) {
    return (size_t)B * ((N + 255) / 256) * C * 2 * sizeof(double);
}

// GroupNorm(+activation) backward, reduction part. dout [B,N,ldd], y [B,N,ldy] = pre-norm values (pointwise layers: the
// conv output; EdgeConv: the selected extreme `ysel`), stats [B,G,2] from the forward. count = values per group
// (C/G * N for pointwise, C/G * N * k for EdgeConv). Outputs: S [B,N,C], dgamma_b / dbeta_b [B,C] (sum over B on the
// host), ak [B,G,2] = (alpha, kappa).
extern "C" int sed_gn_bwd_reduce_f32(int B, int N, int C, int G, double count, const float* dout, int ldd,
                                     const float* y, int ldy, const float* stats, const float* gamma,
                                     const float* beta, int act, float slope, float* S, float* dgamma_b,
                                     float* dbeta_b, float* ak, void* partials, size_t partials_bytes,
                                     hipStream_t stream) {
    if (B <= 0 || N <= 0 || !dout || !y || !stats || !gamma || !beta || !S || !dgamma_b || !dbeta_b || !ak || !partials)
        return SED_EINVAL;
    if (C % 64 != 0 || G <= 0 || G > 64 || C % G != 0 || ldd < C || ldy < C) return SED_EUNSUPPORTED;
    if (partials_bytes < sed_gn_bwd_partials_bytes(B, N, C)) return SED_EINVAL;
    const int nchunk = (N + 255) / 256;
    gn_bwd_reduce_kernel<<<dim3(C / 64, B, nchunk), 256, 0, stream>>>(dout, ldd, y, ldy, stats, gamma, beta, act, slope,
                                                                      N, C, G, S, (double*)partials);
    SED_LAUNCH_CHECK();
    gn_bwd_finalize_kernel<<<B, 256, 0, stream>>>((const double*)partials, nchunk, C, G, count, stats, gamma, dgamma_b,
                                                  dbeta_b, ak);
    SED_LAUNCH_CHECK();
    return SED_OK;
}

// pointwise layers: S <- dy = S + alpha_g + kappa_g y
extern "C" int sed_gn_bwd_apply_f32(int B, int N, int C, int G, float* S, const float* y, int ldy, const float* ak,
                                    hipStream_t stream) {
    if (B <= 0 || N <= 0 || !S || !y || !ak) return SED_EINVAL;
    if (C % 4 != 0 || G <= 0 || C % G != 0 || (C / G) % 4 != 0 || ldy % 4 != 0) return SED_EUNSUPPORTED;
    const size_t n4 = (size_t)N * (C / 4);
    gn_bwd_apply_kernel<<<dim3((unsigned)((n4 + 255) / 256), B), 256, 0, stream>>>(S, y, ldy, ak, N, C, G);
    SED_LAUNCH_CHECK();
    return SED_OK;
}

static int edgeconv_bwd_nwg(int N) {
    const int nblk = (N + 31) / 32;
    int nwg = (nblk + 3) / 4;
    return nwg > 16 ? 16 : nwg;
}

extern "C" size_t sed_edgeconv_bwd_partials_bytes(int B, int N, int C, int Cout) {
    const size_t nslot = (size_t)B * edgeconv_bwd_nwg(N) * 4;                 // + the chunk sums of the two-level reduction
    return (nslot + (nslot + 31) / 32) * 2 * C * Cout * sizeof(float);
}

// (the bf16 form sums the slabs before it stores: ask with Cout = 32)
extern "C" size_t sed_edgeconv_bwd_edge_ws_bytes(int B, int N, int C, int Cout, int k) {
    return (size_t)B * (Cout / 32) * ((size_t)N * k + N) * C * sizeof(float);
}

// EdgeConv backward. Inputs as the forward plus S [B,N,Cout], jsel [B,N,Cout], ak [B,G,2] (sed_gn_bwd_reduce_f32 on
// dout / ysel with count = Cout/G * N * k). Outputs dW1t, dW2t [C][Cout] (overwritten) and, when dx != NULL,
// dx [B,N,lddx] (only C = 64 layers have an input gradient):
//   * rptr == NULL: accumulated with fp32 atomics (caller zero-initialises; order of the additions varies run to run);
//   * rptr [B,N+1], redge [B,N k] = the reverse graph (edge ids p k + j stably sorted by their target idx[p][j], row t =
//     redge[rptr[t] .. rptr[t+1])) and edge_ws (sed_edgeconv_bwd_edge_ws_bytes): deterministic -- per-edge contributions
//     are stored and gathered per target row in ascending edge order; columns 0..63 of dx are overwritten, the rest is left
//     alone. Up to 4 slabs (Cout <= 128).
// bf16 != 0 (C = 64, Cout = 64 / 128): weight-gradient products on the bf16 matrix pipe (edgeconv_bwd_weight_bf16_kernel) and,
// with the reverse graph, the input-gradient products too (edgeconv_bwd_input_bf16_kernel); operands rounded to nearest
// even, fp32 accumulate, the same fixed-order reductions.
extern "C" int sed_edgeconv_bwd_f32(int B, int N, int C, int Cout, int k, int G, const float* x, int ldx,
                                    const int* idx, const float* W1t, const float* W2t, const float* S,
                                    const uint8_t* jsel, const float* ak, float* dW1t, float* dW2t, float* dx,
                                    int lddx, void* partials, size_t partials_bytes, const int* rptr, const int* redge,
                                    void* edge_ws, size_t edge_ws_bytes, int bf16, hipStream_t stream) {
    if (B <= 0 || N <= 0 || k <= 0 || k > 255 || !x || !idx || !W1t || !W2t || !S || !jsel || !ak || !dW1t || !dW2t ||
        !partials)
        return SED_EINVAL;
    if (Cout % 32 != 0 || G <= 0 || (Cout / G) % 32 != 0 || ldx < C) return SED_EUNSUPPORTED;
    if (partials_bytes < sed_edgeconv_bwd_partials_bytes(B, N, C, Cout)) return SED_EINVAL;
    const int nwg = edgeconv_bwd_nwg(N);
    float* part = (float*)partials;
    dim3 grid(nwg, B, Cout / 32);
    int nslot = B * nwg * 4;
    if (bf16 && C == 64 && (Cout == 64 || Cout == 128)) {
        // slots = workgroups x point blocks per workgroup <= 64 per cloud = the buffer sed_edgeconv_bwd_partials_bytes sizes
        const int PB = 4 / (Cout / 32), ngrp = ((N + 31) / 32 + PB - 1) / PB;
        int nwgx = 4 * nwg / PB;
        if (nwgx > ngrp) nwgx = ngrp;
        if (Cout == 128)
            edgeconv_bwd_weight_bf16_kernel<4><<<dim3(nwgx, B), 256, 0, stream>>>(x, ldx, idx, k, W1t, W2t, G, S, jsel, ak,
                                                                                  part, N);
        else
            edgeconv_bwd_weight_bf16_kernel<2><<<dim3(nwgx, B), 256, 0, stream>>>(x, ldx, idx, k, W1t, W2t, G, S, jsel, ak,
                                                                                  part, N);
        nslot = B * nwgx * PB;
    } else if (C == 6)
        edgeconv_bwd_weight_kernel<3><<<grid, 256, 0, stream>>>(x, ldx, idx, k, W1t, W2t, Cout, G, S, jsel, ak, part, N);
    else if (C == 64)
        edgeconv_bwd_weight_kernel<32><<<grid, 256, 0, stream>>>(x, ldx, idx, k, W1t, W2t, Cout, G, S, jsel, ak, part, N);
    else
        return SED_EUNSUPPORTED;
    SED_LAUNCH_CHECK();
    // partial layout [slot][2][C][Cout]
    const int n = C * Cout;
    // level 1: chunks of 32 slots -> the head of the (now consumed) partial buffer is NOT reusable while it is being read, so
    // the chunk sums go behind the last slot; sed_edgeconv_bwd_partials_bytes reserves that room
    const int per = 32, nchunk = (nslot + per - 1) / per;
    float* lvl = part + (size_t)nslot * 2 * n;
    edgeconv_bwd_weight_reduce_kernel<<<dim3((2 * n + 255) / 256, nchunk), 256, 0, stream>>>(part, nslot, (size_t)2 * n, 2 * n,
                                                                                            per, lvl, (size_t)2 * n);
    SED_LAUNCH_CHECK();
    edgeconv_bwd_weight_reduce_kernel<<<dim3((n + 255) / 256, 1), 256, 0, stream>>>(lvl, nchunk, (size_t)2 * n, n, nchunk,
                                                                                   dW1t, 0);
    SED_LAUNCH_CHECK();
    edgeconv_bwd_weight_reduce_kernel<<<dim3((n + 255) / 256, 1), 256, 0, stream>>>(lvl + n, nchunk, (size_t)2 * n, n, nchunk,
                                                                                   dW2t, 0);
    SED_LAUNCH_CHECK();
    if (dx) {
        if (C != 64 || lddx < C) return SED_EUNSUPPORTED;
        const dim3 ig((N + 127) / 128, B, Cout / 32);
        if (rptr) {
            if (!redge || !edge_ws || Cout > 128) return SED_EINVAL;
            if (edge_ws_bytes < sed_edgeconv_bwd_edge_ws_bytes(B, N, C, bf16 ? 32 : Cout, k)) return SED_EINVAL;
            float* E = (float*)edge_ws;
            if (bf16) {
                float* dxself = E + (size_t)B * N * k * C;
                const size_t sm = (size_t)(2 * Cout * 72 + 2 * (Cout / 32) * 64 * 40) * 2 + (size_t)4 * 32 * (Cout + 4) * 4;
                static size_t sm_set = 0;
                if (sm > sm_set) {
                    hipError_t e = hipFuncSetAttribute((const void*)edgeconv_bwd_input_bf16_kernel,
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
                    if (e != hipSuccess) return (int)e;
                    sm_set = sm;
                }
                edgeconv_bwd_input_bf16_kernel<<<dim3((N + 127) / 128, B), 256, sm, stream>>>(
                    x, ldx, idx, k, W1t, W2t, Cout, G, S, jsel, ak, N, E, dxself);
                SED_LAUNCH_CHECK();
                edge_gather_kernel<<<dim3((N + 3) / 4, B), 256, 0, stream>>>(E, dxself, rptr, redge, 1, N, k, dx, lddx);
            } else {
                float* dxself = E + (size_t)B * (Cout / 32) * N * k * C;
                edgeconv_bwd_input_kernel<32, true><<<ig, 256, 0, stream>>>(x, ldx, idx, k, W1t, W2t, Cout, G, S, jsel, ak,
                                                                             dx, lddx, N, E, dxself);
                SED_LAUNCH_CHECK();
                edge_gather_kernel<<<dim3((N + 3) / 4, B), 256, 0, stream>>>(E, dxself, rptr, redge, Cout / 32, N, k, dx,
                                                                             lddx);
            }
        } else {
            edgeconv_bwd_input_kernel<32, false><<<ig, 256, 0, stream>>>(x, ldx, idx, k, W1t, W2t, Cout, G, S, jsel, ak,
                                                                          dx, lddx, N, nullptr, nullptr);
        }
        SED_LAUNCH_CHECK();
    }
    return SED_OK;
}
