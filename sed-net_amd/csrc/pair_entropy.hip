// Pairwise-distance statistics of the HPNet entropy weights, without materialising the M x M distance matrix.
//
// Replaces the two chunked double loops of /root/reference/src/smooth_normal_matrix.py:131-151 (compute_entropy):
//   mode 0:  sum_{i,j < M} ||u_i - u_j||                                  (-> average_dst, :133-138)
//   mode 1:  sum_{i,j < M} H(exp(-alpha ||u_i - u_j||)),  H(s) = -s log(s + eps) - (1 - s) log(1 - s + eps)   (:144-151)
// with u = features / interval already scaled by the caller (:134). Distances are formed from explicit differences
// like the reference (no |a|^2 + |b|^2 - 2ab cancellation); the matrix is symmetric, so only tiles bj >= bi are
// computed and off-diagonal tiles count twice. One workgroup = one 64 x 64 tile of pairs, 4 x 4 pairs per thread,
// features staged k-major through LDS; fp64 per-tile partial sums, summed by the caller in fixed order.
#include "common.h"
#include "split16.h"

namespace {

constexpr int KC = 32;      // feature chunk staged per LDS round
constexpr int LDT = 68;     // 64 rows + 4 pad (float4-aligned rows)

template <int MODE>
__global__ __launch_bounds__(256) void pair_entropy_kernel(const float* __restrict__ u, int ldu, int M, int K,
                                                           float alpha, const float* __restrict__ alpha_dev,
                                                           double* __restrict__ part) {
    if (alpha_dev) alpha = *alpha_dev;                         // (batched callers keep alpha on the device: no host round trip)
    const int bi = blockIdx.y, bj = blockIdx.x;
    const int nb = gridDim.x;
    double* out = part + (size_t)bi * nb + bj;
    if (bj < bi) {
        if (threadIdx.x == 0) *out = 0.0;
        return;
    }
    __shared__ __attribute__((aligned(16))) float As[KC * LDT], Bs[KC * LDT];
    __shared__ double red[4];
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    float acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = 0.f;

    for (int k0 = 0; k0 < K; k0 += KC) {
        // stage 64 rows x KC features of both row blocks, transposed to [k][row]; zero fill outside M / K
        for (int i = tid; i < 64 * KC; i += 256) {
            const int row = i / KC, k = i % KC;
            const int ra = bi * 64 + row, rb = bj * 64 + row;
            const bool kin = k0 + k < K;
            As[k * LDT + row] = (kin && ra < M) ? u[(size_t)ra * ldu + k0 + k] : 0.f;
            Bs[k * LDT + row] = (kin && rb < M) ? u[(size_t)rb * ldu + k0 + k] : 0.f;
        }
        __syncthreads();
        const int kn = min(KC, K - k0);
        for (int k = 0; k < kn; ++k) {
            const f32x4 a4 = *(const f32x4*)(As + k * LDT + ty * 4);
            const f32x4 b4 = *(const f32x4*)(Bs + k * LDT + tx * 4);
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const float d = a4[a] - b4[b];
                    acc[a][b] = fmaf(d, d, acc[a][b]);
                }
        }
        __syncthreads();
    }
    const float eps = 1e-7f;
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const bool ok = bi * 64 + ty * 4 + a < M && bj * 64 + tx * 4 + b < M;
            const float d = sqrtf(acc[a][b]);
            float v;
            if (MODE == 0) {
                v = d;
            } else {
                const float e = expf(-alpha * d);
                v = -e * logf(e + eps) - (1.f - e) * logf(1.f - e + eps);
            }
            s += ok ? v : 0.f;
        }
    double ds = (double)s;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ds += __shfl_xor(ds, off, 64);
    if ((tid & 63) == 0) red[tid >> 6] = ds;
    __syncthreads();
    if (tid == 0) *out = (red[0] + red[1] + red[2] + red[3]) * (bi == bj ? 1.0 : 2.0);
}

// The same statistics for K = 128 columns on the matrix pipe (round 3: the 128-d embedding's entropy was 0.9 ms per cloud on the
// vector ALUs, 60 % of the HPNet stage): ||u_i - u_j||^2 = |u_i|^2 + |u_j|^2 - 2 u_i.u_j with the dot products as split-fp16 MFMAs
// (l h + h l + h h on v_mfma_f32_32x32x16_f16, fp32 accumulate: <= 3 2^-24 |u_i||u_j| off the exact product) and the norms in fp32.
// The caller passes CENTRED rows (column means subtracted: distances do not change, the cancellation in the expansion does --
// |u| <= 0.5 sqrt(K) instead of an arbitrary offset); pairs (i, i) are set to 0 exactly. pair_entropy_split_kernel writes the
// (h, l) fp16 digits of 2^11 u (row = 128 h | 128 l halves) and the fp32 squared norms once per cloud; one workgroup of the pair
// kernel = one 64 x 64 tile of pairs, wave w = the 32 x 32 sub-tile (w >> 1, w & 1), both row blocks copied to LDS (272-byte rows).
constexpr int PE_K = 128;
constexpr float PE_S = 2048.0f;

__global__ __launch_bounds__(256) void pair_entropy_split_kernel(const float* __restrict__ u, int ldu, int M,
                                                                 h16* __restrict__ hl, float* __restrict__ n2) {
    const int row = blockIdx.x * 8 + (threadIdx.x >> 5), c4 = threadIdx.x & 31;        // 32 threads per row
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row < M) v = *(const f32x4*)(u + (size_t)row * ldu + 4 * c4);
    float q = v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) q += __shfl_xor(q, off, 64);
    if (row >= M) return;
    if (c4 == 0) n2[row] = q;
    h16 h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float x = v[e] * PE_S;
        h[e] = (h16)x;
        l[e] = (h16)(x - (float)h[e]);
    }
    *(h16x4*)(hl + (size_t)row * 2 * PE_K + 4 * c4) = h16x4{h[0], h[1], h[2], h[3]};
    *(h16x4*)(hl + (size_t)row * 2 * PE_K + PE_K + 4 * c4) = h16x4{l[0], l[1], l[2], l[3]};
}

template <int MODE>
__global__ __launch_bounds__(512) void pair_entropy_mfma_kernel(const h16* __restrict__ hl, const float* __restrict__ n2g, int M,
                                                                float alpha, const float* __restrict__ alpha_dev,
                                                                double* __restrict__ part) {
    // one workgroup = one 128 x 128 tile of pairs (the 64 x 64 form re-read every row 157 times from L2: 4.3 TB/s, the bound);
    // wave w = rows 32 (w >> 1) .. + 31 x columns 64 (w & 1) .. + 63: two 32 x 32 MFMA tiles
    if (alpha_dev) alpha = *alpha_dev;
    constexpr int K = PE_K, ROWB = 2 * K + 16, T = 128;        // bytes per staged row of a plane, tile edge
    const int bi = blockIdx.y, bj = blockIdx.x;
    const int nb = gridDim.x;
    double* out = part + (size_t)bi * nb + bj;
    if (bj < bi) return;                                       // (the partial buffer is zeroed by the caller)
    extern __shared__ __attribute__((aligned(16))) uint8_t pe_lds[];          // [row block a / b][h / l][T * ROWB]
    __shared__ float n2[2][T];
    __shared__ double red[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, hi = lane >> 5;
    typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
        const int r0 = (blk ? bj : bi) * T;
#pragma unroll
        for (int i = 0; i < 8; ++i) {                           // 128 rows x 32 pieces of 16 bytes (16 of h, 16 of l)
            const int f = tid + 512 * i, row = f >> 5, pc = f & 31;
            u32x4v v = {0u, 0u, 0u, 0u};
            if (r0 + row < M) v = *(const u32x4v*)((const uint8_t*)hl + (size_t)(r0 + row) * 4 * K + 16 * pc);
            *(u32x4v*)(pe_lds + ((blk * 2 + (pc >> 4)) * T + row) * ROWB + 16 * (pc & 15)) = v;
        }
        if (tid < T) n2[blk][tid] = r0 + tid < M ? n2g[r0 + tid] : 0.f;
    }
    __syncthreads();
    const int i0 = 32 * (wave >> 1), j0 = 64 * (wave & 1);
    f32x16 acc[2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    const uint8_t* ah = pe_lds + ((0 * 2 + 0) * T + i0 + li) * ROWB + 16 * hi;
    const uint8_t* al = pe_lds + ((0 * 2 + 1) * T + i0 + li) * ROWB + 16 * hi;
    const uint8_t* bh = pe_lds + ((1 * 2 + 0) * T + j0 + li) * ROWB + 16 * hi;
    const uint8_t* bl = pe_lds + ((1 * 2 + 1) * T + j0 + li) * ROWB + 16 * hi;
#pragma unroll
    for (int t = 0; t < K / 16; ++t) {                          // D[m = row i][n = row j]
        const h16x8 xh = *(const h16x8*)(ah + 32 * t), xl = *(const h16x8*)(al + 32 * t);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const h16x8 yh = *(const h16x8*)(bh + 32 * c * ROWB + 32 * t), yl = *(const h16x8*)(bl + 32 * c * ROWB + 32 * t);
            acc[c] = mfma16(xl, yh, acc[c]);
            acc[c] = mfma16(xh, yl, acc[c]);
            acc[c] = mfma16(xh, yh, acc[c]);
        }
    }
    const float eps = 1e-7f;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int jj = j0 + 32 * c + li, gj = bj * T + jj;
        const float nj = n2[1][jj];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ii = i0 + mfma_row(r, hi), gi = bi * T + ii;
            float d2 = (n2[0][ii] + nj) - 2.0f * (acc[c][r] * (1.0f / (PE_S * PE_S)));
            d2 = gi == gj ? 0.f : fmaxf(d2, 0.f);
            const float d = sqrtf(d2);
            float v;
            if (MODE == 0) {
                v = d;
            } else {
                const float e = expf(-alpha * d);
                v = -e * logf(e + eps) - (1.f - e) * logf(1.f - e + eps);
            }
            s += (gi < M && gj < M) ? v : 0.f;
        }
    }
    double ds = (double)s;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ds += __shfl_xor(ds, off, 64);
    if (lane == 0) red[wave] = ds;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        for (int w = 0; w < 8; ++w) t += red[w];
        *out = t * (bi == bj ? 1.0 : 2.0);
    }
}

}  // namespace

extern "C" size_t sed_pair_entropy_partials(int M) {
    const size_t nb = (size_t)(M + 63) / 64;
    return nb * nb;
}

// u [M,ldu] (first K columns used) on the device; partials [sed_pair_entropy_partials(M)] doubles, overwritten:
// their sum (taken by the caller) is the statistic over all M^2 ordered pairs.
extern "C" int sed_pair_entropy_f32(int M, int K, const float* u, int ldu, int mode, float alpha, const float* alpha_dev,
                                    double* partials, hipStream_t stream) {
    if (M <= 0 || K <= 0 || !u || !partials || ldu < K) return SED_EINVAL;
    if (mode != 0 && mode != 1) return SED_EINVAL;
    const int nb = (M + 63) / 64;
    dim3 grid(nb, nb);
    if (mode == 0) pair_entropy_kernel<0><<<grid, 256, 0, stream>>>(u, ldu, M, K, alpha, alpha_dev, partials);
    else pair_entropy_kernel<1><<<grid, 256, 0, stream>>>(u, ldu, M, K, alpha, alpha_dev, partials);
    SED_LAUNCH_CHECK();
    return SED_OK;
}

// K = 128 on the matrix pipe (pair_entropy_mfma_kernel): sed_pair_entropy_split_f32 turns CENTRED rows u [M,ldu] (the caller has
// subtracted the column means; ldu a multiple of 4) into the kernel's operands -- split [M][256] fp16 digits and norms [M] fp32,
// sed_pair_entropy_split_bytes(M) bytes together --, sed_pair_entropy_mfma_f32 takes them for either mode.
extern "C" size_t sed_pair_entropy_split_bytes(int M) { return M > 0 ? (size_t)M * (2 * PE_K * sizeof(h16) + sizeof(float)) : 0; }

extern "C" int sed_pair_entropy_split_f32(int M, int K, const float* u, int ldu, void* split, hipStream_t stream) {
    if (M <= 0 || !u || !split || ldu < K) return SED_EINVAL;
    if (K != PE_K || ldu % 4 != 0) return SED_EUNSUPPORTED;
    h16* hl = (h16*)split;
    float* n2 = (float*)((uint8_t*)split + (size_t)M * 2 * PE_K * sizeof(h16));
    pair_entropy_split_kernel<<<(M + 7) / 8, 256, 0, stream>>>(u, ldu, M, hl, n2);
    SED_LAUNCH_CHECK();
    return SED_OK;
}

extern "C" int sed_pair_entropy_mfma_f32(int M, const void* split, int mode, float alpha, const float* alpha_dev, double* partials,
                                         hipStream_t stream) {
    if (M <= 0 || !split || !partials || (mode != 0 && mode != 1)) return SED_EINVAL;
    const h16* hl = (const h16*)split;
    const float* n2 = (const float*)((const uint8_t*)split + (size_t)M * 2 * PE_K * sizeof(h16));
    const int nb = (M + 127) / 128;
    constexpr int sm = 2 * 2 * 128 * (2 * PE_K + 16);           // 139 264 B of stage planes
    static std::atomic<unsigned long long> attr{0};      // devices whose limit has been raised (common.h)
    int attr_err = 0;
    if (sed_first_on_device(attr, &attr_err)) {
        hipError_t e = hipFuncSetAttribute((const void*)pair_entropy_mfma_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, sm);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)pair_entropy_mfma_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, sm);
        if (e != hipSuccess) return (int)e;
        sed_mark_device(attr);
    } else if (attr_err) return attr_err;
    // (the caller sums sed_pair_entropy_partials(M) entries: this kernel fills the upper triangle of the first nb x nb)
    hipError_t e = hipMemsetAsync(partials, 0, sed_pair_entropy_partials(M) * sizeof(double), stream);
    if (e != hipSuccess) return (int)e;
    dim3 grid(nb, nb);
    if (mode == 0) pair_entropy_mfma_kernel<0><<<grid, 512, sm, stream>>>(hl, n2, M, alpha, alpha_dev, partials);
    else pair_entropy_mfma_kernel<1><<<grid, 512, sm, stream>>>(hl, n2, M, alpha, alpha_dev, partials);
    SED_LAUNCH_CHECK();
    return SED_OK;
}
