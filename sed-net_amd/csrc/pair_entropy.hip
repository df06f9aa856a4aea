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

namespace {

constexpr int KC = 32;      // feature chunk staged per LDS round
constexpr int LDT = 68;     // 64 rows + 4 pad (float4-aligned rows)

template <int MODE>
__global__ __launch_bounds__(256) void pair_entropy_kernel(const float* __restrict__ u, int ldu, int M, int K,
                                                           float alpha, double* __restrict__ part) {
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

}  // namespace

extern "C" size_t sed_pair_entropy_partials(int M) {
    const size_t nb = (size_t)(M + 63) / 64;
    return nb * nb;
}

// u [M,ldu] (first K columns used) on the device; partials [sed_pair_entropy_partials(M)] doubles, overwritten:
// their sum (taken by the caller) is the statistic over all M^2 ordered pairs.
extern "C" int sed_pair_entropy_f32(int M, int K, const float* u, int ldu, int mode, float alpha, double* partials,
                                    hipStream_t stream) {
    if (M <= 0 || K <= 0 || !u || !partials || ldu < K) return SED_EINVAL;
    if (mode != 0 && mode != 1) return SED_EINVAL;
    const int nb = (M + 63) / 64;
    dim3 grid(nb, nb);
    if (mode == 0) pair_entropy_kernel<0><<<grid, 256, 0, stream>>>(u, ldu, M, K, alpha, partials);
    else pair_entropy_kernel<1><<<grid, 256, 0, stream>>>(u, ldu, M, K, alpha, partials);
    SED_LAUNCH_CHECK();
    return SED_OK;
}
