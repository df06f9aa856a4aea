// General GEMM for the backward of the pointwise layers, bf16 or fp32 products.
//
//   C[M,N] = sum_k A(m,k) B(k,n)        A(m,k) = transA ? A[k lda + m] : A[m lda + k]
//                                       B(k,n) = transB ? B[n ldb + k] : B[k ldb + n]
// The reference gets these products from torch.autograd (cuBLAS) when it differentiates Conv1d (train_sed_net.py:272);
// round 1 called torch.matmul -> rocBLAS for them. The two shapes of a layer's backward:
//   dX [P, K]    = dy [P, Cout] . W [Cout, K]          (no transposes; P = B N points)
//   dW [Cout, K] = dy^T [Cout, P] . X [P, K]           (transA; the reduction runs over the P points: split over grid.z
//                                                        into partial products that a second kernel adds in fixed order)
// One workgroup = a 128 x 128 tile of C, 4 waves of 64 x 64 (2 x 2 MFMA tiles), reduction in chunks of 32 through a
// double-buffered LDS ring. Operands are fp32 in memory and are converted while they are staged: LDS holds both tiles
// "k-contiguous" ([row][32 k], bf16: 80-byte rows; fp32: 33-float rows), whatever their layout in memory (a transposed
// operand is scattered element-wise into LDS), so every bf16 MFMA operand is one ds_read_b128.
//   BF16: v_mfma_f32_32x32x16_bf16, operands rounded to nearest even, fp32 accumulate (training, BASELINE configs[4]);
//   else: v_mfma_f32_32x32x2_f32, exact fp32 fma chains (the fp32 training path).
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

template <bool BF16>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const float* __restrict__ A, int lda, int transA,
                                                      const float* __restrict__ Bm, int ldb, int transB,
                                                      float* __restrict__ C, int ldc, int M, int N, int K,
                                                      int kchunk, size_t c_split_stride) {
    constexpr int LDH = 40;                          // bf16 row stride (halves)
    constexpr int LDF = 33;                          // fp32 row stride (floats)
    constexpr int TILE_BYTES = BF16 ? 128 * LDH * 2 : 128 * LDF * 4;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];        // [2 buffers][A tile | B tile]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, hi = lane >> 5;
    const int m0 = blockIdx.x * 128, n0 = blockIdx.y * 128;
    const int k0 = blockIdx.z * kchunk, k1 = min(K, k0 + kchunk);
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;

    // staging: 128 rows x 32 k = 1024 float4 per operand per chunk -> 4 per thread
    f32x4 sa[4], sb[4];
    auto load_op = [&](const float* P, int ld, int trans_mem_rows, int r0, int R, int kk, f32x4 (&dst)[4]) {
        // trans_mem_rows: 1 if memory is [k][row] (row contiguous), 0 if [row][k] (k contiguous)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = tid + 256 * u;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (!trans_mem_rows) {
                const int row = i >> 3, k4 = (i & 7) * 4;              // float4 along k
                const int r = r0 + row, k = kk + k4;
                if (r < R) {
                    const float* src = P + (size_t)r * ld + k;
                    if (k + 3 < k1) v = *(const f32x4*)src;
                    else
                        for (int e = 0; e < 4; ++e) if (k + e < k1) v[e] = src[e];
                }
            } else {
                const int kr = i >> 5, r4 = (i & 31) * 4;              // float4 along the row index
                const int k = kk + kr, r = r0 + r4;
                if (k < k1) {
                    const float* src = P + (size_t)k * ld + r;
                    if (r + 3 < R) v = *(const f32x4*)src;
                    else
                        for (int e = 0; e < 4; ++e) if (r + e < R) v[e] = src[e];
                }
            }
            dst[u] = v;
        }
    };
    auto store_op = [&](uint8_t* tile, int trans_mem_rows, const f32x4 (&src)[4]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = tid + 256 * u;
            if (!trans_mem_rows) {
                const int row = i >> 3, k4 = (i & 7) * 4;
                if (BF16) {
                    bf16x4 h;
#pragma unroll
                    for (int e = 0; e < 4; ++e) h[e] = (__bf16)src[u][e];
                    *(bf16x4*)(tile + (row * LDH + k4) * 2) = h;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) ((float*)tile)[row * LDF + k4 + e] = src[u][e];
                }
            } else {
                const int kr = i >> 5, r4 = (i & 31) * 4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (BF16) ((__bf16*)tile)[(r4 + e) * LDH + kr] = (__bf16)src[u][e];
                    else ((float*)tile)[(r4 + e) * LDF + kr] = src[u][e];
                }
            }
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int a_rows_contig = transA ? 1 : 0;        // A in memory [k][m] when transA
    const int b_rows_contig = transB ? 0 : 1;        // B in memory [k][n] unless transB
    const bool a_vec_ok = (lda % 4 == 0), b_vec_ok = (ldb % 4 == 0);
    (void)a_vec_ok; (void)b_vec_ok;
    int cur = 0;
    if (k0 < k1) {
        load_op(A, lda, a_rows_contig, m0, M, k0, sa);
        load_op(Bm, ldb, b_rows_contig, n0, N, k0, sb);
        store_op(smem, a_rows_contig, sa);
        store_op(smem + TILE_BYTES, b_rows_contig, sb);
    }
    __syncthreads();
    for (int kk = k0; kk < k1; kk += 32) {
        const bool more = kk + 32 < k1;
        if (more) {
            load_op(A, lda, a_rows_contig, m0, M, kk + 32, sa);
            load_op(Bm, ldb, b_rows_contig, n0, N, kk + 32, sb);
        }
        const uint8_t* at = smem + cur * 2 * TILE_BYTES;
        const uint8_t* bt = at + TILE_BYTES;
        if (BF16) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                bf16x8 af[2], bf[2];
#pragma unroll
                for (int a = 0; a < 2; ++a) af[a] = *(const bf16x8*)(at + ((wm + 32 * a + li) * LDH + 16 * s2 + 8 * hi) * 2);
#pragma unroll
                for (int b = 0; b < 2; ++b) bf[b] = *(const bf16x8*)(bt + ((wn + 32 * b + li) * LDH + 16 * s2 + 8 * hi) * 2);
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bf[b], acc[a][b], 0, 0, 0);
            }
        } else {
            const float* af32 = (const float*)at;
            const float* bf32 = (const float*)bt;
#pragma unroll 4
            for (int s = 0; s < 16; ++s) {
                float av[2], bv[2];
#pragma unroll
                for (int a = 0; a < 2; ++a) av[a] = af32[(wm + 32 * a + li) * LDF + 2 * s + hi];
#pragma unroll
                for (int b = 0; b < 2; ++b) bv[b] = bf32[(wn + 32 * b + li) * LDF + 2 * s + hi];
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) acc[a][b] = mfma32(av[a], bv[b], acc[a][b]);
            }
        }
        if (more) {
            store_op(smem + (cur ^ 1) * 2 * TILE_BYTES, a_rows_contig, sa);
            store_op(smem + (cur ^ 1) * 2 * TILE_BYTES + TILE_BYTES, b_rows_contig, sb);
        }
        __syncthreads();
        cur ^= 1;
    }
    float* Cz = C + (size_t)blockIdx.z * c_split_stride;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm + 32 * a + mfma_row(r, hi), n = n0 + wn + 32 * b + li;
                if (m < M && n < N) Cz[(size_t)m * ldc + n] = acc[a][b][r];
            }
}

// C[i] = sum_z part[z][i] in fixed order (deterministic split-K)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, size_t stride, int nsplit,
                                                            float* __restrict__ C, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int z = 0; z < nsplit; ++z) s += part[(size_t)z * stride + i];
    C[i] = s;
}

}  // namespace

// number of reduction splits sed_gemm_f32 uses for this shape (workspace = nsplit * M * N floats when nsplit > 1)
extern "C" int sed_gemm_splits(int M, int N, int K) {
    const long tiles = (long)((M + 127) / 128) * ((N + 127) / 128);
    if (tiles >= 256 || K <= 4096) return 1;
    long s = (512 + tiles - 1) / tiles;                   // aim at ~2 workgroups per CU
    const long maxs = (K + 1023) / 1024;                  // at least 1024 reduction steps per split
    if (s > maxs) s = maxs;
    return (int)(s < 1 ? 1 : (s > 256 ? 256 : s));
}

// C [M,N] (ldc == N when split) = op(A) op(B); bf16 != 0 -> bf16 products (fp32 accumulate). workspace: >= splits*M*N floats
// when sed_gemm_splits(M, N, K) > 1, else may be NULL.
extern "C" int sed_gemm_f32(int M, int N, int K, const float* A, int lda, int transA, const float* B, int ldb,
                            int transB, float* C, int ldc, int bf16, void* workspace, size_t workspace_bytes,
                            hipStream_t stream) {
    if (M <= 0 || N <= 0 || K <= 0 || !A || !B || !C) return SED_EINVAL;
    if ((transA ? lda < M : lda < K) || (transB ? ldb < K : ldb < N) || ldc < N) return SED_EINVAL;
    if (lda % 4 != 0 || ldb % 4 != 0) return SED_EUNSUPPORTED;            // float4 staging
    const int nsplit = sed_gemm_splits(M, N, K);
    float* dst = C;
    size_t stride = 0;
    int ldo = ldc;
    if (nsplit > 1) {
        if (!workspace || workspace_bytes < (size_t)nsplit * M * N * sizeof(float)) return SED_EINVAL;
        dst = (float*)workspace;
        stride = (size_t)M * N;
        ldo = N;
    }
    int kchunk = ((K + nsplit - 1) / nsplit + 31) / 32 * 32;
    const dim3 grid((M + 127) / 128, (N + 127) / 128, nsplit);
    if (bf16) {
        const size_t sm = 2 * 2 * 128 * 40 * 2;
        gemm_kernel<true><<<grid, 256, sm, stream>>>(A, lda, transA, B, ldb, transB, dst, ldo, M, N, K, kchunk, stride);
    } else {
        const size_t sm = 2 * 2 * 128 * 33 * 4;
        static std::atomic<unsigned long long> attr{0};      // devices whose limit has been raised (common.h)
        int attr_err = 0;
        if (sed_first_on_device(attr, &attr_err)) {
            hipError_t e = hipFuncSetAttribute((const void*)gemm_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)sm);
            if (e != hipSuccess) return (int)e;
            sed_mark_device(attr);
        } else if (attr_err) return attr_err;
        gemm_kernel<false><<<grid, 256, sm, stream>>>(A, lda, transA, B, ldb, transB, dst, ldo, M, N, K, kchunk, stride);
    }
    SED_LAUNCH_CHECK();
    if (nsplit > 1) {
        const size_t n = (size_t)M * N;
        if (ldc != N) return SED_EUNSUPPORTED;
        splitk_reduce_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(dst, stride, nsplit, C, n);
        SED_LAUNCH_CHECK();
    }
    return SED_OK;
}
