// Batched CSR x dense product for the HPNet spectral step.
//
// /root/reference/src/smooth_normal_matrix.py:42-92 builds a dense N x N affinity (400 MB per 10 000-point cloud) that
// holds 50 non-zeros per row plus a constant 1e-12 background, symmetrises it with three dense matmuls and hands it to
// torch.lobpcg (:198). The matrix the eigen-solver sees is
//     A_sym = 1/2 (S + S^T) + 1e-12 d d^T ,   S_ij = (s_ij - 1e-12) d_i d_j on the 50-neighbour pattern ,  d = rowsum^-1/2
// i.e. a sparse matrix with 100 entries per row on average plus a rank-one term. This kernel applies the sparse part to a
// block of vectors (Y = M X, M in CSR, X [B,N,ncol], ncol <= 16); the rank-one term is two tiny dense products in the
// caller (src/smooth_normal_matrix.py). One wave per row, lanes stride the row's entries (the in-degree of the
// "farthest-50" graph is very uneven: periphery points are everybody's farthest neighbour), fixed-order wave reduction:
// deterministic, no atomics. Bound: HBM/L2 gather of ncol floats per entry (100 N ncol 4 B = 48 MB per cloud per product).
#include "common.h"

namespace {

template <int NC>
__global__ __launch_bounds__(256) void csr_spmm_kernel(const int* __restrict__ rowptr, const int* __restrict__ col,
                                                       const float* __restrict__ val, const float* __restrict__ X,
                                                       int ldx, float* __restrict__ Y, int ldy, int N, size_t nnz_stride) {
    const int cloud = blockIdx.y;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= N) return;
    const int* rp = rowptr + (size_t)cloud * (N + 1);
    const int* cc = col + (size_t)cloud * nnz_stride;
    const float* vv = val + (size_t)cloud * nnz_stride;
    const float* Xc = X + (size_t)cloud * N * ldx;
    float acc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] = 0.f;
    for (int e = rp[row] + lane; e < rp[row + 1]; e += 64) {
        const float v = vv[e];
        const float* x = Xc + (size_t)cc[e] * ldx;
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[c] = fmaf(v, x[c], acc[c]);
    }
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc[c] += __shfl_xor(acc[c], off, 64);
    if (lane == 0) {
        float* y = Y + ((size_t)cloud * N + row) * ldy;
#pragma unroll
        for (int c = 0; c < NC; ++c) y[c] = acc[c];
    }
}

}  // namespace

// Y [B,N,ldy] (first ncol columns) = M X for B matrices in CSR with a common nnz capacity (`nnz_stride` entries per cloud:
// rowptr [B,N+1] indexes into col / val [B,nnz_stride]); X [B,N,ldx] (first ncol columns); ncol in {4, 8, 12, 16, 24, 36}.
extern "C" int sed_csr_spmm_f32(int B, int N, int ncol, size_t nnz_stride, const int* rowptr, const int* col,
                                const float* val, const float* X, int ldx, float* Y, int ldy, hipStream_t stream) {
    if (B <= 0 || N <= 0 || !rowptr || !col || !val || !X || !Y || ldx < ncol || ldy < ncol) return SED_EINVAL;
    const dim3 grid((N + 3) / 4, B);
    switch (ncol) {
        case 4: csr_spmm_kernel<4><<<grid, 256, 0, stream>>>(rowptr, col, val, X, ldx, Y, ldy, N, nnz_stride); break;
        case 8: csr_spmm_kernel<8><<<grid, 256, 0, stream>>>(rowptr, col, val, X, ldx, Y, ldy, N, nnz_stride); break;
        case 12: csr_spmm_kernel<12><<<grid, 256, 0, stream>>>(rowptr, col, val, X, ldx, Y, ldy, N, nnz_stride); break;
        case 16: csr_spmm_kernel<16><<<grid, 256, 0, stream>>>(rowptr, col, val, X, ldx, Y, ldy, N, nnz_stride); break;
        case 24: csr_spmm_kernel<24><<<grid, 256, 0, stream>>>(rowptr, col, val, X, ldx, Y, ldy, N, nnz_stride); break;
        case 36: csr_spmm_kernel<36><<<grid, 256, 0, stream>>>(rowptr, col, val, X, ldx, Y, ldy, N, nnz_stride); break;
        default: return SED_EUNSUPPORTED;
    }
    SED_LAUNCH_CHECK();
    return SED_OK;
}
