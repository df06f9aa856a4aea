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

// ------------------------------------------------------------------------------------------------------------
// The CSR of M = 1/2 (S + S^T) built on the device (round 4; VERDICT r3 item 3: the torch build sorted 1 M (row, col) keys per
// cloud with rocPRIM's merge sort and went through scatter_add_ / gather / cumsum -- 20 % of the HPNet stage's kernel time).
// /root/reference/src/smooth_normal_matrix.py:42-92: s_ij = exp(-acos(clamp(n_i . n_j, +-0.99))^2 / (2 sigma^2)) on the
// farthest-`knn` pattern nn [B,N,knn], zeros replaced by the dense matrix's 1e-12 background, d = rowsum^-1/2 with the background
// of the other N - knn columns counted in. Row c of M holds
//     its knn FORWARD entries   (c, nn[c][j])  in the graph's own order (j = 0 .. knn - 1), then
//     its TRANSPOSED entries    (c, p) for every p with c in nn[p], p ascending,
// each with value 1/2 (s - 1e-12) d_row d_col (a pair that is in both lists appears twice, like in the torch build; the product
// kernel adds them). The transposed half needs no sort: a bitmap T[c][p] (N x N bits per cloud, set with atomicOr -- order-free) is
// walked once per row for word prefix sums (a wave per row); an edge then finds its position as a rank in that row -- one thread per edge. Any fixed order is CORRECT (the product
// sums a row's entries in storage order, fp32: another order moves the eigenvectors at rounding level, like another lobpcg
// seed); this one is deterministic. in-degree of the farthest-50 graph is very uneven (periphery points are everybody's farthest
// neighbour): a row can hold thousands of transposed entries -- no thread ever walks them.
namespace {

__global__ __launch_bounds__(256) void aff_rows_kernel(const float* __restrict__ nrm, const int* __restrict__ nn, int N, int knn,
                                                       float inv2s2, float* __restrict__ seff, float* __restrict__ d,
                                                       unsigned* __restrict__ bitmap, int W) {
    const int cloud = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const float* nc = nrm + (size_t)cloud * N * 3;
    const int* nb = nn + ((size_t)cloud * N + i) * knn;
    float* so = seff + ((size_t)cloud * N + i) * knn;
    const float nx = nc[3 * i], ny = nc[3 * i + 1], nz = nc[3 * i + 2];
    float rowsum = 0.f;
    for (int j = 0; j < knn; ++j) {
        const int c = nb[j];
        const float dot = fminf(fmaxf(nx * nc[3 * c] + ny * nc[3 * c + 1] + nz * nc[3 * c + 2], -0.99f), 0.99f);      // :70
        const float a = acosf(dot);
        float s = expf(-a * a * inv2s2);                                                                          // :71
        if (s == 0.f) s = 1e-12f;                                                                                 // :77-80
        so[j] = s;
        rowsum += s;
        atomicOr(bitmap + ((size_t)cloud * N + c) * W + (i >> 5), 1u << (i & 31));
    }
    rowsum += (float)(N - knn) * 1e-12f;
    d[(size_t)cloud * N + i] = 1.0f / sqrtf(rowsum);
}

// one wave per row of the transposed pattern: exclusive prefix sums of the popcounts of its bitmap words (uint16: N <= 65 535) and
// the row's in-degree
__global__ __launch_bounds__(256) void aff_prefix_kernel(const unsigned* __restrict__ bitmap, int N, int W,
                                                         unsigned short* __restrict__ wprefix, int* __restrict__ indeg) {
    const int cloud = blockIdx.y, row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= N) return;
    const unsigned* bm = bitmap + ((size_t)cloud * N + row) * W;
    unsigned short* wp = wprefix + ((size_t)cloud * N + row) * W;
    int run = 0;
    for (int w0 = 0; w0 < W; w0 += 64) {
        const int w = w0 + lane;
        const int cnt = w < W ? __builtin_popcount(bm[w]) : 0;
        int incl = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int v = __shfl_up(incl, off, 64);
            if (lane >= off) incl += v;
        }
        if (w < W) wp[w] = (unsigned short)(run + incl - cnt);
        run += __shfl(incl, 63, 64);
    }
    if (lane == 0) indeg[(size_t)cloud * N + row] = run;
}

__global__ __launch_bounds__(1024) void aff_rowptr_kernel(const int* __restrict__ indeg, int N, int knn, int* __restrict__ rowptr) {
    __shared__ int part[1024];
    const int cloud = blockIdx.x, tid = threadIdx.x;
    const int* in = indeg + (size_t)cloud * N;
    int* rp = rowptr + (size_t)cloud * (N + 1);
    const int per = (N + 1023) / 1024, lo = tid * per, hi = min(N, lo + per);
    int s = 0;
    for (int r = lo; r < hi; ++r) s += knn + in[r];
    part[tid] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {            // inclusive scan of the 1024 partial sums
        const int v = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int acc = tid ? part[tid - 1] : 0;
    if (tid == 0) rp[0] = 0;
    for (int r = lo; r < hi; ++r) { acc += knn + in[r]; rp[r + 1] = acc; }
}

// one thread per graph edge (p, j), c = nn[p][j]: the forward entry of row p and the transposed entry of row c, whose position
// among row c's transposed entries is the rank of p among the set bits of row c's bitmap
__global__ __launch_bounds__(256) void aff_fill_kernel(const int* __restrict__ nn, const float* __restrict__ seff,
                                                       const float* __restrict__ d, const unsigned* __restrict__ bitmap,
                                                       const unsigned short* __restrict__ wprefix, const int* __restrict__ rowptr,
                                                       int N, int knn, int W, size_t nnz_stride, int* __restrict__ col,
                                                       float* __restrict__ val) {
    const int cloud = blockIdx.y;
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (size_t)N * knn) return;
    const int p = (int)(e / knn), j = (int)(e - (size_t)p * knn);
    const int c = nn[(size_t)cloud * N * knn + e];
    const float* dc = d + (size_t)cloud * N;
    const int* rp = rowptr + (size_t)cloud * (N + 1);
    int* cc = col + (size_t)cloud * nnz_stride;
    float* vv = val + (size_t)cloud * nnz_stride;
    const float v = 0.5f * (seff[(size_t)cloud * N * knn + e] - 1e-12f) * dc[p] * dc[c];
    cc[rp[p] + j] = c;
    vv[rp[p] + j] = v;
    const size_t wi = ((size_t)cloud * N + c) * W + (p >> 5);
    const int rank = wprefix[wi] + __builtin_popcount(bitmap[wi] & ((1u << (p & 31)) - 1u));
    cc[rp[c] + knn + rank] = p;
    vv[rp[c] + knn + rank] = v;
}

}  // namespace

// workspace: s [B,N,knn] f32 | in-degrees [B,N] i32 | bitmap [B,N,ceil(N/32)] u32 | word prefix sums [B,N,ceil(N/32)] u16
extern "C" size_t sed_hpnet_affinity_csr_workspace_bytes(int B, int N, int knn) {
    if (B <= 0 || N <= 0 || knn <= 0) return 0;
    const size_t W = (size_t)(N + 31) / 32;
    return ((size_t)B * N * knn * sizeof(float) + 255) / 256 * 256 + ((size_t)B * N * sizeof(int) + 255) / 256 * 256 +
           ((size_t)B * N * W * sizeof(unsigned) + 255) / 256 * 256 + (size_t)B * N * W * sizeof(unsigned short);
}

// normals [B,N,3] (unit), nn [B,N,knn] (the farthest-knn graph, sed_knn_fused_far_f32) -> rowptr [B,N+1], col / val [B, 2 knn N],
// d [B,N]: the operator of src/smooth_normal_matrix.py:42-92 as sed_csr_spmm_f32 takes it (A_sym = M + 1e-12 d d^T). N <= 65 535.
extern "C" int sed_hpnet_affinity_csr_f32(int B, int N, int knn, float sigma, const float* normals, const int* nn, int* rowptr,
                                          int* col, float* val, float* d, void* ws, size_t ws_bytes, hipStream_t stream) {
    if (B <= 0 || N <= 0 || knn <= 0 || knn > N || !(sigma > 0.f) || !normals || !nn || !rowptr || !col || !val || !d || !ws)
        return SED_EINVAL;
    if (N > 65535) return SED_EUNSUPPORTED;
    if (ws_bytes < sed_hpnet_affinity_csr_workspace_bytes(B, N, knn)) return SED_EINVAL;
    const int W = (N + 31) / 32;
    float* seff = (float*)ws;
    int* indeg = (int*)((uint8_t*)ws + ((size_t)B * N * knn * sizeof(float) + 255) / 256 * 256);
    unsigned* bitmap = (unsigned*)((uint8_t*)indeg + ((size_t)B * N * sizeof(int) + 255) / 256 * 256);
    unsigned short* wprefix = (unsigned short*)((uint8_t*)bitmap + ((size_t)B * N * W * sizeof(unsigned) + 255) / 256 * 256);
    hipError_t e = hipMemsetAsync(bitmap, 0, (size_t)B * N * W * sizeof(unsigned), stream);
    if (e != hipSuccess) return (int)e;
    aff_rows_kernel<<<dim3((N + 255) / 256, B), 256, 0, stream>>>(normals, nn, N, knn, 1.0f / (2.0f * sigma * sigma), seff, d, bitmap, W);
    aff_prefix_kernel<<<dim3((N + 3) / 4, B), 256, 0, stream>>>(bitmap, N, W, wprefix, indeg);
    aff_rowptr_kernel<<<B, 1024, 0, stream>>>(indeg, N, knn, rowptr);
    aff_fill_kernel<<<dim3((unsigned)(((size_t)N * knn + 255) / 256), B), 256, 0, stream>>>(nn, seff, d, bitmap, wprefix, rowptr, N, knn, W,
                                                                                        (size_t)2 * knn * N, col, val);
    SED_LAUNCH_CHECK();
    return SED_OK;
}
