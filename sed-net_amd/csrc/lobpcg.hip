// Device side of the batched LOBPCG behind the HPNet spectral step:
// /root/reference/src/smooth_normal_matrix.py:198 calls torch.lobpcg(A, k = 12, niter = 10) on a dense N x N matrix; here the
// operator is sparse (hpnet_sparse.hip) and the solver's own arithmetic -- the tall-skinny Gram products of the search block
// S = [X, R, P] (N x 36), the 36 x 36 Rayleigh-Ritz problems and the block updates -- runs in the kernels below, all clouds of a
// batch at once, with nothing copied to the host inside the iteration (round 2 solved the Ritz problems with LAPACK on the host:
// one D->H and one H->D copy per iteration).
//   tsgemm_tn   out[b] = A[b]^T B[b] for tall-skinny A [N, ma], B [N, mb] (ma, mb <= 36): fp64 accumulation, two-stage fixed-order
//               reduction (deterministic)
//   ritz        per cloud, one wave: the generalised symmetric problem H c = theta G c (G = S^T S, H = S^T A S) by two cyclic
//               Jacobi eigen-decompositions in fp64 (G = U w U^T -> whitening with a cut-off for numerically dependent
//               directions; then the whitened H), largest k Ritz pairs
//   resid / project / scale / update / rank1   the row-wise block operations of the iteration
#include "common.h"

namespace {

constexpr int MMAX = 36;          // 3 k, k <= 12
constexpr int LDM = MMAX + 1;     // LDS row stride of the small matrices (doubles)

// ---- out_part[b][blk][i][j] = sum over the block's rows of A[n][i] B[n][j] ------------------------------------------------
constexpr int TS_ROWS = 512;      // rows per workgroup
__global__ __launch_bounds__(256) void tsgemm_tn_partial_kernel(const float* __restrict__ A, int lda, int ma,
                                                                const float* __restrict__ Bm, int ldb, int mb, int N,
                                                                double* __restrict__ part) {
    __shared__ float ta[64][MMAX + 1], tb[64][MMAX + 1];
    const int blk = blockIdx.x, cloud = blockIdx.y, tid = threadIdx.x, nblk = gridDim.x;
    const float* Ac = A + (size_t)cloud * N * lda;
    const float* Bc = Bm + (size_t)cloud * N * ldb;
    const int ne = ma * mb;
    double acc[6] = {0, 0, 0, 0, 0, 0};                   // <= ceil(1296 / 256) outputs per thread
    const int r0 = blk * TS_ROWS, r1 = min(N, r0 + TS_ROWS);
    for (int t0 = r0; t0 < r1; t0 += 64) {
        const int nr = min(64, r1 - t0);
        __syncthreads();
        for (int e = tid; e < 64 * ma; e += 256) {
            const int r = e / ma, c = e - r * ma;
            ta[r][c] = r < nr ? Ac[(size_t)(t0 + r) * lda + c] : 0.f;
        }
        for (int e = tid; e < 64 * mb; e += 256) {
            const int r = e / mb, c = e - r * mb;
            tb[r][c] = r < nr ? Bc[(size_t)(t0 + r) * ldb + c] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int e = tid + 256 * u;
            if (e < ne) {
                const int i = e / mb, j = e - i * mb;
                double s = 0.0;
                for (int r = 0; r < 64; ++r) s += (double)ta[r][i] * (double)tb[r][j];
                acc[u] += s;
            }
        }
    }
    double* out = part + ((size_t)cloud * nblk + blk) * ne;
#pragma unroll
    for (int u = 0; u < 6; ++u) {
        const int e = tid + 256 * u;
        if (e < ne) out[e] = acc[u];
    }
}

__global__ __launch_bounds__(256) void tsgemm_reduce_kernel(const double* __restrict__ part, int nblk, int ne,
                                                            double* __restrict__ out) {
    const int cloud = blockIdx.x;
    for (int e = threadIdx.x; e < ne; e += 256) {
        double s = 0.0;
        for (int b = 0; b < nblk; ++b) s += part[((size_t)cloud * nblk + b) * ne + e];        // fixed order
        out[(size_t)cloud * ne + e] = s;
    }
}

// ---- cyclic Jacobi eigen-decomposition of a symmetric M x M matrix held in LDS (one wave; M even, compile-time) --------------
// A -> diagonal (eigenvalues), V = accumulated rotations (columns = eigenvectors). Round-robin ordering: M - 1 rounds of M / 2
// disjoint rotations per sweep; sweeps until the off-diagonal mass is at fp64 round-off (quadratic convergence: 3 - 8 sweeps;
// the Gram matrix of an orthonormalised block is nearly diagonal to begin with), at most 12.
constexpr int RITZ_THREADS = 256;  // round 5: four waves per Ritz problem (round 3 / 4: one -- 1.25 ms per call, nine calls in the LOBPCG chain)
template <int M>
__device__ void jacobi_eigh(double (*A)[LDM], double (*V)[LDM], double* cs /* [2][MMAX/2] */, int* pq /* [2][MMAX/2] */, int lane,
                            double* red /* [2][RITZ_THREADS / 64] */) {
    constexpr int NTH = RITZ_THREADS, NWV = NTH / 64;
    constexpr int HALF = M / 2, NE = HALF * M, NU = (NE + NTH - 1) / NTH;
    for (int e = lane; e < M * M; e += NTH) V[e / M][e % M] = (e / M == e % M) ? 1.0 : 0.0;
    __syncthreads();
    for (int sweep = 0; sweep < 12; ++sweep) {
        double off = 0.0, dia = 0.0;                          // convergence: off-diagonal against diagonal mass
        for (int e = lane; e < M * M; e += NTH) {
            const double a = A[e / M][e % M];
            if (e / M == e % M) dia += a * a; else off += a * a;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { off += __shfl_xor(off, o, 64); dia += __shfl_xor(dia, o, 64); }
        if ((lane & 63) == 0) { red[lane >> 6] = off; red[NWV + (lane >> 6)] = dia; }
        __syncthreads();
        off = 0.0; dia = 0.0;
#pragma unroll
        for (int w = 0; w < NWV; ++w) { off += red[w]; dia += red[NWV + w]; }      // fixed order: the same decision in every thread
        __syncthreads();
        if (off <= 1e-24 * dia) break;                        // off-diagonal / diagonal <= 1e-12: far below the fp32 the results are used in
        for (int r = 0; r < M - 1; ++r) {
            if (lane < HALF) {
                int p, q;
                if (lane == 0) { p = M - 1; q = r; }
                else { p = (r + lane) % (M - 1); q = (r - lane + (M - 1)) % (M - 1); }
                if (p > q) { const int t = p; p = q; q = t; }
                const double apq = A[p][q], app = A[p][p], aqq = A[q][q];
                double c = 1.0, s = 0.0;
                if (fabs(apq) > 1e-300) {
                    const double theta = (aqq - app) / (2.0 * apq);
                    const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                    c = 1.0 / sqrt(t * t + 1.0);
                    s = t * c;
                }
                cs[lane] = c; cs[MMAX / 2 + lane] = s; pq[lane] = p; pq[MMAX / 2 + lane] = q;
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < NU; ++u) {                     // columns p, q of A and of V
                const int e = lane + NTH * u;
                if (e < NE) {
                    const int i = e / M, j = e - i * M;
                    const double c = cs[i], s = cs[MMAX / 2 + i];
                    const int p = pq[i], q = pq[MMAX / 2 + i];
                    const double ap = A[j][p], aq = A[j][q];
                    A[j][p] = c * ap - s * aq;
                    A[j][q] = s * ap + c * aq;
                    const double vp = V[j][p], vq = V[j][q];
                    V[j][p] = c * vp - s * vq;
                    V[j][q] = s * vp + c * vq;
                }
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < NU; ++u) {                     // rows p, q of A
                const int e = lane + NTH * u;
                if (e < NE) {
                    const int i = e / M, j = e - i * M;
                    const double c = cs[i], s = cs[MMAX / 2 + i];
                    const int p = pq[i], q = pq[MMAX / 2 + i];
                    const double ap = A[p][j], aq = A[q][j];
                    A[p][j] = c * ap - s * aq;
                    A[q][j] = s * ap + c * aq;
                }
            }
            __syncthreads();
        }
    }
    __syncthreads();
}

// G, H [B][M][M] fp64 -> C [B][M][k] fp32 with C^T G C = I spanning the k largest Ritz pairs, theta [B][k] fp32
template <int M>
__global__ __launch_bounds__(RITZ_THREADS) void ritz_kernel(const double* __restrict__ G, const double* __restrict__ H, int k,
                                                  float* __restrict__ C, float* __restrict__ theta) {
    constexpr int m = M;
    __shared__ double A[MMAX][LDM], V[MMAX][LDM], Hh[MMAX][LDM], Wh[MMAX][LDM], Tm[MMAX][LDM];
    __shared__ double cs[MMAX];
    __shared__ int pq[MMAX];
    __shared__ int top[MMAX];
    __shared__ double red[2 * RITZ_THREADS / 64];
    __shared__ double dsc[MMAX];                              // (eigen route) 1 / |s_j|: the block's columns scaled to unit length
    __shared__ int drop[MMAX];                                // (eigen route) whitened directions that were cut off
    constexpr int NTH = RITZ_THREADS;
    const int cloud = blockIdx.x, lane = threadIdx.x;
    if (lane < MMAX) drop[lane] = 0;
    const double* Gc = G + (size_t)cloud * m * m;
    const double* Hc = H + (size_t)cloud * m * m;
    for (int e = lane; e < m * m; e += NTH) {
        const int i = e / m, j = e % m;
        A[i][j] = 0.5 * (Gc[i * m + j] + Gc[j * m + i]);
        Hh[i][j] = 0.5 * (Hc[i * m + j] + Hc[j * m + i]);
    }
    __syncthreads();
    // Whitening Wh with Wh^T G Wh = I. Usual case: Cholesky G = L L^T, Wh = L^-T (a 36-step column sweep + a forward substitution
    // per lane: a fraction of a Jacobi decomposition). If [X, R, P] is numerically dependent (a pivot below 1e-3 of the largest,
    // or not finite) the eigen-decomposition of G with a cut-off is used instead, like round 2's host solver.
    // Round 6 (the root cause of the intermittent abort of rounds 4-5): the block update S C runs in
    // fp32, so a whitening that scales a direction by 1 / sqrt(w) multiplies the update's rounding by the same factor. Rounds 3-5
    // accepted Cholesky pivots down to 1e-5 of the largest and, in the eigen route, directions down to 1e-10 of the largest
    // eigenvalue of the UNSCALED Gram matrix: on a cloud whose leading pairs converge within three iterations (P collapses onto
    // span [X, R]) about 4 % of the random starts amplified noise by 1e5, P grew to 1e4, the cut-off -- relative to P's 1e9 --
    // then removed the unit-length X directions themselves, zero Ritz values entered the top k, X got zero columns and the
    // eigenvector entropy divided by a zero interval: NaN spectral columns, non-unit rows into the clustering stage. Now: Cholesky
    // only for pivot ratios >= 1e-3; otherwise the columns are scaled to unit length first, directions below 1e-6 of the largest
    // eigenvalue of the SCALED Gram matrix are cut (amplification <= 1e3), and a cut direction can never be selected.
    for (int e = lane; e < m * m; e += NTH) Tm[e / m][e % m] = A[e / m][e % m];        // L is built in Tm (lower triangle)
    __syncthreads();
    bool ok = true;
    double dmin = 1e300, dmax = 0.0;
    for (int j = 0; j < m; ++j) {
        if (lane >= j && lane < m) {
            double sacc = Tm[lane][j];
            for (int l = 0; l < j; ++l) sacc -= Tm[lane][l] * Tm[j][l];
            Tm[lane][j] = sacc;                               // un-normalised column; the pivot is entry [j][j]
        }
        __syncthreads();
        const double piv = Tm[j][j];
        if (!(piv > 0.0) || !(piv < 1e300)) { ok = false; break; }
        const double dj = sqrt(piv);
        dmin = fmin(dmin, dj); dmax = fmax(dmax, dj);
        __syncthreads();
        if (lane >= j && lane < m) Tm[lane][j] = lane == j ? dj : Tm[lane][j] / dj;
        __syncthreads();
    }
    ok = ok && dmin >= 1e-3 * dmax;
    if (ok) {
        // lane c: column c of L^-1 by forward substitution (y_i = (delta_ic - sum_{l<i} L_il y_l) / L_ii), stored as Wh[c][i] = (L^-T)
        if (lane < m) {
            double y[M];
#pragma unroll
            for (int i = 0; i < M; ++i) {
                double sacc = i == lane ? 1.0 : 0.0;
#pragma unroll
                for (int l = 0; l < i; ++l) sacc -= Tm[i][l] * y[l];
                y[i] = i < lane ? 0.0 : sacc / Tm[i][i];
            }
#pragma unroll
            for (int i = 0; i < M; ++i) Wh[lane][i] = y[i];   // Wh = L^-T: Wh[c][i] = (L^-1)[i][c]
        }
        __syncthreads();
    } else {
        if (lane < m) {
            const double g = A[lane][lane];
            dsc[lane] = (g > 0.0 && g < 1e300) ? 1.0 / sqrt(g) : 0.0;       // a zero or non-finite column leaves the block
        }
        __syncthreads();
        for (int e = lane; e < m * m; e += NTH) {
            const int i = e / m, j = e % m;
            const double v = A[i][j] * dsc[i] * dsc[j];
            Tm[i][j] = (v == v && fabs(v) < 1e300) ? v : 0.0;
        }
        __syncthreads();
        for (int e = lane; e < m * m; e += NTH) A[e / m][e % m] = Tm[e / m][e % m];
        __syncthreads();
        jacobi_eigh<M>(A, V, cs, pq, lane, red);
        double wmax = 0.0;
        for (int j = 0; j < m; ++j) wmax = fmax(wmax, A[j][j]);
        for (int e = lane; e < m * m; e += NTH) {
            const int i = e / m, j = e % m;
            const double w = A[j][j];
            const bool keep = w > 1e-6 * wmax && w > 0.0;
            Wh[i][j] = keep ? dsc[i] * V[i][j] / sqrt(w) : 0.0;
            if (i == 0) drop[j] = keep ? 0 : 1;
        }
        __syncthreads();
    }
    for (int e = lane; e < m * m; e += NTH) {                  // Tm = H Wh
        const int i = e / m, j = e % m;
        double s = 0.0;
        for (int l = 0; l < m; ++l) s += Hh[i][l] * Wh[l][j];
        Tm[i][j] = s;
    }
    __syncthreads();
    for (int e = lane; e < m * m; e += NTH) {                  // A = Wh^T H Wh
        const int i = e / m, j = e % m;
        double s = 0.0;
        for (int l = 0; l < m; ++l) s += Wh[l][i] * Tm[l][j];
        A[i][j] = s;
    }
    __syncthreads();
    for (int e = lane; e < m * m; e += NTH) {                  // ... symmetrised (Tm as scratch)
        const int i = e / m, j = e % m;
        if (i < j) Tm[i][j] = 0.5 * (A[i][j] + A[j][i]);
    }
    __syncthreads();
    for (int e = lane; e < m * m; e += NTH) {
        const int i = e / m, j = e % m;
        if (i < j) { A[i][j] = Tm[i][j]; A[j][i] = Tm[i][j]; }
    }
    __syncthreads();
    jacobi_eigh<M>(A, V, cs, pq, lane, red);
    if (lane == 0) {                                          // the k largest Ritz values, descending, ties -> lowest index
        // (a cut direction is decoupled -- its row and column of the whitened problem are zero, Jacobi never rotates it -- and must
        //  not be taken for a Ritz pair with value 0)
        auto val = [&](int t) { return drop[t] ? -1e300 : A[t][t]; };
        for (int j = 0; j < m; ++j) top[j] = j;
        for (int a = 0; a < k; ++a) {
            int best = a;
            for (int b2 = a + 1; b2 < m; ++b2)
                if (val(top[b2]) > val(top[best])) best = b2;
            const int t = top[a]; top[a] = top[best]; top[best] = t;
        }
    }
    __syncthreads();
    for (int e = lane; e < m * k; e += NTH) {                  // C = Wh V[:, top]
        const int i = e / k, a = e % k;
        double s = 0.0;
        for (int l = 0; l < m; ++l) s += Wh[i][l] * V[l][top[a]];
        C[((size_t)cloud * m + i) * k + a] = (float)s;
    }
    if (lane < k) theta[(size_t)cloud * k + lane] = (float)A[top[lane]][top[lane]];
}

// ---- row-wise block operations on S, AS [B][N][ld] with columns [X (k) | R (k) | P (k)] --------------------------------------
// R = AX - X lam
__global__ __launch_bounds__(256) void lobpcg_resid_kernel(float* __restrict__ S, const float* __restrict__ AS, int ld, int k,
                                                           const float* __restrict__ lam, size_t rows, int N) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * k) return;
    const size_t r = i / k;
    const int c = (int)(i - r * k);
    const int cloud = (int)(r / N);
    S[r * ld + k + c] = AS[r * ld + c] - S[r * ld + c] * lam[(size_t)cloud * k + c];
}

// R -= X M (M = X^T R, [B][k][k] fp64) and per-block partial column sums of R^2
__global__ __launch_bounds__(256) void lobpcg_project_kernel(float* __restrict__ S, int ld, int k, const double* __restrict__ M,
                                                             int N, double* __restrict__ part /* [B][nblk][k] */) {
    __shared__ float Ms[12][12];
    __shared__ double red[256][13];
    const int blk = blockIdx.x, cloud = blockIdx.y, tid = threadIdx.x, nblk = gridDim.x;
    for (int e = tid; e < k * k; e += 256) Ms[e / k][e % k] = (float)M[(size_t)cloud * k * k + e];
    __syncthreads();
    double sq[12];
#pragma unroll
    for (int c = 0; c < 12; ++c) sq[c] = 0.0;
    for (int r = blk * 256 + tid; r < N; r += nblk * 256) {
        float* row = S + ((size_t)cloud * N + r) * ld;
        float x[12], v[12];
#pragma unroll
        for (int c = 0; c < 12; ++c) { x[c] = c < k ? row[c] : 0.f; v[c] = c < k ? row[k + c] : 0.f; }
#pragma unroll
        for (int c = 0; c < 12; ++c) {
            float s = v[c];
#pragma unroll
            for (int l = 0; l < 12; ++l) s = fmaf(-x[l], (l < k && c < k) ? Ms[l][c] : 0.f, s);
            if (c < k) { row[k + c] = s; sq[c] += (double)s * (double)s; }
        }
    }
#pragma unroll
    for (int c = 0; c < 12; ++c) red[tid][c] = sq[c];
    __syncthreads();
    if (tid < k) {
        double s = 0.0;
        for (int t = 0; t < 256; ++t) s += red[t][tid];
        part[((size_t)cloud * nblk + blk) * k + tid] = s;
    }
}

// R /= ||R||_column (norms from the partials, reduced in fixed order by every block)
__global__ __launch_bounds__(256) void lobpcg_scale_kernel(float* __restrict__ S, int ld, int k, const double* __restrict__ part,
                                                           int npart, int N) {
    __shared__ float inv[12];
    const int cloud = blockIdx.y, tid = threadIdx.x;
    if (tid < k) {
        double s = 0.0;
        for (int b = 0; b < npart; ++b) s += part[((size_t)cloud * npart + b) * k + tid];
        inv[tid] = 1.0f / fmaxf((float)sqrt(s), 1e-30f);
    }
    __syncthreads();
    const int i = blockIdx.x * 256 + tid;
    if (i >= N * k) return;
    const int r = i / k, c = i - r * k;
    S[((size_t)cloud * N + r) * ld + k + c] *= inv[c];
}

// new X = S C, AX = AS C, P = S Cp, AP = AS Cp (Cp = C with its first k rows zeroed), in place (a row only needs itself)
__global__ __launch_bounds__(256) void lobpcg_update_kernel(float* __restrict__ S, float* __restrict__ AS, int ld, int m, int k,
                                                            const float* __restrict__ C, int N) {
    __shared__ float Cs[MMAX][12];
    const int cloud = blockIdx.y, tid = threadIdx.x;
    for (int e = tid; e < MMAX * 12; e += 256) {
        const int i = e / 12, a = e % 12;
        Cs[i][a] = (i < m && a < k) ? C[((size_t)cloud * m + i) * k + a] : 0.f;
    }
    __syncthreads();
    const int r = blockIdx.x * 256 + tid;
    if (r >= N) return;
    float* s = S + ((size_t)cloud * N + r) * ld;
    float* as = AS + ((size_t)cloud * N + r) * ld;
    float sv[MMAX], av[MMAX];
#pragma unroll
    for (int i = 0; i < MMAX; ++i) { sv[i] = i < m ? s[i] : 0.f; av[i] = i < m ? as[i] : 0.f; }
#pragma unroll
    for (int a = 0; a < 12; ++a) {
        float x = 0.f, ax = 0.f, p = 0.f, ap = 0.f;
#pragma unroll
        for (int i = 0; i < MMAX; ++i) {
            const float c = Cs[i][a];
            x = fmaf(sv[i], c, x);
            ax = fmaf(av[i], c, ax);
            if (i >= 12) { p = fmaf(sv[i], c, p); ap = fmaf(av[i], c, ap); }
        }
        if (a < k) { s[a] = x; as[a] = ax; s[2 * k + a] = p; as[2 * k + a] = ap; }
    }
}

// Y[n][c] += alpha d[n] t[c]  (the rank-one background of the affinity operator; t = d^T X, [B][k] fp64)
__global__ __launch_bounds__(256) void rank1_add_kernel(float* __restrict__ Y, int ldy, int k, const float* __restrict__ d,
                                                        const double* __restrict__ t, float alpha, int N) {
    const int cloud = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N * k) return;
    const int r = i / k, c = i - r * k;
    Y[((size_t)cloud * N + r) * ldy + c] += alpha * d[(size_t)cloud * N + r] * (float)t[(size_t)cloud * k + c];
}

}  // namespace

extern "C" size_t sed_tsgemm_tn_workspace_bytes(int B, int N, int ma, int mb) {
    if (B <= 0 || N <= 0 || ma <= 0 || mb <= 0) return 0;
    return (size_t)B * ((N + TS_ROWS - 1) / TS_ROWS) * ma * mb * sizeof(double);
}

// out [B][ma][mb] (fp64) = A^T B for tall-skinny A [B][N][lda] (first ma columns) and Bm [B][N][ldb] (first mb columns)
extern "C" int sed_tsgemm_tn_f64(int B, int N, int ma, int mb, const float* A, int lda, const float* Bm, int ldb, double* out,
                                 void* ws, size_t ws_bytes, hipStream_t stream) {
    if (B <= 0 || N <= 0 || !A || !Bm || !out || !ws || lda < ma || ldb < mb) return SED_EINVAL;
    if (ma < 1 || mb < 1 || ma > MMAX || mb > MMAX) return SED_EUNSUPPORTED;
    if (ws_bytes < sed_tsgemm_tn_workspace_bytes(B, N, ma, mb)) return SED_EINVAL;
    const int nblk = (N + TS_ROWS - 1) / TS_ROWS;
    tsgemm_tn_partial_kernel<<<dim3(nblk, B), 256, 0, stream>>>(A, lda, ma, Bm, ldb, mb, N, (double*)ws);
    tsgemm_reduce_kernel<<<B, 256, 0, stream>>>((const double*)ws, nblk, ma * mb, out);
    SED_LAUNCH_CHECK();
    return SED_OK;
}

// Rayleigh-Ritz: G = S^T S, H = S^T A S [B][m][m] fp64 (m in {12, 24, 36}) -> C [B][m][k] fp32 (C^T G C = I, the k largest Ritz
// pairs), theta [B][k] fp32, descending.  src/smooth_normal_matrix.py:198 (inside torch.lobpcg)
extern "C" int sed_ritz_f64(int B, int m, int k, const double* G, const double* H, float* C, float* theta, hipStream_t stream) {
    if (B <= 0 || !G || !H || !C || !theta) return SED_EINVAL;
    if ((m != 12 && m != 24 && m != 36) || k < 1 || k > 12) return SED_EUNSUPPORTED;
    if (m == 12) ritz_kernel<12><<<B, RITZ_THREADS, 0, stream>>>(G, H, k, C, theta);
    else if (m == 24) ritz_kernel<24><<<B, RITZ_THREADS, 0, stream>>>(G, H, k, C, theta);
    else ritz_kernel<36><<<B, RITZ_THREADS, 0, stream>>>(G, H, k, C, theta);
    SED_LAUNCH_CHECK();
    return SED_OK;
}

// One LOBPCG residual step on the search block S [B][N][ld] = [X | R | P] (k columns each), AS likewise:
//   R = AX - X lam;  R -= X (X^T R);  R /= ||R|| per column.  ws: sed_lobpcg_workspace_bytes(B, N, k).
extern "C" size_t sed_lobpcg_workspace_bytes(int B, int N, int k) {
    if (B <= 0 || N <= 0 || k <= 0) return 0;
    const size_t nblk = (N + TS_ROWS - 1) / TS_ROWS;
    return (size_t)B * nblk * k * k * sizeof(double) + (size_t)B * k * k * sizeof(double) + (size_t)B * 64 * k * sizeof(double) + 256;
}

extern "C" int sed_lobpcg_residual_f32(int B, int N, int k, float* S, const float* AS, int ld, const float* lam, void* ws,
                                       size_t ws_bytes, hipStream_t stream) {
    if (B <= 0 || N <= 0 || !S || !AS || !lam || !ws || ld < 2 * k) return SED_EINVAL;
    if (k < 1 || k > 12) return SED_EUNSUPPORTED;
    if (ws_bytes < sed_lobpcg_workspace_bytes(B, N, k)) return SED_EINVAL;
    const size_t rows = (size_t)B * N;
    const int nblk = (N + TS_ROWS - 1) / TS_ROWS;
    double* part = (double*)ws;
    double* M = part + (size_t)B * nblk * k * k;
    double* sq = M + (size_t)B * k * k;
    lobpcg_resid_kernel<<<(unsigned)((rows * k + 255) / 256), 256, 0, stream>>>(S, AS, ld, k, lam, rows, N);
    tsgemm_tn_partial_kernel<<<dim3(nblk, B), 256, 0, stream>>>(S, ld, k, S + k, ld, k, N, part);          // X^T R
    tsgemm_reduce_kernel<<<B, 256, 0, stream>>>(part, nblk, k * k, M);
    const int npart = 64 < (N + 255) / 256 ? 64 : (N + 255) / 256;
    lobpcg_project_kernel<<<dim3(npart, B), 256, 0, stream>>>(S, ld, k, M, N, sq);
    lobpcg_scale_kernel<<<dim3((N * k + 255) / 256, B), 256, 0, stream>>>(S, ld, k, sq, npart, N);
    SED_LAUNCH_CHECK();
    return SED_OK;
}

// X <- S C, AX <- AS C, P <- S Cp, AP <- AS Cp in place on S, AS [B][N][ld] (m = 2 k or 3 k columns in use; ld >= 3 k)
extern "C" int sed_lobpcg_update_f32(int B, int N, int m, int k, float* S, float* AS, int ld, const float* C, hipStream_t stream) {
    if (B <= 0 || N <= 0 || !S || !AS || !C || ld < 3 * k || m > ld) return SED_EINVAL;
    if (k != 12 || (m != 24 && m != 36 && m != 12)) return SED_EUNSUPPORTED;
    lobpcg_update_kernel<<<dim3((N + 255) / 256, B), 256, 0, stream>>>(S, AS, ld, m, k, C, N);
    SED_LAUNCH_CHECK();
    return SED_OK;
}

// Y [B][N][ldy] (first k columns) += alpha d t^T, d [B][N], t [B][k] fp64
extern "C" int sed_rank1_add_f32(int B, int N, int k, float* Y, int ldy, const float* d, const double* t, float alpha,
                                 hipStream_t stream) {
    if (B <= 0 || N <= 0 || k <= 0 || !Y || !d || !t || ldy < k) return SED_EINVAL;
    rank1_add_kernel<<<dim3((N * k + 255) / 256, B), 256, 0, stream>>>(Y, ldy, k, d, t, alpha, N);
    SED_LAUNCH_CHECK();
    return SED_OK;
}
