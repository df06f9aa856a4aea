// Batched weighted least-squares primitive fits (plane / sphere / cylinder / cone) and closed-form residuals:
// one workgroup per (cloud, segment), no host round trips.
//
// Replaces the per-segment Python loop /root/reference/src/primitive_forward.py:929-1051 (fit_one_shape_torch)
// -> src/fitting_optimization.py:160-245 -> Fit.fit_{plane,sphere,cylinder,cone}_torch (:712-847) with their
// thin SVD / QR / matrix_rank calls (src/fitting_utils.py:36-85, :436-445) and one np.linalg.cond D->H sync per
// segment, and /root/reference/src/primitives.py:89-195 (distance_from_*).
//
// Every fit is a few streaming passes over the segment's points (HBM/L2-bound: 28 B per point per pass)
// feeding 3x3 problems:  per-point terms are formed in fp32 exactly as the reference forms them (same
// operation order), the segmented reductions are fp64 in a fixed order (thread-strided partials -> wave
// shuffles -> 4 waves), and the 3x3 eigen / linear solves run in fp64 on thread 0 (Jacobi), reproducing the
// reference's branch structure: rank test with torch.matrix_rank's tolerance, QR branch == normal equations,
// ridge branch with best_lambda, cone condition-number bail-out.
#include "common.h"

namespace {

constexpr float kEPS = 1.1920928955078125e-07f;       // np.finfo(np.float32).eps, primitive_forward.py:23
enum { T_PLANE = 1, T_CONE = 3, T_CYLINDER = 4, T_SPHERE = 5 };   // primitive_forward.py:1007-1024

// ---------------------------------------------------------------------------------------------------------
// block reduction of K doubles (fixed order). Result valid in every thread.
template <int K>
__device__ void block_sum(double (&v)[K], double* sh /*[4][K]*/) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < K; ++i)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v[i] += __shfl_xor(v[i], off, 64);
    __syncthreads();                     // previous users of sh are done
    if (lane == 0)
#pragma unroll
        for (int i = 0; i < K; ++i) sh[wave * K + i] = v[i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < K; ++i) v[i] = ((sh[i] + sh[K + i]) + sh[2 * K + i]) + sh[3 * K + i];
}

// ---------------------------------------------------------------------------------------------------------
// 3x3 symmetric eigen decomposition (cyclic Jacobi, fp64). m = {xx,xy,xz,yy,yz,zz}. Eigenvalues ascending in w,
// eigenvectors in the columns of V.
__device__ void eig3(const double m[6], double w[3], double V[3][3]) {
    double A[3][3] = {{m[0], m[1], m[2]}, {m[1], m[3], m[4]}, {m[2], m[4], m[5]}};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) V[i][j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 32; ++sweep) {
        const double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
        const double diag = A[0][0] * A[0][0] + A[1][1] * A[1][1] + A[2][2] * A[2][2];
        if (off <= 1e-40 * diag || off == 0.0) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (A[p][q] == 0.0) continue;
                const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; ++k) {            // A <- A J
                    const double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - s * akq;
                    A[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; ++k) {            // A <- J^T A
                    const double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = c * apk - s * aqk;
                    A[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; ++k) {
                    const double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq;
                    V[k][q] = s * vkp + c * vkq;
                }
            }
    }
    w[0] = A[0][0]; w[1] = A[1][1]; w[2] = A[2][2];
    for (int i = 0; i < 2; ++i)                           // sort ascending
        for (int j = 0; j < 2 - i; ++j)
            if (w[j] > w[j + 1]) {
                const double t = w[j]; w[j] = w[j + 1]; w[j + 1] = t;
                for (int k = 0; k < 3; ++k) { const double u = V[k][j]; V[k][j] = V[k][j + 1]; V[k][j + 1] = u; }
            }
}

__device__ bool solve3(const double m[6], const double y[3], double x[3]) {
    const double a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5];
    const double c00 = d * f - e * e, c01 = c * e - b * f, c02 = b * e - c * d;
    const double det = a * c00 + b * c01 + c * c02;
    if (det == 0.0) { x[0] = x[1] = x[2] = 0.0; return false; }
    const double c11 = a * f - c * c, c12 = b * c - a * e, c22 = a * d - b * b;
    x[0] = (c00 * y[0] + c01 * y[1] + c02 * y[2]) / det;
    x[1] = (c01 * y[0] + c11 * y[1] + c12 * y[2]) / det;
    x[2] = (c02 * y[0] + c12 * y[1] + c22 * y[2]) / det;
    return true;
}

__device__ void symsq3(const double m[6], double out[6]) {       // out = M^T M for symmetric M
    const double M[3][3] = {{m[0], m[1], m[2]}, {m[1], m[3], m[4]}, {m[2], m[4], m[5]}};
    int t = 0;
    for (int i = 0; i < 3; ++i)
        for (int j = i; j < 3; ++j) out[t++] = M[0][i] * M[0][j] + M[1][i] * M[1][j] + M[2][i] * M[2][j];
}

// rank of a matrix whose singular values are sv[0..2] (any order), torch.matrix_rank default tolerance:
// sigma_max * max(rows, cols) * eps(fp32)   (fitting_utils.py:48, :79)
__device__ int rank_from_sv(const double sv[3], int maxdim) {
    const double smax = fmax(sv[0], fmax(sv[1], sv[2]));
    const double tol = smax * (double)maxdim * (double)kEPS;
    return (sv[0] > tol) + (sv[1] > tol) + (sv[2] > tol);
}

// LeastSquares.lstsq for an n x 3 system given G = A^T A and g = A^T Y (fitting_utils.py:36-65).
__device__ void lstsq3(const double G[6], const double g[3], int n_rows, double x[3]) {
    double w[3], V[3][3];
    eig3(G, w, V);
    double sv[3] = {sqrt(fmax(w[0], 0.0)), sqrt(fmax(w[1], 0.0)), sqrt(fmax(w[2], 0.0))};
    if (rank_from_sv(sv, n_rows > 3 ? n_rows : 3) == 3) {      // full column rank: x = R^-1 Q^T Y == G^-1 g
        solve3(G, g, x);
        return;
    }
    // rank deficient: ridge on the normal equations, recursively (square systems from here on)
    double M[6], y[3];
    for (int i = 0; i < 6; ++i) M[i] = G[i];       // AtA
    for (int i = 0; i < 3; ++i) y[i] = g[i];       // A^T Y
    for (int depth = 0; depth < 4; ++depth) {
        // best_lambda(AtA): smallest of 1e-6 * 10^i making AtA + lambda I full rank (fitting_utils.py:68-85)
        double mw[3], mV[3][3];
        eig3(M, mw, mV);
        double lamb = 1e-6;
        for (int i = 0; i < 7; ++i) {
            double s2[3] = {fabs(mw[0] + lamb), fabs(mw[1] + lamb), fabs(mw[2] + lamb)};
            if (rank_from_sv(s2, 3) == 3) break;
            lamb *= 10.0;
        }
        double Ad[6] = {M[0] + lamb, M[1], M[2], M[3] + lamb, M[4], M[5] + lamb};
        double s3[3] = {fabs(mw[0] + lamb), fabs(mw[1] + lamb), fabs(mw[2] + lamb)};
        if (rank_from_sv(s3, 3) == 3 || depth == 3) {
            solve3(Ad, y, x);
            return;
        }
        // still deficient: lstsq(A_dash, Y_dash) recurses with AtA = A_dash^T A_dash, Y = A_dash^T Y_dash
        const double Ad3[3][3] = {{Ad[0], Ad[1], Ad[2]}, {Ad[1], Ad[3], Ad[4]}, {Ad[2], Ad[4], Ad[5]}};
        double y2[3];
        for (int i = 0; i < 3; ++i) y2[i] = Ad3[0][i] * y[0] + Ad3[1][i] * y[1] + Ad3[2][i] * y[2];
        symsq3(Ad, M);
        for (int i = 0; i < 3; ++i) y[i] = y2[i];
    }
}

// ---------------------------------------------------------------------------------------------------------
struct SegView {
    const float* pts;      // [N,3]
    const float* nrm;      // [N,3]
    const int* labels;     // [N] or null (every point belongs)
    const float* w;        // per-point weight [N*wstride] or null (1)
    int wstride;           // element stride between consecutive points' weights
    int N, seg;
    float wadd;            // EPS added to every weight (primitive_forward.py:947,963)
    __device__ bool in(int i) const { return labels ? labels[i] == seg : true; }
    __device__ float weight(int i) const { return (w ? w[(size_t)i * wstride] : 1.0f) + wadd; }
};

// sphere fit on q_i = p_i (axis == nullptr) or q_i = p_i - (p_i . a) a (cylinder), primitive_forward.py:750-773
__device__ void fit_sphere_block(const SegView& v, const float* axis, int count, double* sh, float* sb,
                                 float centre[3], float* radius) {
    const int tid = threadIdx.x;
    auto project = [&](int i, float q[3]) {
        q[0] = v.pts[3 * i]; q[1] = v.pts[3 * i + 1]; q[2] = v.pts[3 * i + 2];
        if (axis) {
            const float t = fmaf(q[2], axis[2], fmaf(q[1], axis[1], __fmul_rn(q[0], axis[0])));   // points @ a
            q[0] = __fsub_rn(q[0], __fmul_rn(t, axis[0]));
            q[1] = __fsub_rn(q[1], __fmul_rn(t, axis[1]));
            q[2] = __fsub_rn(q[2], __fmul_rn(t, axis[2]));
        }
    };
    // pass 1: sum w, sum w q, sum w |q|^2
    double a1[5] = {0, 0, 0, 0, 0};
    for (int i = tid; i < v.N; i += 256)
        if (v.in(i)) {
            float q[3];
            project(i, q);
            const float w = v.weight(i);
            const float qq = __fadd_rn(__fadd_rn(__fmul_rn(q[0], q[0]), __fmul_rn(q[1], q[1])), __fmul_rn(q[2], q[2]));
            a1[0] += (double)w;
            a1[1] += (double)__fmul_rn(q[0], w);
            a1[2] += (double)__fmul_rn(q[1], w);
            a1[3] += (double)__fmul_rn(q[2], w);
            a1[4] += (double)__fmul_rn(w, qq);
        }
    block_sum<5>(a1, sh);
    const float sum_w = __fadd_rn((float)a1[0], kEPS);
    const float mq[3] = {(float)a1[1] / sum_w, (float)a1[2] / sum_w, (float)a1[3] / sum_w};
    const float normalization = (float)a1[4] / sum_w;
    // pass 2: Gram of A = w * 2(-q + mean), Y = w * (w |q|^2 - normalization)      (:754-763)
    double a2[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = tid; i < v.N; i += 256)
        if (v.in(i)) {
            float q[3];
            project(i, q);
            const float w = v.weight(i);
            const float qq = __fadd_rn(__fadd_rn(__fmul_rn(q[0], q[0]), __fmul_rn(q[1], q[1])), __fmul_rn(q[2], q[2]));
            float A[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) A[c] = __fmul_rn(w, __fmul_rn(2.0f, __fadd_rn(-q[c], mq[c])));
            const float Y = __fmul_rn(w, __fsub_rn(__fmul_rn(w, qq), normalization));
            a2[0] += (double)A[0] * A[0]; a2[1] += (double)A[0] * A[1]; a2[2] += (double)A[0] * A[2];
            a2[3] += (double)A[1] * A[1]; a2[4] += (double)A[1] * A[2]; a2[5] += (double)A[2] * A[2];
            a2[6] += (double)A[0] * Y; a2[7] += (double)A[1] * Y; a2[8] += (double)A[2] * Y;
        }
    block_sum<9>(a2, sh);
    if (tid == 0) {
        double x[3];
        lstsq3(a2, a2 + 6, count, x);
        sb[0] = (float)(-x[0]); sb[1] = (float)(-x[1]); sb[2] = (float)(-x[2]);      // center = -lstsq(A, Y)
    }
    __syncthreads();
    centre[0] = sb[0]; centre[1] = sb[1]; centre[2] = sb[2];
    // pass 3: radius^2 = sum w |q - c|^2 / sum_w                                       (:770-772)
    double a3[1] = {0};
    for (int i = tid; i < v.N; i += 256)
        if (v.in(i)) {
            float q[3];
            project(i, q);
            const float d0 = __fsub_rn(q[0], centre[0]), d1 = __fsub_rn(q[1], centre[1]), d2 = __fsub_rn(q[2], centre[2]);
            const float dd = __fadd_rn(__fadd_rn(__fmul_rn(d0, d0), __fmul_rn(d1, d1)), __fmul_rn(d2, d2));
            a3[0] += (double)__fmul_rn(v.weight(i), dd);
        }
    block_sum<1>(a3, sh);
    float r2 = (float)a3[0] / sum_w;
    r2 = fmaxf(r2, 1e-3f);
    *radius = sqrtf(fmaxf(r2, 1e-5f));
}

// smallest right-singular vector of rows w_i * (u_i - shift): eigenvector of sum w^2 (u - s)(u - s)^T
__device__ void smallest_dir_block(const SegView& v, const float* vecs, const float* shift, double* sh, float* sb,
                                   float dir[3]) {
    const int tid = threadIdx.x;
    double a[6] = {0, 0, 0, 0, 0, 0};
    for (int i = tid; i < v.N; i += 256)
        if (v.in(i)) {
            const float w = v.weight(i);
            float u[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float t = vecs[3 * i + c];
                if (shift) t = __fsub_rn(t, shift[c]);
                u[c] = __fmul_rn(w, t);
            }
            a[0] += (double)u[0] * u[0]; a[1] += (double)u[0] * u[1]; a[2] += (double)u[0] * u[2];
            a[3] += (double)u[1] * u[1]; a[4] += (double)u[1] * u[2]; a[5] += (double)u[2] * u[2];
        }
    block_sum<6>(a, sh);
    if (tid == 0) {
        double w[3], V[3][3];
        eig3(a, w, V);
        sb[0] = (float)V[0][0]; sb[1] = (float)V[1][0]; sb[2] = (float)V[2][0];
    }
    __syncthreads();
    dir[0] = sb[0]; dir[1] = sb[1]; dir[2] = sb[2];
    __syncthreads();
}

__global__ __launch_bounds__(256) void fit_segments_kernel(const float* __restrict__ points,
                                                           const float* __restrict__ normals,
                                                           const int* __restrict__ labels,
                                                           const int* __restrict__ seg_type,
                                                           const float* __restrict__ weights, int wmode,
                                                           float weight_eps, int N, int S,
                                                           int min_points, float* __restrict__ params,
                                                           int* __restrict__ valid) {
    __shared__ double sh[4 * 12];
    __shared__ float sb[8];
    const int seg = blockIdx.x, cloud = blockIdx.y, tid = threadIdx.x;
    SegView v;
    v.pts = points + (size_t)cloud * N * 3;
    v.nrm = normals + (size_t)cloud * N * 3;
    v.labels = labels ? labels + (size_t)cloud * N : nullptr;
    v.N = N; v.seg = seg; v.wadd = weight_eps;
    // wmode 0: unit weights; 1: per-point weight [B,N] of the point's own segment; 2: soft weights [B,N,S]
    v.w = wmode == 0 ? nullptr : (wmode == 1 ? weights + (size_t)cloud * N : weights + (size_t)cloud * N * S + seg);
    v.wstride = wmode == 2 ? S : 1;
    float* out = params + ((size_t)cloud * S + seg) * 8;
    const int type = seg_type[(size_t)cloud * S + seg];

    double cnt[1] = {0};
    for (int i = tid; i < N; i += 256) cnt[0] += v.in(i) ? 1.0 : 0.0;
    block_sum<1>(cnt, sh);
    const int count = (int)cnt[0];
    const bool known = type == T_PLANE || type == T_CONE || type == T_CYLINDER || type == T_SPHERE;
    if (count < min_points || !known) {          // primitive_forward.py:974-978 (block-uniform)
        if (tid < 8) out[tid] = 0.f;
        if (tid == 0) valid[(size_t)cloud * S + seg] = 0;
        return;
    }

    float res[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (type == T_PLANE || type == T_CONE) {
        // plane fit of the points (plane) or of the normals (cone axis, :831)   primitive_forward.py:712-733
        const float* vecs = type == T_PLANE ? v.pts : v.nrm;
        double a1[7] = {0, 0, 0, 0, 0, 0, 0};
        for (int i = tid; i < N; i += 256)
            if (v.in(i)) {
                const float w = v.weight(i);
                a1[0] += (double)w;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    a1[1 + c] += (double)__fmul_rn(w, vecs[3 * i + c]);
                    a1[4 + c] += (double)v.nrm[3 * i + c];                 // unweighted sum of normals (:832)
                }
            }
        block_sum<7>(a1, sh);
        const float wsum = __fadd_rn((float)a1[0], kEPS);
        const float mean[3] = {(float)a1[1] / wsum, (float)a1[2] / wsum, (float)a1[3] / wsum};
        float a[3];
        smallest_dir_block(v, vecs, mean, sh, sb, a);
        if (type == T_PLANE) {
            // d = sum w (a . p) / wsum
            const float d = (float)(((double)a[0] * a1[1] + (double)a[1] * a1[2] + (double)a[2] * a1[3]) / (double)wsum);
            res[0] = a[0]; res[1] = a[1]; res[2] = a[2]; res[3] = d;
        } else {
            // ---- cone (:812-847)
            if ((double)a[0] * a1[4] + (double)a[1] * a1[5] + (double)a[2] * a1[6] > 0.0) { a[0] = -a[0]; a[1] = -a[1]; a[2] = -a[2]; }
            double a2[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
            for (int i = tid; i < N; i += 256)
                if (v.in(i)) {
                    const float w = v.weight(i);
                    const float n0 = v.nrm[3 * i], n1 = v.nrm[3 * i + 1], n2 = v.nrm[3 * i + 2];
                    const float p0 = v.pts[3 * i], p1 = v.pts[3 * i + 1], p2 = v.pts[3 * i + 2];
                    const float A[3] = {__fmul_rn(w, n0), __fmul_rn(w, n1), __fmul_rn(w, n2)};
                    const float np = __fadd_rn(__fadd_rn(__fmul_rn(n0, p0), __fmul_rn(n1, p1)), __fmul_rn(n2, p2));
                    const float Y = __fmul_rn(w, np);
                    a2[0] += (double)A[0] * A[0]; a2[1] += (double)A[0] * A[1]; a2[2] += (double)A[0] * A[2];
                    a2[3] += (double)A[1] * A[1]; a2[4] += (double)A[1] * A[2]; a2[5] += (double)A[2] * A[2];
                    a2[6] += (double)A[0] * Y; a2[7] += (double)A[1] * Y; a2[8] += (double)A[2] * Y;
                }
            block_sum<9>(a2, sh);
            if (tid == 0) {
                double w3[3], V[3][3];
                eig3(a2, w3, V);
                const double smin = sqrt(fmax(w3[0], 0.0)), smax = sqrt(fmax(w3[2], 0.0));
                if (!(smax <= 1e5 * smin)) {             // np.linalg.cond(A) > 1e5 -> zero cone (:822-827)
                    sb[3] = 1.f;
                } else {
                    double x[3];
                    lstsq3(a2, a2 + 6, count, x);
                    sb[0] = (float)x[0]; sb[1] = (float)x[1]; sb[2] = (float)x[2]; sb[3] = 0.f;
                }
            }
            __syncthreads();
            const bool zero_cone = sb[3] != 0.f;
            const float c[3] = {sb[0], sb[1], sb[2]};
            __syncthreads();
            if (zero_cone) {
                res[3] = 1.f;                          // apex 0, axis (1,0,0), theta 0
            } else {
                double a3[1] = {0};
                for (int i = tid; i < N; i += 256)
                    if (v.in(i)) {
                        const float d0 = __fsub_rn(v.pts[3 * i], c[0]), d1 = __fsub_rn(v.pts[3 * i + 1], c[1]),
                                    d2 = __fsub_rn(v.pts[3 * i + 2], c[2]);
                        const float nr = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(d0, d0), __fmul_rn(d1, d1)), __fmul_rn(d2, d2)));
                        const float den = fmaxf(nr, 1e-12f);                      // F.normalize eps
                        float cs = fmaf(d2 / den, a[2], fmaf(d1 / den, a[1], __fmul_rn(d0 / den, a[0])));
                        cs = fminf(fabsf(cs), 0.999f);
                        a3[0] += (double)__fmul_rn(v.weight(i), acosf(cs));
                    }
                block_sum<1>(a3, sh);
                float theta = (float)a3[0] / wsum;
                theta = fminf(fmaxf(theta, 1e-3f), 3.142f / 2 - 1e-3f);            // :846
                res[0] = c[0]; res[1] = c[1]; res[2] = c[2];
                res[3] = a[0]; res[4] = a[1]; res[5] = a[2]; res[6] = theta;
            }
        }
    } else if (type == T_SPHERE) {
        float c[3], r;
        fit_sphere_block(v, nullptr, count, sh, sb, c, &r);
        res[0] = c[0]; res[1] = c[1]; res[2] = c[2]; res[3] = r;
    } else {   // T_CYLINDER (:788-810)
        float a[3];
        smallest_dir_block(v, v.nrm, nullptr, sh, sb, a);
        const float nrm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(a[0], a[0]), __fmul_rn(a[1], a[1])), __fmul_rn(a[2], a[2]))) + kEPS;
        a[0] /= nrm; a[1] /= nrm; a[2] /= nrm;
        float c[3], r;
        fit_sphere_block(v, a, count, sh, sb, c, &r);
        res[0] = a[0]; res[1] = a[1]; res[2] = a[2]; res[3] = c[0]; res[4] = c[1]; res[5] = c[2]; res[6] = r;
    }
    if (tid < 8) out[tid] = res[tid];
    if (tid == 0) valid[(size_t)cloud * S + seg] = 1;
}

// ---------------------------------------------------------------------------------------------------------
// residuals: squared distance of each point to the primitive of its own segment (primitives.py:89-195)
__device__ __forceinline__ float prim_distance(int type, const float* q, const float p[3]) {
    if (type == T_PLANE) {
        const float t = fmaf(p[2], q[2], fmaf(p[1], q[1], __fmul_rn(p[0], q[0])));
        const float e = __fsub_rn(t, q[3]);
        return __fmul_rn(e, e);
    } else if (type == T_SPHERE) {
        const float d0 = p[0] - q[0], d1 = p[1] - q[1], d2 = p[2] - q[2];
        const float e = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(d0, d0), __fmul_rn(d1, d1)), __fmul_rn(d2, d2))) - q[3];
        return __fmul_rn(e, e);
    } else if (type == T_CYLINDER) {
        const float v0 = p[0] - q[3], v1 = p[1] - q[4], v2 = p[2] - q[5];
        const float t = fmaf(v2, q[2], fmaf(v1, q[1], __fmul_rn(v0, q[0])));
        float ds = __fsub_rn(__fadd_rn(__fadd_rn(__fmul_rn(v0, v0), __fmul_rn(v1, v1)), __fmul_rn(v2, v2)), __fmul_rn(t, t));
        ds = fmaxf(ds, 1e-5f);
        const float e = sqrtf(ds) - q[6];
        return __fmul_rn(e, e);
    } else {   // cone: apex q[0..2], axis q[3..5], theta q[6]
        const float v0 = __fadd_rn(p[0] - q[0], 1e-8f), v1 = __fadd_rn(p[1] - q[1], 1e-8f), v2 = __fadd_rn(p[2] - q[2], 1e-8f);
        const float mod = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(v0, v0), __fmul_rn(v1, v1)), __fmul_rn(v2, v2)));
        float ax = fmaf(v2, q[5], fmaf(v1, q[4], __fmul_rn(v0, q[3]))) / (mod + 1e-7f);
        ax = fminf(fmaxf(ax, -0.999f), 0.999f);
        const float da = fminf(fabsf(acosf(ax) - q[6]), 3.142f / 2.0f);
        const float e = __fmul_rn(mod, sinf(da));
        return __fmul_rn(e, e);
    }
}

__global__ __launch_bounds__(256) void residual_segments_kernel(const float* __restrict__ points,
                                                                const int* __restrict__ labels,
                                                                const int* __restrict__ seg_type,
                                                                const float* __restrict__ params,
                                                                const int* __restrict__ valid, int N, int S,
                                                                int take_sqrt, float* __restrict__ per_point,
                                                                float* __restrict__ seg_mean) {
    __shared__ double sh[4 * 2];
    const int seg = blockIdx.x, cloud = blockIdx.y, tid = threadIdx.x;
    const size_t sidx = (size_t)cloud * S + seg;
    const int type = seg_type[sidx];
    const bool ok = valid[sidx] != 0;
    float q[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) q[i] = params[sidx * 8 + i];
    const float* pts = points + (size_t)cloud * N * 3;
    const int* lab = labels ? labels + (size_t)cloud * N : nullptr;
    double acc[2] = {0, 0};
    for (int i = tid; i < N; i += 256)
        if (!lab || lab[i] == seg) {
            float d = 0.f;
            if (ok) {
                const float p[3] = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
                d = prim_distance(type, q, p);
                if (take_sqrt) d = sqrtf(fmaxf(d, 1e-5f));                   // guard_sqrt
            }
            if (per_point) per_point[((size_t)cloud * N + i) * (lab ? 1 : S) + (lab ? 0 : seg)] = d;
            acc[0] += (double)d;
            acc[1] += 1.0;
        }
    block_sum<2>(acc, sh);
    if (tid == 0) seg_mean[sidx] = ok && acc[1] > 0 ? (float)(acc[0] / acc[1]) : 0.f;
}

}  // namespace

// points/normals [B,N,3]; labels [B,N] int32 in 0..S-1 or NULL (every point belongs to every segment);
// seg_type [B,S] (1 plane, 3 cone, 4 cylinder, 5 sphere; anything else -> skipped);
// weights: wmode 0 none (unit), 1 [B,N] weight of each point for its own segment, 2 [B,N,S] soft weights;
//          weight_eps is added to every weight (fit_one_shape_torch adds EPS, primitive_forward.py:947,963;
//          direct Fit.fit_*_torch calls pass 0).
// params [B,S,8]: plane (a,d) | sphere (c,r) | cylinder (a,c,r) | cone (apex,axis,theta); valid [B,S].
extern "C" int sed_fit_segments_f32(int B, int N, int S, const float* points, const float* normals, const int* labels,
                                    const int* seg_type, const float* weights, int wmode, float weight_eps,
                                    int min_points, float* params, int* valid, hipStream_t stream) {
    if (B <= 0 || N <= 0 || S <= 0 || !points || !normals || !seg_type || !params || !valid) return SED_EINVAL;
    if (wmode < 0 || wmode > 2 || (wmode != 0 && !weights)) return SED_EINVAL;
    fit_segments_kernel<<<dim3(S, B), 256, 0, stream>>>(points, normals, labels, seg_type, weights, wmode, weight_eps,
                                                        N, S, min_points, params, valid);
    SED_LAUNCH_CHECK();
    return SED_OK;
}

// per_point: NULL, or [B,N] (labels given: distance to the point's own segment) / [B,N,S] (labels NULL);
// seg_mean [B,S] = mean over the segment's points (0 for invalid segments).
extern "C" int sed_residual_segments_f32(int B, int N, int S, const float* points, const int* labels,
                                         const int* seg_type, const float* params, const int* valid, int take_sqrt,
                                         float* per_point, float* seg_mean, hipStream_t stream) {
    if (B <= 0 || N <= 0 || S <= 0 || !points || !seg_type || !params || !valid || !seg_mean) return SED_EINVAL;
    residual_segments_kernel<<<dim3(S, B), 256, 0, stream>>>(points, labels, seg_type, params, valid, N, S, take_sqrt,
                                                             per_point, seg_mean);
    SED_LAUNCH_CHECK();
    return SED_OK;
}

// LeastSquares.lstsq for an m x 3 system (fitting_utils.py:36-65): x [3] (device). Single workgroup.
namespace {
__global__ __launch_bounds__(256) void lstsq3_kernel(const float* __restrict__ A, const float* __restrict__ Y, int m,
                                                     float* __restrict__ x) {
    __shared__ double sh[4 * 9];
    double a[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = threadIdx.x; i < m; i += 256) {
        const float r0 = A[3 * i], r1 = A[3 * i + 1], r2 = A[3 * i + 2], y = Y[i];
        a[0] += (double)r0 * r0; a[1] += (double)r0 * r1; a[2] += (double)r0 * r2;
        a[3] += (double)r1 * r1; a[4] += (double)r1 * r2; a[5] += (double)r2 * r2;
        a[6] += (double)r0 * y; a[7] += (double)r1 * y; a[8] += (double)r2 * y;
    }
    block_sum<9>(a, sh);
    if (threadIdx.x == 0) {
        double s[3];
        lstsq3(a, a + 6, m, s);
        x[0] = (float)s[0]; x[1] = (float)s[1]; x[2] = (float)s[2];
    }
}
}  // namespace

extern "C" int sed_lstsq3_f32(int m, const float* A, const float* Y, float* x, hipStream_t stream) {
    if (m <= 0 || !A || !Y || !x) return SED_EINVAL;
    lstsq3_kernel<<<1, 256, 0, stream>>>(A, Y, m, x);
    SED_LAUNCH_CHECK();
    return SED_OK;
}
