// The per-cloud evaluation the reference script logs (generate_predictions_aug.py:389-441), batched and entirely on the device:
// segment IoU, primitive-type IoU and chamfer recall over Hungarian-matched (predicted, true) segments.
//
// Replaces /root/reference/src/segment_utils.py:194-242 (SIOU_matched_segments_usecd), :424-494
// (mean_IOU_primitive_segment_usecd), :509-517 (primitive_type_segment_torch), :609-627 (relaxed_iou_fast),
// src/fitting_utils.py:362-376 (match: lapsolver.solve_dense) for hard labels. SURVEY.md section 8 row f-4: the reference does
// the assignment on the host per cloud; here one wave per cloud solves it.
//
//   1. seg_tables_kernel     one workgroup per cloud: K x K overlap counts, segment sizes, (predicted segment, predicted type)
//                            histogram, first point of every true segment -- integer LDS atomics, order-free. Writes the cost
//                            matrix 1 - relaxed IoU with the reference's fp32 operation order.
//   2. seg_assign_kernel     one WAVE per cloud: shortest-augmenting-path Hungarian algorithm with dual variables in fp64; lane j
//                            owns column j (K <= 64): the column scan of a step is one instruction per lane, its minimum one
//                            wave reduction (ties -> lowest column). Exact optimum; among equal-cost optima (empty segments give
//                            masses of cost-1 ties) the choice may differ from another solver's -- no logged number depends on it
//                            (pairs with an empty side are skipped, :456).
//   3. seg_pair_chamfer_kernel  one workgroup per (cloud, predicted segment): chamfer distance between the matched point sets
//                            (index lists from a stable sort by label, supplied by the caller), fp32 distances in the order of
//                            pointops.hip, fp64 sums.
//   4. seg_reduce_kernel     per cloud: means over the matched pairs in row order (fp64, like numpy).
#include "common.h"

namespace {

constexpr int SEG_KMAX = 64;

__device__ __forceinline__ int fold_type(int t) { return (t == 0 || t == 6 || t == 7) ? 9 : (t == 8 ? 2 : t); }   // :210-218

struct SegWs {                  // per cloud, in the workspace
    int* dots;                  // [K][K] overlap counts
    int* np;                    // [K] predicted segment sizes
    int* ng;                    // [K] true segment sizes
    int* ptype;                 // [K] type of predicted segment k: argmax_L histogram (first maximum)
    int* gtype;                 // [K] folded type of the first point of true segment k (-1: empty)
    double* cost;               // [K][K]
    double* pair_cd;            // [K] chamfer distance of the pair (row r, its column), -1: skipped
};

__host__ __device__ inline size_t seg_ws_ints(int K) { return (size_t)K * K + 4 * (size_t)K; }
__host__ __device__ inline size_t seg_ws_doubles(int K) { return (size_t)K * K + (size_t)K; }

__device__ __forceinline__ SegWs seg_ws(void* ws, int B, int K, int cloud) {
    double* d = (double*)ws + (size_t)cloud * seg_ws_doubles(K);
    int* i = (int*)((double*)ws + (size_t)B * seg_ws_doubles(K)) + (size_t)cloud * seg_ws_ints(K);
    SegWs w;
    w.cost = d;
    w.pair_cd = d + (size_t)K * K;
    w.dots = i;
    w.np = i + (size_t)K * K;
    w.ng = w.np + K;
    w.ptype = w.ng + K;
    w.gtype = w.ptype + K;
    return w;
}

__global__ __launch_bounds__(256) void seg_tables_kernel(const int* __restrict__ pred, const int* __restrict__ gt,
                                                         const int* __restrict__ pred_types,
                                                         const int* __restrict__ gt_types, int N, int K, int B, void* ws,
                                                         int* __restrict__ bad) {
    extern __shared__ int sh[];                         // dots [K][K] | np [K] | ng [K] | hist [K][10] | first [K]
    int* dots = sh;
    int* np_ = dots + K * K;
    int* ng_ = np_ + K;
    int* hist = ng_ + K;
    int* first = hist + K * 10;
    const int cloud = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < K * K + 2 * K + 10 * K; i += 256) sh[i] = 0;
    for (int i = tid; i < K; i += 256) first[i] = 0x7FFFFFFF;
    __syncthreads();
    const size_t base = (size_t)cloud * N;
    for (int i = tid; i < N; i += 256) {
        const int p = pred[base + i], g = gt[base + i];
        if (p < 0 || p >= K || g < 0 || g >= K) { *bad = 1; continue; }         // to_one_hot(., 50) would raise (:536-545)
        atomicAdd(&dots[p * K + g], 1);
        atomicAdd(&np_[p], 1);
        atomicAdd(&ng_[g], 1);
        const int t = fold_type(pred_types[base + i]);
        if (t >= 0 && t < 10) atomicAdd(&hist[p * 10 + t], 1); else *bad = 1;
        atomicMin(&first[g], i);
    }
    __syncthreads();
    SegWs w = seg_ws(ws, B, K, cloud);
    for (int i = tid; i < K * K; i += 256) {
        const int r = i / K, c = i - r * K;
        const float d = (float)dots[i];
        // relaxed_iou_fast (:609-627) in fp32, torch's order: ((norms_p + norms_g) - dots) + 1e-7; then 1.0 - cost (match :373)
        const float den = __fadd_rn(__fsub_rn(__fadd_rn((float)np_[r], (float)ng_[c]), d), 1e-7f);
        w.cost[i] = (double)__fsub_rn(1.0f, __fdiv_rn(d, den));
        w.dots[i] = dots[i];
    }
    for (int k = tid; k < K; k += 256) {
        w.np[k] = np_[k];
        w.ng[k] = ng_[k];
        int best = hist[k * 10], bt = 0;
        for (int t = 1; t < 10; ++t)
            if (hist[k * 10 + t] > best) { best = hist[k * 10 + t]; bt = t; }
        w.ptype[k] = bt;
        w.gtype[k] = first[k] == 0x7FFFFFFF ? -1 : fold_type(gt_types[base + first[k]]);
    }
}

__device__ __forceinline__ double wave_min_f64(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_xor(v, off, 64));
    return v;
}

// Hungarian algorithm, rows inserted one by one, each by a shortest augmenting path over reduced costs (dual variables u, v).
// Lane j = column j; p = row matched to the column (-1 none), way = previous column on the path.
__global__ __launch_bounds__(64) void seg_assign_kernel(int K, int B, void* ws, int* __restrict__ col_of_row) {
    __shared__ double u[SEG_KMAX];
    __shared__ int pcol[SEG_KMAX], wayc[SEG_KMAX];
    const int cloud = blockIdx.x, j = threadIdx.x;
    const SegWs w = seg_ws(ws, B, K, cloud);
    const bool col = j < K;
    const double INF = 1.0e300;
    double v = 0.0;
    int p = -1;
    u[j] = 0.0;
    __syncthreads();
    for (int i = 0; i < K; ++i) {
        // the path starts at a virtual column matched to row i
        double minv = INF;
        bool used = false;
        int way = -1;                                  // -1: reached from the virtual column
        int i0 = i, j0 = -1;                           // current row; column it was reached through
        for (;;) {
            double cur = INF;
            if (col && !used) {
                cur = w.cost[(size_t)i0 * K + j] - u[i0] - v;
                if (cur < minv) { minv = cur; way = j0; }
            }
            const double cand = (col && !used) ? minv : INF;
            const double delta = wave_min_f64(cand);
            const unsigned long long eq = __builtin_amdgcn_ballot_w64(cand == delta);
            const int j1 = __builtin_ctzll(eq);        // lowest column among equals
            // dual update: rows on the tree (the start row and the rows of used columns) += delta, used columns -= delta
            if (j == 0) u[i] += delta;
            if (col && used) { u[p] += delta; v -= delta; }
            else if (col) minv -= delta;
            __syncthreads();
            if (j == j1) used = true;
            j0 = j1;
            const int pj = __builtin_amdgcn_readlane(p, j1);
            if (pj < 0) break;                         // free column: augment
            i0 = pj;
        }
        // augment: walk back along `way`
        pcol[j] = p;
        wayc[j] = way;
        __syncthreads();
        if (j == 0) {
            int jj = j0;
            while (jj >= 0) {
                const int jp = wayc[jj];
                pcol[jj] = jp >= 0 ? pcol[jp] : i;
                jj = jp;
            }
        }
        __syncthreads();
        p = pcol[j];
        __syncthreads();
    }
    if (col) col_of_row[(size_t)cloud * K + p] = j;     // every column ends matched (square problem)
}

// chamfer distance between points[pred == r] and points[gt == c], c = the column of row r (both sets non-empty).
// idx_p / idx_g: the cloud's point indices sorted (stably) by predicted / true label; off from the segment sizes.
__global__ __launch_bounds__(256) void seg_pair_chamfer_kernel(const float* __restrict__ points /* [B][N][3] */, int N, int K,
                                                               int B, const int* __restrict__ idx_p,
                                                               const int* __restrict__ idx_g,
                                                               const int* __restrict__ col_of_row, void* ws) {
    constexpr int TILE = 512;
    __shared__ float buf[TILE * 3];
    __shared__ double red[4];
    __shared__ int offs[2];
    const int cloud = blockIdx.y, r = blockIdx.x, tid = threadIdx.x;
    const SegWs w = seg_ws(ws, B, K, cloud);
    const int c = col_of_row[(size_t)cloud * K + r];
    const int na = w.np[r], nb = w.ng[c];
    if (na == 0 || nb == 0) {                          // :456 "use only matched segments"
        if (tid == 0) w.pair_cd[r] = -1.0;
        return;
    }
    if (tid == 0) {
        int oa = 0, ob = 0;
        for (int k = 0; k < r; ++k) oa += w.np[k];
        for (int k = 0; k < c; ++k) ob += w.ng[k];
        offs[0] = oa;
        offs[1] = ob;
    }
    __syncthreads();
    const float* P = points + (size_t)cloud * N * 3;
    double total = 0.0;
    for (int side = 0; side < 2; ++side) {
        const int* ia = (side ? idx_g : idx_p) + (size_t)cloud * N + offs[side];
        const int* ib = (side ? idx_p : idx_g) + (size_t)cloud * N + offs[side ^ 1];
        const int n1 = side ? nb : na, n2 = side ? na : nb;
        double s = 0.0;
        for (int a0 = 0; a0 < n1; a0 += 256) {
            const int a = a0 + tid;
            const int pa = ia[a < n1 ? a : n1 - 1];
            const float x1 = P[pa * 3], y1 = P[pa * 3 + 1], z1 = P[pa * 3 + 2];
            float best = 0.f;
            for (int k0 = 0; k0 < n2; k0 += TILE) {
                const int cnt = min(TILE, n2 - k0);
                __syncthreads();
                for (int t = tid; t < cnt; t += 256) {
                    const int pb = ib[k0 + t];
                    buf[t * 3] = P[pb * 3];
                    buf[t * 3 + 1] = P[pb * 3 + 1];
                    buf[t * 3 + 2] = P[pb * 3 + 2];
                }
                __syncthreads();
                for (int k = 0; k < cnt; ++k) {        // pointops.hip chamfer_nn_kernel's arithmetic
                    const float dx = buf[k * 3] - x1, dy = buf[k * 3 + 1] - y1, dz = buf[k * 3 + 2] - z1;
                    const float d = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
                    if ((k0 + k) == 0 || d < best) best = d;
                }
            }
            if (a < n1) s += (double)best;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = s;
        __syncthreads();
        total += (red[0] + red[1] + red[2] + red[3]) / (double)n1;
    }
    if (tid == 0) w.pair_cd[r] = total / 2.0;          // src/utils.py:273-296: (mean_i + mean_j) / 2
}

// out[cloud] = (segment IoU, type IoU, chamfer recall, matched pairs used); pairs[cloud][r] = (gt type, predicted type) or (-1, -1)
__global__ __launch_bounds__(64) void seg_reduce_kernel(int K, int B, void* ws, const int* __restrict__ col_of_row,
                                                        double* __restrict__ out, int* __restrict__ pairs) {
    const int cloud = blockIdx.x;
    if (threadIdx.x != 0) return;
    const SegWs w = seg_ws(ws, B, K, cloud);
    double s_iou = 0.0, p_iou = 0.0;
    int used = 0, recalled = 0, n_gt = 0;
    for (int k = 0; k < K; ++k) n_gt += w.ng[k] > 0 ? 1 : 0;                   // np.unique(labels[b]).shape[0]  (:447)
    for (int r = 0; r < K; ++r) {
        const int c = col_of_row[(size_t)cloud * K + r];
        int gt_t = -1, pr_t = -1;
        if (w.np[r] > 0 && w.ng[c] > 0) {
            const int tp = w.dots[r * K + c];
            s_iou += (double)tp / ((double)(w.np[r] + w.ng[c] - tp) + 1e-8);  // :464
            if (w.pair_cd[r] / 2.0 < 0.1) ++recalled;                          // :471-474
            gt_t = w.gtype[c];
            pr_t = w.ptype[r];
            p_iou += gt_t == pr_t ? 1.0 : 0.0;
            ++used;
        }
        if (pairs) { pairs[((size_t)cloud * K + r) * 2] = gt_t; pairs[((size_t)cloud * K + r) * 2 + 1] = pr_t; }
    }
    const double nan_ = __longlong_as_double(0x7FF8000000000000LL);             // np.mean([]) of a cloud without pairs
    out[cloud * 4 + 0] = used ? s_iou / used : nan_;
    out[cloud * 4 + 1] = used ? p_iou / used : nan_;
    out[cloud * 4 + 2] = n_gt ? (double)recalled / n_gt : nan_;
    out[cloud * 4 + 3] = (double)used;
}

}  // namespace

extern "C" size_t sed_segment_metrics_workspace_bytes(int B, int K) {
    if (B <= 0 || K <= 0 || K > SEG_KMAX) return 0;
    return (size_t)B * (seg_ws_doubles(K) * sizeof(double) + seg_ws_ints(K) * sizeof(int)) + 64;
}

extern "C" int sed_segment_metrics_f32(int B, int N, int K, const int* pred_labels, const int* gt_labels, const int* pred_types,
                                       const int* gt_types, const float* points, const int* idx_by_pred, const int* idx_by_gt,
                                       double* metrics, int* col_of_row, int* pairs, int* bad, void* workspace,
                                       size_t workspace_bytes, hipStream_t stream) {
    if (B <= 0 || N <= 0 || K <= 0 || K > SEG_KMAX) return SED_EINVAL;
    if (!pred_labels || !gt_labels || !pred_types || !gt_types || !points || !idx_by_pred || !idx_by_gt || !metrics || !col_of_row ||
        !bad || !workspace)
        return SED_EINVAL;
    if (workspace_bytes < sed_segment_metrics_workspace_bytes(B, K)) return SED_EINVAL;
    const size_t sh = ((size_t)K * K + 2 * K + 10 * K + K) * sizeof(int);
    seg_tables_kernel<<<B, 256, sh, stream>>>(pred_labels, gt_labels, pred_types, gt_types, N, K, B, workspace, bad);
    SED_LAUNCH_CHECK();
    seg_assign_kernel<<<B, 64, 0, stream>>>(K, B, workspace, col_of_row);
    SED_LAUNCH_CHECK();
    seg_pair_chamfer_kernel<<<dim3(K, B), 256, 0, stream>>>(points, N, K, B, idx_by_pred, idx_by_gt, col_of_row, workspace);
    SED_LAUNCH_CHECK();
    seg_reduce_kernel<<<B, 64, 0, stream>>>(K, B, workspace, col_of_row, metrics, pairs);
    SED_LAUNCH_CHECK();
    return SED_OK;
}
