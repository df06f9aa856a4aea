/* sednet_hip.h -- C ABI of libsedhip.so: the MI355X (gfx950) kernels behind SED-Net's inference hot path.
 *
 * The reference (yuanqili78/SED-Net) has no FFI on this path: it is reached through a Python operator
 * surface (src/SEDNet.py, src/PointNet.py, src/mean_shift.py, src/primitive_forward.py, src/primitives.py)
 * built from stock torch ops. This header is what a maintainer binds (ctypes) to replace
 * those ops; each entry point cites the reference code it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the comment says "host"; the caller owns every buffer;
 *     the library allocates nothing and keeps no global state (re-entrant): every behavioural choice (schedule, number of
 *     weight digits, sampling stride, product arithmetic) is an ARGUMENT of the call it affects.
 *   - `stream` is a hipStream_t (pass torch.cuda.current_stream().cuda_stream); all work is enqueued on it,
 *     nothing synchronises.
 *   - return value: 0 = success, SED_EINVAL (-1) bad argument, SED_EUNSUPPORTED (-2) size outside the
 *     instantiated range, otherwise a hipError_t. Never exits, never prints
 *     (the reference's CUDA helpers printf or exit(-1): chamfer_distance.cu:152-154, cuda_utils.h:35-44).
 *   - point-major layout: per-point feature rows are contiguous, [B, N, D] fp32 with D a multiple of 32
 *     (zero padded) and at most 160; indices are int32.
 */
#ifndef SEDNET_HIP_H
#define SEDNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* sed_stream_t; /* == hipStream_t */

#define SED_OK 0
#define SED_EINVAL (-1)
#define SED_EUNSUPPORTED (-2)

/* ---- build / identity -------------------------------------------------------------------------------- */
int sed_abi_version(void);           /* bumped on any signature change */
const char* sed_build_arch(void);    /* "gfx950" */

/* ---- pairwise rows + exact row selection (kNN graph, bandwidth) ---------------------------------------
 * D [B,N,ldD] workspace rows, consumed once by the selection calls. */

/* D[i][j] = 2 - 2 x_i.x_j                                   src/mean_shift.py:130 */
int sed_pairdist_ms_f32(int B, int N, int d, const float* X, float* D, int ldD, sed_stream_t stream);
/* D[i][j] = -(((-xx_j) + 2 x_i.x_j) - xx_i); first C of d channels are real.   src/PointNet.py:76-78 (knn) */
int sed_pairdist_knn_f32(int B, int N, int d, int C, const float* X, float* xx_ws /*[B*N]*/, float* D, int ldD,
                         sed_stream_t stream);
/* D[i][j] = Dp (1 + W Dn) on x6 [B,6,N] channel-major (xyz, unit normal).     src/PointNet.py:107-128 */
int sed_pairdist_pn_f32(int B, int N, float W, const float* x6, float* D, int ldD, sed_stream_t stream);
/* k-th smallest per row -> kth [B,N]                         src/mean_shift.py:133-135 */
int sed_row_kth_f32(int B, int N, int ldD, int k, const float* D, float* kth, sed_stream_t stream);
/* indices of the k smallest per row, ascending by (value, index) -> idx [B,N,k]   src/PointNet.py:83,133 (topk) */
int sed_row_topk_idx_f32(int B, int N, int ldD, int k, const float* D, int* idx, sed_stream_t stream);

/* Fused streaming kNN (two score sweeps + short candidate lists, no N x N matrix). *overflow (device int) is set to
 * 1 when a candidate list overflowed (masses of duplicate points): the result is then invalid and the caller falls
 * back to sed_pairdist_* + sed_row_topk_idx_f32. k <= sed_knn_fused_max_k() (85), both variants.
 * src/PointNet.py:62-87 (knn) and :90-137 (knn_points_normals) */
size_t sed_knn_fused_workspace_bytes(int B, int N);
int sed_knn_fused_max_k(void);
int sed_knn_fused_f32(int B, int N, int d, int C, int k, const float* X, int* idx, void* ws, size_t ws_bytes,
                      int* overflow, sed_stream_t stream);
int sed_knn_pn_fused_f32(int B, int N, int k, float W, const float* x6, int* idx, void* ws, size_t ws_bytes,
                         int* overflow, sed_stream_t stream);
/* Round 6 (ABI 7): sed_knn_fused_f32 computed on an ORDERED copy of the rows. perm [B,N] int32 = a permutation of 0 .. N-1 per
 * cloud (perm[b][j] = the row that takes position j; NULL = the caller's order). idx is bit-identical to sed_knn_fused_f32's for
 * EVERY permutation (scores belong to (query, key) pairs, candidates keep their original index, ties go by it); a tile-coherent
 * order (rows of a 32-row tile near each other) lets a wave dismiss the key tiles that hold none of its candidates with one
 * comparison per lane. d = 64 / 128; other widths ignore perm.                         src/PointNet.py:62-87
 * sed_spatial_order_f32: such an order from the network's INPUT x6 [B,6,N] (Morton order of the xyz + normal bounding box, 5 bits
 * per channel; one workgroup per cloud; N <= sed_spatial_order_max_points() = 16384). A function of the cloud alone; shared by
 * the feature layers and by the type / instance models of a step. Scheduling only: no reference counterpart. */
int sed_knn_fused_order_f32(int B, int N, int d, int C, int k, const float* X, const int* perm, int* idx, void* ws,
                            size_t ws_bytes, int* overflow, sed_stream_t stream);
int sed_spatial_order_max_points(void);
int sed_spatial_order_f32(int B, int N, const float* x6, int* perm, sed_stream_t stream);
/* the k FARTHEST points per row (same selection on negated distances; farthest first, ties -> lowest index).
 * Replaces knn_idx (square_distance(...).topk(k), largest)            src/smooth_normal_matrix.py:33-40 */
int sed_knn_fused_far_f32(int B, int N, int d, int C, int k, const float* X, int* idx, void* ws, size_t ws_bytes,
                          int* overflow, sed_stream_t stream);
/* Y [B,N,ldy] (first ncol columns) = M X, M = B matrices in CSR sharing the entry capacity nnz_stride (rowptr [B,N+1], col /
 * val [B,nnz_stride]); X [B,N,ldx]; ncol in {4,8,12,16,24,36}. The sparse part of the HPNet affinity operator (one wave per
 * row, deterministic).                                                  src/smooth_normal_matrix.py:42-92, :198 */
int sed_csr_spmm_f32(int B, int N, int ncol, size_t nnz_stride, const int* rowptr, const int* col, const float* val,
                     const float* X, int ldx, float* Y, int ldy, sed_stream_t stream);
/* The CSR of that sparse part built on the device: normals [B,N,3] (unit), nn [B,N,knn] = the farthest-knn graph
 * (sed_knn_fused_far_f32) -> rowptr [B,N+1], col / val [B, 2 knn N], d [B,N] = rowsum^-1/2 (the dense matrix's 1e-12 background
 * counted in). Row c: its knn forward entries in the graph's order, then the transposed entries (c, p : c in nn[p]) with p
 * ascending, each 1/2 (s - 1e-12) d_row d_col; no sort (a bitmap of the transposed pattern is walked row by row), deterministic.
 * src/smooth_normal_matrix.py:42-92 (construction_affinity_matrix_normal) */
size_t sed_hpnet_affinity_csr_workspace_bytes(int B, int N, int knn);
int sed_hpnet_affinity_csr_f32(int B, int N, int knn, float sigma, const float* normals, const int* nn, int* rowptr, int* col,
                               float* val, float* d, void* ws, size_t ws_bytes, sed_stream_t stream);
/* ---- batched LOBPCG on the device (the arithmetic inside torch.lobpcg(A, k = 12, niter = 10), src/smooth_normal_matrix.py:198;
 * lobpcg.hip). The search block lives in two caller-owned buffers S, AS [B,N,ld] with columns [X (k) | R (k) | P (k)]. Nothing is
 * copied to the host between the calls of an iteration.
 * out [B,ma,mb] fp64 = A^T Bm for tall-skinny A [B,N,lda], Bm [B,N,ldb] (ma, mb <= 36; fp64 accumulation, fixed-order) */
size_t sed_tsgemm_tn_workspace_bytes(int B, int N, int ma, int mb);
int sed_tsgemm_tn_f64(int B, int N, int ma, int mb, const float* A, int lda, const float* Bm, int ldb, double* out, void* ws,
                      size_t ws_bytes, sed_stream_t stream);
/* Rayleigh-Ritz: G = S^T S, H = S^T A S [B,m,m] fp64 (m in {12, 24, 36}) -> C [B,m,k] fp32 with C^T G C = I spanning the k largest
 * Ritz pairs (two cyclic Jacobi eigen-decompositions in fp64 per cloud, numerically dependent directions cut off), theta [B,k] */
int sed_ritz_f64(int B, int m, int k, const double* G, const double* H, float* C, float* theta, sed_stream_t stream);
/* R = AX - X lam;  R -= X (X^T R);  R /= ||R|| per column   (lam [B,k]) */
size_t sed_lobpcg_workspace_bytes(int B, int N, int k);
int sed_lobpcg_residual_f32(int B, int N, int k, float* S, const float* AS, int ld, const float* lam, void* ws, size_t ws_bytes,
                            sed_stream_t stream);
/* X <- S C, AX <- AS C, P <- S Cp, AP <- AS Cp in place (m = k, 2 k or 3 k columns in use; Cp = C with its first k rows zeroed) */
int sed_lobpcg_update_f32(int B, int N, int m, int k, float* S, float* AS, int ld, const float* C, sed_stream_t stream);
/* Y [B,N,ldy] (first k columns) += alpha d t^T, d [B,N], t [B,k] fp64: the rank-one background of the affinity operator */
int sed_rank1_add_f32(int B, int N, int k, float* Y, int ldy, const float* d, const double* t, float alpha, sed_stream_t stream);

/* ---- mean-shift clustering --------------------------------------------------------------------------- */
/* bw[b] = max(mean_i sqrt(max(kth[b,i], 1e-6)), min_bw)      src/mean_shift.py:135-137, :34 */
int sed_ms_bandwidth_finalize_f32(int B, int N, float min_bw, const float* kth, float* bw, sed_stream_t stream);
/* K-th smallest (1-based, self included) of 2 - 2 x_i.x_j per row WITHOUT the N x N matrix (two MFMA sweeps + short
 * candidate lists; bit-identical to sed_pairdist_ms_f32 + sed_row_kth_f32). d in {32,64,96,128,160}, K <= sed_ms_kth_fused_max_k(N)
 * (160; 224 on clouds of >= 4096 points, where the first sweep samples every other key tile).
 * overflow [B] (device ints): overflow[b] becomes 1 if a candidate list of cloud b overflowed: kth[b] is then invalid, use
 * the materialised path for that cloud.
 * src/mean_shift.py:115-137 (compute_bandwidth: dist = 2 - 2 X X^T, topk(K)). */
int sed_ms_kth_fused_max_k(int N);
size_t sed_ms_kth_fused_workspace_bytes(int B, int N);
/* sampling: first sweep of clouds of >= 8192 points on every fourth key tile (0 = default, or 4) or on every other one (2);
 * results identical (the second sweep verifies the threshold and flags the cloud otherwise) */
/* X_sorted / order (both or both NULL): the same rows in an order in which 32-row tiles are compact (sed_ms_sparse_prepare_f32's Xs
 * and order: sorted row i = row order[i]): the second sweep then runs on them and visits, per 128-row block, only the key tiles whose
 * cap can hold a value below the block's thresholds (triangle inequality on the tiles' unit means, ms_tiles.hip) -- the same K-th
 * values bit for bit (kth stays in X's row order), a fraction of the tiles on clustered rows. The first sweep keeps the caller's
 * order: its sampled threshold assumes a row's nearest keys are spread over the key tiles. */
int sed_ms_kth_fused_f32(int B, int N, int d, int K, const float* X, float* kth, void* ws, size_t ws_bytes,
                         int* overflow, int sampling, const float* X_sorted, const int* order, sed_stream_t stream);
/* `iters` gaussian mean-shift iterations on unit rows, X [B,N,d] -> newX [B,N,d]; bw [B] on device.
 * src/mean_shift.py:45-79 (mean_shift_), src/guard.py:7-9 */
int sed_ms_iterate_f32(int B, int N, int d, int iters, const float* bw, const float* X, float* newX,
                       sed_stream_t stream);
/* Options of the iteration entry points, passed PER CALL (NULL = all defaults); nothing is remembered between calls.
 * schedule (d = 128 has several that differ only in the order tile contributions are summed):
 *   0 choose by size (default: split-fp16 whenever the workspace is given, its key-chunked form when few clouds would leave CUs
 *     idle), 1 batched fp32 (one workgroup = 128 queries, all keys, all iterations), 2 split-key fp32 (32 queries, keys split over
 *     8 waves), 3 key-chunked fp32, 4 split-fp16 one launch, 5 key-chunked split-fp16 (one launch pair per iteration).
 *   split-fp16 (ms_iterate_f16.hip): the two fp32 products evaluated as fp16 MFMAs on round-to-nearest (h, l) splits of the fp32
 *   operands -- dropped terms <= 3 * 2^-24 relative, fp32 accumulation; clouds whose rows are not unit vectors fall back to the
 *   exact fp32 kernel on the device.
 * weight_digits: fp16 digits of the kernel weights in the split-fp16 second product: 0 (default) / 2 = (h, l) pairs, 6 MFMAs per
 *   32 x 32 x 128 block pair, fp32-equivalent: after 50 iterations the rows sit as far from the exact fp32 kernel as two fp32
 *   summation orders sit from each other (5e-5 on a trained network's 10 000-point embedding), labels equal the reference's up to
 *   true ties; 1 = fp16 heads only, consistently in numerator and row sum (5 MFMAs, 14 % faster; rows within 8e-4 on the same
 *   embedding, 0.2 % of the labels move; clouds in which a weighted mean nearly cancels are flagged on the device and redone with
 *   2 digits). */
typedef struct sed_ms_options {
    int schedule;
    int weight_digits;
} sed_ms_options_t;
/* Same, with a caller-owned workspace of sed_ms_iterate_workspace_bytes(B, N, d, opt) bytes (0 = none needed). */
size_t sed_ms_iterate_workspace_bytes(int B, int N, int d, const sed_ms_options_t* opt);
/* the schedule sed_ms_iterate_ws_f32 takes for this shape and these options when given that workspace (values as
 * sed_ms_options_t.schedule, 1 .. 5); 0 = unsupported shape or options */
int sed_ms_iterate_plan(int B, int N, int d, const sed_ms_options_t* opt);
/* the kernel (template instantiation, as a profiler prints it) that does the iterations of such a call: a static string, "" if the
 * shape / options are unsupported. For measurement records (bench.py's roofline.kernel); not part of the reference's surface. */
const char* sed_ms_iterate_kernel_name(int B, int N, int d, const sed_ms_options_t* opt);
int sed_ms_iterate_ws_f32(int B, int N, int d, int iters, const float* bw, const float* X, float* newX,
                          void* workspace, size_t workspace_bytes, const sed_ms_options_t* opt, sed_stream_t stream);
/* Farthest-point pivot rows for the row order of the block-sparse schedule: greedy k-centre on the unit sphere among rows
 * 0, stride, 2 stride, ... (<= 4096 candidates per cloud), all P steps in one launch. picks [B,P] row indices (the first is
 * row 0), picked [B,P,128] those rows. d = 128. */
int sed_fps_pivots_f32(int B, int N, int d, int stride, int P, const float* X, int* picks, float* picked,
                       sed_stream_t stream);
/* Block-sparse schedule with the products on the fp16 matrix pipe (split-fp16, ms_sparse_f16.hip: ms_sparse_f16_kernel): identical
 * arithmetic, except that 32 x 32 (keys x queries) blocks in which every kernel weight is provably <= e^skip_below are skipped;
 * a row sum (>= 1, the self weight) changes by <= N e^skip_below relative. The Python mirror passes skip_below = -27.04 = ln 2^-39:
 * a weight below 2^-39 rounds to zero when the split-fp16 kernels convert 2^14 p to fp16 -- in the dense kernel too --, so that
 * value drops exactly what the dense kernel cannot represent (N 2^-39 = 1.8e-8 at N = 10 000). Between mask rebuilds a wave also
 * leaves out the first product of a block all of whose weights are provably below that point until the next rebuild.
 * X: unit rows sorted so that 32-row tiles are cluster-pure (any order is correct; the order decides how much is skipped). Every
 * tile t has TWO unit reference vectors -- normalised means of two groups of its rows (before / after a cluster border, or any
 * split) -- stored as row (2 (t / 32) + w) 32 + t % 32 of tile_ref [B, nref, d], nref = sed_ms_iterate_bounds_f16_refs(N),
 * unused rows zero; tile_cosalpha [B, nref]: the smallest dot product of a row of the group with its reference. Every
 * iteration in which a query has turned by more than 0.005 rad since the last time, every wave measures its 32 queries against
 * all references on the matrix pipe and skips the blocks with angle(q, ref) - alpha >= acos(1 + skip_below b^2) + margin for all
 * its queries and both references. workspace: stage images of rows and references + work queues; stats: NULL or
 * sed_ms_iterate_bounds_f16_stats_words() device uint64 counters that are ADDED to (5 in a release build: stage visits of
 * workgroups, first products of waves, second products of waves, stages x iterations per wave = the dense count, mask constructions
 * of workgroups; a -DF16S_PROFILE=1 measurement build appends 7 clock counters). weight_digits: as in sed_ms_options_t.
 * One work item = 128 query rows of a cloud for all iterations, run by PERSISTENT 4-wave workgroups (two per CU): a first launch
 * builds every item's first stage list and reports its length; the items are then queued per XCD -- whole clouds, heaviest first
 * -- and the resident workgroups take items from their XCD's queue (then from the others') through atomic counters in the
 * workspace. form, a bit set: bit 0 = a cloud's items in row order instead of longest first (neighbouring items = queries of the
 * same clusters run at the same time: a smaller working set per L2); bit 1 = ONE resident workgroup per CU instead of two (a
 * measurement switch); bit 2 = (d = 160 only) the 512-register build of the kernel with one workgroup per CU (a measurement switch:
 * DESIGN.md Appendix A); 0 = default; anything above 7 is SED_EINVAL. An item's result does not depend on any other item: a cloud's
 * rows are the same bits whatever else is in the call and whichever form queues it. Clouds whose rows are not unit vectors run
 * the exact dense fp32 kernel. N <= 16 384; d = 128, or 160 (rows padded from the HPNet flow's 140 columns); iters = 0 copies
 * the rows.
 * stop_below (ABI 6; 0 = off, at most 1e-3; an EXPERIMENT beside the contract -- the reference has no such test, it always runs
 * `iterations` steps, mean_shift.py:45-79 -- and off in every default): a WORK ITEM (128 query rows, four waves) all of whose queries
 * have moved by a chord <= stop_below in one iteration ends there: its rows are written as they are. The decision is per item, not
 * per wave: an item's loop bound is cut only when all of its waves' largest steps are at or below the threshold, so with stop_below > 0
 * a row depends on the item's 128 queries and the "bit-identical with 4 / 2 / 1 waves per item" property of the kernel holds for
 * stop_below = 0 only. Rows stay a function of the cloud alone (an item never looks at another cloud). At 5e-6 -- about the step the
 * kernel's chained fp32 accumulation keeps producing at a fixed point -- the iteration launch of the benchmark step is 12 % shorter
 * and 57 of 64 bench clouds keep their labels (profiles/r05_stop_below.md, profiles/r05_freeze_probe.md). */
int sed_ms_iterate_bounds_f16_refs(int N);
int sed_ms_iterate_bounds_f16_stats_words(void);
size_t sed_ms_iterate_bounds_f16_workspace_bytes(int B, int N);
int sed_ms_iterate_bounds_f16_f32(int B, int N, int d, int iters, const float* bw, const float* X, float* newX,
                                  float skip_below, const float* tile_ref, const float* tile_cosalpha, float margin,
                                  void* workspace, size_t workspace_bytes, void* stats, int weight_digits, int form,
                                  float stop_below, sed_stream_t stream);
/* the kernel instantiation that runs the iterations of such a call (static string; "" if unsupported) */
const char* sed_ms_iterate_bounds_f16_kernel_name(int d, int weight_digits);
/* Preparation of the block-sparse schedule, all on the device (ms_sparse_prep.hip): P <= 64 farthest-point pivots among every
 * stride-th row, one k-means step, single-linkage super-groups of the means (merge_angle, radians), rows stable-sorted by
 * (super-group, group) -> order [B,N] (sorted position -> row), Xs [B,N,128] the rows in that order, and per 32-row tile two
 * references + cos(alpha) in the layout sed_ms_iterate_bounds_f16_f32 takes. Deterministic (integer atomics, sums in row order).
 * sed_unsort_rows_f32: out[b, order[b,i]] = in[b,i] -- the result back in the caller's row order. d = 128 or 160, N <= 16 384. */
size_t sed_ms_sparse_prepare_workspace_bytes(int B, int N, int P);
int sed_ms_sparse_prepare_f32(int B, int N, int d, int P, int stride, float merge_angle, const float* X, int* order, float* Xs,
                              float* tile_ref, float* tile_cosalpha, void* workspace, size_t workspace_bytes,
                              sed_stream_t stream);
int sed_unsort_rows_f32(int B, int N, int d, const float* in, const int* order, float* out, sed_stream_t stream);
/* non-max suppression + labels, no host round trip.           src/mean_shift.py:139-179 (nms)
 * labels [B,N] in 0..n_centres-1 (ordered by centre index), centre_ids [B,N] (first n_centres[b] valid),
 * n_labels [B] = distinct labels used (the guard loop's test, generate_predictions_aug.py:31). */
size_t sed_ms_nms_workspace_bytes(int B, int N);
/* centres_sorted / X_sorted / order (all three or all NULL): the same rows in a tile-coherent order (sorted row i = original row
 * order[i]; sed_ms_sparse_prepare_f32's order): the membership sweep then visits, per 128-point block, only the centre tiles that
 * can hold a centre as close as the points' own converged rows -- the same result bit for bit (first-minimum ties by original
 * index). */
int sed_ms_nms_f32(int B, int N, int d, const float* centres, const float* X, const float* bw, int* labels,
                   int* centre_ids, int* n_centres, int* n_labels, void* ws, size_t ws_bytes, const float* centres_sorted,
                   const float* X_sorted, const int* order, sed_stream_t stream);

/* ---- DGCNN backbone ---------------------------------------------------------------------------------- */
/* Fused EdgeConv (gather + [x_j - x_i ; x_i] + Conv2d 1x1 + GroupNorm statistics + max over k).
 * x [B,N,ldx] point-major with C in {6, 64} real channels; idx [B,N,k]; W1t/W2t [C][Cout] = transposed
 * difference / centre halves of the conv weight; sgn [Cout] = +1 where GroupNorm gamma >= 0 else -1.
 * -> ysel [B,N,Cout] (max_k y where gamma >= 0 else min_k y), stats [B][G][2] = (mean, rstd).
 * src/PointNet.py:150-171 + src/SEDNet.py:37-45 + :82 */
size_t sed_edgeconv_partials_bytes(int B, int N, int Cout);
/* products (inference forward of the 64-channel layers): 0 = three-way bf16 splits of both operands on the bf16 matrix pipe
 * (default; six MFMAs per 16 channels, fp32-equivalent), 1 = fp32-input MFMA chains. */
int sed_edgeconv_fwd_f32(int B, int N, int C, int Cout, int k, int G, const float* x, int ldx, const int* idx,
                         const float* W1t, const float* W2t, const float* sgn, float eps, float* ysel, float* stats,
                         void* partials, size_t partials_bytes, int products, sed_stream_t stream);
/* Training forward of the same layer: additionally records jsel [B,N,Cout] u8 = the neighbour slot whose value was
 * selected by the max over k (first one on ties; k <= 255); torch.max's backward routes the gradient there. */
int sed_edgeconv_fwd_train_f32(int B, int N, int C, int Cout, int k, int G, const float* x, int ldx, const int* idx,
                               const float* W1t, const float* W2t, const float* sgn, float eps, float* ysel,
                               float* stats, uint8_t* jsel, void* partials, size_t partials_bytes,
                               sed_stream_t stream);

/* ---- backward of the fused layers (training step) -------------------------------- */
/* What torch.autograd derives for Conv -> GroupNorm -> activation [-> max over k] (src/SEDNet.py:37-45, 78-98,
 * 300-329; train_sed_net.py:233-285 calls loss.backward()).
 * GroupNorm(+activation) backward, reduction part: dout [B,N,ldd]; y [B,N,ldy] = pre-norm values (pointwise layers:
 * the conv output; EdgeConv: ysel); stats from the forward; count = values per group (C/G*N, or C/G*N*k for EdgeConv).
 * Outputs S [B,N,C] = rstd*gamma*dout*act'(z), per-cloud dgamma_b / dbeta_b [B,C], ak [B,G,2] = (alpha, kappa) such that
 * d(pre-norm)[p,j,o] = S[p,o][j == j*] + alpha_g + kappa_g * y[p,j,o]. act: 0 none, 1 ReLU, 2 LeakyReLU(slope). */
size_t sed_gn_bwd_partials_bytes(int B, int N, int C);
int sed_gn_bwd_reduce_f32(int B, int N, int C, int G, double count, const float* dout, int ldd, const float* y,
                          int ldy, const float* stats, const float* gamma, const float* beta, int act, float slope,
                          float* S, float* dgamma_b, float* dbeta_b, float* ak, void* partials,
                          size_t partials_bytes, sed_stream_t stream);
/* pointwise layers: S <- dy = S + alpha_g + kappa_g * y (the two GEMMs dX = dy W, dW = dy^T X are plain library GEMMs) */
int sed_gn_bwd_apply_f32(int B, int N, int C, int G, float* S, const float* y, int ldy, const float* ak,
                         sed_stream_t stream);
/* EdgeConv backward without materialising y / dy: dW1t, dW2t [C][Cout] (overwritten; deterministic) and, if dx != NULL
 * (C = 64 layers), the input gradient dx [B,N,lddx]:
 *   rptr == NULL: dx += with fp32 atomics (caller zero-initialises; the order of the additions varies run to run);
 *   rptr [B,N+1], redge [B,N k]: the reverse graph -- edge ids p k + j stably sorted by their target idx[p][j]; row t owns
 *   redge[rptr[t] .. rptr[t+1]) -- and edge_ws (sed_edgeconv_bwd_edge_ws_bytes): DETERMINISTIC, per-edge contributions are
 *   stored and summed per target row in ascending edge order; columns 0..63 of dx are overwritten. Cout <= 128.
 *   bf16 != 0 (with the reverse graph only): input-gradient products on the bf16 matrix pipe (training, configs[4]). */
size_t sed_edgeconv_bwd_partials_bytes(int B, int N, int C, int Cout);
size_t sed_edgeconv_bwd_edge_ws_bytes(int B, int N, int C, int Cout, int k);
int sed_edgeconv_bwd_f32(int B, int N, int C, int Cout, int k, int G, const float* x, int ldx, const int* idx,
                         const float* W1t, const float* W2t, const float* S, const uint8_t* jsel, const float* ak,
                         float* dW1t, float* dW2t, float* dx, int lddx, void* partials, size_t partials_bytes,
                         const int* rptr, const int* redge, void* edge_ws, size_t edge_ws_bytes, int bf16,
                         sed_stream_t stream);

/* Point-wise conv as GEMM: Y = X Wt + bias + cbias[b]; flags 1 ReLU | 2 store Y | 4 GroupNorm partial sums |
 * 8 per-channel max/min over points. Wt [K][Coutp] zero padded (K % 32 == 0, Coutp % 64 == 0).
 * src/SEDNet.py:94 (mlp1), :303-329 (heads) */
size_t sed_pointwise_partials_bytes(int B, int N, int Coutp);
size_t sed_pointwise_colext_bytes(int B, int N, int Coutp);
int sed_pointwise_fwd_f32(int B, int N, int K, int Coutp, int Cout, const float* X, int ldx, const float* Wt,
                          const float* bias, const float* cbias, float* Y, int ldy, void* partials, void* colext,
                          int flags, sed_stream_t stream);
/* ---- inference GEMMs on the bf16 matrix pipe, fp32-equivalent (three-way bf16 splits v = b1 + b2 + b3, 6 bf16 MFMAs per
 * product, dropped terms <= 2^-25 relative, fp32 accumulation, no scales; see pointwise_split_kernel) ----------------------
 * sed_pointwise_split_weights_f32: W [Cout][K] fp32 (the Conv1d weight) -> `wsplit` (sed_pointwise_split_weights_bytes),
 * once per model. sed_pointwise_fwd_split_f32: the contract of sed_pointwise_fwd_f32 with `wsplit` in place of Wt. */
size_t sed_pointwise_split_weights_bytes(int Coutp, int K);
int sed_pointwise_split_weights_f32(int Cout, int Coutp, int K, const float* W, int ldw, void* wsplit,
                                    sed_stream_t stream);
int sed_pointwise_fwd_split_f32(int B, int N, int K, int Coutp, int Cout, const float* X, int ldx, const void* wsplit,
                                const float* bias, const float* cbias, float* Y, int ldy, void* partials, void* colext,
                                int flags, sed_stream_t stream);
/* (ABI 8, round 6) the same GEMM on a layer's PRE-normalisation output: every activation goes through that layer's GroupNorm +
 * activation as it is loaded -- x = act(X a_k + b_k), a_k = rstd_g gamma_k, b_k = beta_k - mean_g a_k: sed_gn_apply_f32's arithmetic
 * (scale 1, no addend), the same bits -- so the normalised tensor is never written or read (src/SEDNet.py:300-317: bn1 -> conv2,
 * bn2 -> mlp_prim_prob1 / mlp_seg_prob1, bn_prim_prob1 -> mlp_prim_prob2 / edge_module / asis, the edge module's norm -> its last conv). in_stats [B][in_G][2] from sed_gn_finalize_f32, in_gamma / in_beta [K], in_act 0 none /
 * 1 ReLU. K <= 512, K % in_G == 0 (SED_EUNSUPPORTED otherwise). */
int sed_pointwise_fwd_split_gn_f32(int B, int N, int K, int Coutp, int Cout, const float* X, int ldx, const void* wsplit,
                                   const float* in_stats, const float* in_gamma, const float* in_beta, int in_G, int in_act,
                                   const float* bias, const float* cbias, float* Y, int ldy, void* partials, void* colext,
                                   int flags, sed_stream_t stream);
/* ---- the same GEMMs in the two-plane split-fp16 form of the selection kernels (split16.h: x 2^e = h + l, 3 fp16 MFMAs per
 * product, dropped terms <= 3 2^-24 relative to |x|max |w|max of the row / channel, fp32 accumulation): half the matrix work of
 * the bf16 form, but every input row needs a magnitude bound -- `rowmax` [B*N], the bit pattern of a float >= max_k |X[row][k]|,
 * which sed_gn_apply_f32 (the producer of every such input in SEDNet.py:300-329) leaves behind. Coutp % 128 == 0.
 * sed_pointwise_split16_weights_f32: W [Cout][K] -> planes h, l [Coutp][K] fp16 of W 2^e (e per output channel) + 2^-e [Coutp]. */
size_t sed_pointwise_split16_weights_bytes(int Coutp, int K);
int sed_pointwise_split16_weights_f32(int Cout, int Coutp, int K, const float* W, int ldw, void* wsplit,
                                      sed_stream_t stream);
int sed_pointwise_fwd_split16_f32(int B, int N, int K, int Coutp, int Cout, const float* X, int ldx, const void* wsplit,
                                  const unsigned* rowmax, const float* bias, const float* cbias, float* Y, int ldy,
                                  void* partials, void* colext, int flags, sed_stream_t stream);
/* ---- training products (BASELINE configs[4]: bf16) ---------------------------------------------
 * sed_pointwise_fwd_bf16 / sed_edgeconv_fwd_train_bf16: the fp32 entry points' contracts with the products in bf16
 * (operands rounded to nearest even while staged, v_mfma_f32_32x32x16_bf16, fp32 accumulate, fp32 / fp64 statistics;
 * tensors stay fp32 in memory). The reference trains in fp32 (train_sed_net.py:233-283). */
int sed_pointwise_fwd_bf16(int B, int N, int K, int Coutp, int Cout, const float* X, int ldx, const float* Wt,
                           const float* bias, const float* cbias, float* Y, int ldy, void* partials, void* colext,
                           int flags, sed_stream_t stream);
int sed_edgeconv_fwd_train_bf16(int B, int N, int C, int Cout, int k, int G, const float* x, int ldx, const int* idx,
                                const float* W1t, const float* W2t, const float* sgn, float eps, float* ysel,
                                float* stats, unsigned char* jsel, void* partials, size_t partials_bytes,
                                sed_stream_t stream);
/* C [M,N] = op(A) op(B): A(m,k) = transA ? A[k lda + m] : A[m lda + k], B(k,n) = transB ? B[n ldb + k] : B[k ldb + n];
 * fp32 in memory, products bf16 (bf16 != 0) or exact fp32 MFMA chains. The two GEMMs of a pointwise layer's backward
 * (dX = dy W, dW = dy^T X; the reference gets them from torch.autograd, train_sed_net.py:272). Long reductions are
 * split over sed_gemm_splits(M, N, K) partial products added in fixed order (workspace: splits * M * N floats). */
int sed_gemm_splits(int M, int N, int K);
int sed_gemm_f32(int M, int N, int K, const float* A, int lda, int transA, const float* B, int ldb, int transB, float* C,
                 int ldc, int bf16, void* workspace, size_t workspace_bytes, sed_stream_t stream);
/* (mean, rstd) per (cloud, group) from the partial sums of sed_pointwise_fwd_f32.  torch.nn.GroupNorm */
int sed_gn_finalize_f32(int B, int N, int Coutp, int G, double count, float eps, const void* partials, float* stats,
                        sed_stream_t stream);
/* out = scale * act(GN(Y)) + addend (stats NULL: no norm; act 0 none, 1 ReLU, 2 LeakyReLU(slope)).
 * src/SEDNet.py:303-326 (bn* + relu, and the w_pos_enc fusion adds at :322, :326)
 * rowmax (optional, [B*N] unsigned, zeroed by the caller before the first call that fills a set of rows): atomically raised to
 * the bit pattern of max_c |out[row][c]| -- several calls may fill column ranges of the same rows (C / 4 a power of two). */
int sed_gn_apply_f32(int B, int N, int C, int G, const float* Y, int ldy, const float* stats, const float* gamma,
                     const float* beta, int act, float slope, float scale, const float* addend, int lda, float* out,
                     int ldo, unsigned* rowmax, sed_stream_t stream);
/* (ABI 8, round 6) out = scale3 * Y3 + (scale1 * act1(GN1(Y1)) + act2(GN2(Y2))) -- the three sed_gn_apply_f32 passes behind the
 * embedding head (src/SEDNet.py:320-326: bn_seg_prob1, asis + its residual add, the position-encoding add) in one kernel, each step in
 * the arithmetic sed_gn_apply_f32 uses for it (same bits). Y3 may be NULL. act 0 none / 1 ReLU; C % 4 == 0, 256 % (C / 4) == 0. */
int sed_gn_apply_fused_f32(int B, int N, int C, const float* Y1, int ld1, const float* stats1, const float* gamma1,
                           const float* beta1, int G1, int act1, float scale1, const float* Y2, int ld2, const float* stats2,
                           const float* gamma2, const float* beta2, int G2, int act2, const float* Y3, int ld3, float scale3,
                           float* out, int ldo, sed_stream_t stream);
/* x4[b][o] = relu(GN(max/min over N)) from the column extrema of mlp1.   src/SEDNet.py:95-96 */
int sed_colext_finalize_f32(int B, int N, int C, int G, const void* colext, const float* stats, const float* gamma,
                            const float* beta, float* out, sed_stream_t stream);
/* out[b][o] = bias[o] + sum_c W[o][c] v[b][c]: the repeated-global-feature part of conv1 collapses to a
 * per-cloud bias.   src/SEDNet.py:300-303 */
int sed_gemv_bias_f32(int B, int Cout, int K, const float* W, int ldw, const float* bias, const float* v, float* out,
                      int ldo, sed_stream_t stream);
/* row-wise log-softmax over C channels.   src/SEDNet.py:313 */
int sed_log_softmax_f32(size_t rows, int C, const float* in, int ld, float* out, int ldo, sed_stream_t stream);

/* ---- primitive fits + residuals --------------------------------------------------------------------- */
/* Weighted LSQ fit of one primitive per (cloud, segment); one launch replaces the per-segment Python loop and
 * its SVD/QR/matrix_rank/cond calls.   src/primitive_forward.py:712-847, :929-1051; src/fitting_utils.py:36-85
 * points/normals [B,N,3]; labels [B,N] in 0..S-1 or NULL (all points in every segment); seg_type [B,S]:
 * 1 plane, 3 cone, 4 cylinder, 5 sphere (others skipped); wmode 0 unit weights, 1 weights [B,N], 2 [B,N,S];
 * params [B,S,8]: plane (a,d) | sphere (c,r) | cylinder (a,c,r) | cone (apex,axis,theta); valid [B,S]. */
int sed_fit_segments_f32(int B, int N, int S, const float* points, const float* normals, const int* labels,
                         const int* seg_type, const float* weights, int wmode, float weight_eps, int min_points,
                         float* params, int* valid, sed_stream_t stream);
/* Squared (or guarded-sqrt) distance of every point to its segment's primitive + per-segment mean.
 * src/primitives.py:89-195 (distance_from_*), :36-44 */
int sed_residual_segments_f32(int B, int N, int S, const float* points, const int* labels, const int* seg_type,
                              const float* params, const int* valid, int take_sqrt, float* per_point, float* seg_mean,
                              sed_stream_t stream);
/* LeastSquares.lstsq for an m x 3 system (QR branch / ridge branch).   src/fitting_utils.py:36-65 */
int sed_lstsq3_f32(int m, const float* A, const float* Y, float* x, sed_stream_t stream);

/* ---- native point-cloud ops of the reference -------------------------------- */
/* Chamfer nearest neighbours both ways: dist1/idx1 [B,n], dist2/idx2 [B,m] (squared L2, ties -> lowest index).
 * src/chamfer_distance/chamfer_distance.cu:6-155; pybind cd.forward_cuda (chamfer_distance.cpp:180-185) */
int sed_chamfer_fwd_f32(int B, int n, int m, const float* xyz1, const float* xyz2, float* dist1, int* idx1,
                        float* dist2, int* idx2, sed_stream_t stream);
/* Gradients w.r.t. both clouds, deterministic (gather, no float atomics); outputs fully overwritten.
 * src/chamfer_distance/chamfer_distance.cu:158-205; cd.backward_cuda */
int sed_chamfer_bwd_f32(int B, int n, int m, const float* xyz1, const float* xyz2, const float* grad_dist1,
                        const int* idx1, const float* grad_dist2, const int* idx2, float* grad_xyz1, float* grad_xyz2,
                        sed_stream_t stream);
/* Furthest point sampling, idx [B,m], first sample = point 0, points with |p|^2 <= 1e-3 skipped; temp_ws [B*n].
 * pointnet2/_ext_src/src/sampling_gpu.cu:74-178, sampling.cpp:70-91 */
int sed_furthest_point_sampling_f32(int B, int n, int m, const float* xyz, float* temp_ws, int* idx,
                                    sed_stream_t stream);
/* First nsample neighbours with d^2 < r^2 in index order, padded with the first hit; idx [B,m,nsample] zeroed by the
 * caller.   pointnet2/_ext_src/src/ball_query_gpu.cu:14-49 */
int sed_ball_query_f32(int B, int n, int m, float radius, int nsample, const float* new_xyz, const float* xyz, int* idx,
                       sed_stream_t stream);
/* out[b,c,j,s] = points[b,c,idx[b,j,s]] (nsample = 1: gather_points).
 * pointnet2/_ext_src/src/group_points_gpu.cu:13-42, sampling_gpu.cu:13-33 */
int sed_group_points_f32(int B, int c, int n, int npoints, int nsample, const float* points, const int* idx, float* out,
                         sed_stream_t stream);
/* Three nearest known points of every unknown point.   pointnet2/_ext_src/src/interpolate_gpu.cu:14-64 */
int sed_three_nn_f32(int B, int n, int m, const float* unknown, const float* known, float* dist2, int* idx,
                     sed_stream_t stream);
/* out[b,c,j] = sum_t points[b,c,idx[b,j,t]] weight[b,j,t].   pointnet2/_ext_src/src/interpolate_gpu.cu:77-109 */
int sed_three_interpolate_f32(int B, int c, int m, int n, const float* points, const int* idx, const float* weight,
                              float* out, sed_stream_t stream);

/* ---- stage glue (keeps the batched driver on the device) ----------------------------------------------- */
/* out[r,:dpad] = in[r,:d] / max(||in[r,:d]||, 1e-12), zero padded.   generate_predictions_aug.py:377,:380 */
int sed_row_normalize_f32(size_t rows, int d, int dpad, const float* in, int ldi, float* out, int ldo,
                          sed_stream_t stream);
/* out[r] = argmax_c in[r,c] (first maximum).   generate_predictions_aug.py:365 */
int sed_row_argmax_f32(size_t rows, int C, const float* in, int ld, int* out, sed_stream_t stream);
/* seg_type[b,s] = most frequent types[b, labels[b,:]==s] (ties -> smallest id); seg_count optional.
 * Fitting_patches_and_edges/residual_utils.py:259 (stats.mode) */
int sed_segment_type_vote(int B, int N, int S, int C, const int* labels, const int* types, int* seg_type,
                          int* seg_count, sed_stream_t stream);

/* ---- HPNet entropy weights ------------------------------------------------------ */
/* src/smooth_normal_matrix.py:131-151 (compute_entropy): sum over all ordered pairs (i, j < M) of ||u_i - u_j|| (mode 0)
 * or of H(exp(-alpha ||u_i - u_j||)), H(s) = -s log(s + 1e-7) - (1 - s) log(1 - s + 1e-7) (mode 1), u [M,ldu] already
 * divided by the per-dimension interval. partials [sed_pair_entropy_partials(M)] doubles: the caller sums them. */
size_t sed_pair_entropy_partials(int M);
int sed_pair_entropy_f32(int M, int K, const float* u, int ldu, int mode, float alpha, const float* alpha_dev, double* partials,
                         sed_stream_t stream);     /* alpha_dev: NULL, or one device float read instead of `alpha` (no host round trip) */
/* K = 128 on the matrix pipe: ||a - b||^2 = |a|^2 + |b|^2 - 2 a.b with the dot products as split-fp16 MFMAs (fp32-equivalent) and
 * the norms in fp32. The caller CENTRES the rows first (column means subtracted: pairwise distances are unchanged, the
 * cancellation in the expansion is); sed_pair_entropy_split_f32 writes the kernel's operands (fp16 digits + norms) into `split`
 * (sed_pair_entropy_split_bytes(M) bytes, caller-owned), sed_pair_entropy_mfma_f32 computes either statistic from them.
 * ldu a multiple of 4. */
size_t sed_pair_entropy_split_bytes(int M);
int sed_pair_entropy_split_f32(int M, int K, const float* u, int ldu, void* split, sed_stream_t stream);
int sed_pair_entropy_mfma_f32(int M, const void* split, int mode, float alpha, const float* alpha_dev, double* partials,
                              sed_stream_t stream);

/* ---- evaluation of a batch on the device ------------------------------------------------------------
 * What generate_predictions_aug.py:389-441 logs per cloud -- src/segment_utils.py:194-242 (SIOU_matched_segments_usecd) with
 * :424-494, :509-517, :609-627 and fitting_utils.py:362-376 (match: lapsolver.solve_dense on the host) -- for hard labels:
 * K x K overlap tables, cost 1 - relaxed IoU (the reference's fp32 operation order), the assignment by one wave per cloud
 * (Hungarian algorithm, fp64 duals, K <= 64; exact optimum, ties between equal-cost optima -> lowest column), chamfer distance of
 * every matched pair, means in fp64.
 * pred_labels / gt_labels in [0, K) (K = 50: to_one_hot's width), pred_types / gt_types: per-point type ids (folded here:
 * {0,6,7} -> 9, 8 -> 2), points [B][N][3]; idx_by_pred / idx_by_gt [B][N]: the cloud's point indices sorted STABLY by predicted /
 * true label. metrics [B][4] = (segment IoU, type IoU, chamfer recall, pairs used; NaN where the reference's np.mean([]) is),
 * col_of_row [B][K] the matching, pairs [B][K][2] (optional) = (true type, predicted type) per row or (-1, -1),
 * bad: set to 1 if a label or type is out of range (the reference's one-hot scatter would raise). */
size_t sed_segment_metrics_workspace_bytes(int B, int K);
int sed_segment_metrics_f32(int B, int N, int K, const int* pred_labels, const int* gt_labels, const int* pred_types,
                            const int* gt_types, const float* points, const int* idx_by_pred, const int* idx_by_gt,
                            double* metrics, int* col_of_row, int* pairs, int* bad, void* workspace, size_t workspace_bytes,
                            sed_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SEDNET_HIP_H */
