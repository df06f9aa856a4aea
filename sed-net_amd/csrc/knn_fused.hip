// Fused streaming kNN: pairwise scores + exact top-k without materialising the N x N matrix.
//
// Replaces /root/reference/src/PointNet.py:62-87 (knn) and :90-137 (knn_points_normals): the reference builds
// B x N x N fp32 (400 MB per 10k-point cloud) and runs torch.topk over it. Here each query row makes TWO sweeps
// over the keys, recomputing the scores (same instructions, same order => bit-identical both times):
//   sweep 1: the keys a lane sees are split into 32 buckets per query (2 lanes x 16 accumulator registers of the
//            32x32 MFMA tile); every bucket keeps its M smallest scores. Those 32 M values are distinct elements of
//            the row, so their k-th smallest T (k <= 32 M) is an upper bound of the row's k-th smallest score.
//   sweep 2: every element with score <= T (expected ~1.5 k of N) is appended to a lane-private candidate list
//            (plain stores, no atomics); >= k of them exist by construction.
//   finalize: one wave per query rank-sorts its short list by (score, index) and writes the k winners in order
//            (ties -> lowest index). A list overflow (only possible with masses of duplicate points) raises a
//            flag and the caller re-runs that batch through the exact materialised path (pairwise.hip + select.hip).
// Work: 2 x 2 N^2 C flops on the fp32 MFMA + N * (CAP lists) bytes, instead of 2 x N^2 x 4 bytes of HBM traffic
// and a 32-step bisection per row.
#include "common.h"
#include "split16.h"
#include <type_traits>

// key chunks of the second sweeps: 1 from ~4 clouds per call on; results do not depend on it
int sed_sel_chunks(int B, int N) {
    const long wg = (long)B * ((N + 127) / 128);
    long S = 320 / (wg > 0 ? wg : 1);
    if (N < 2048) S = 1;
    return S >= 4 ? 4 : (S >= 2 ? 2 : 1);             // a power of two <= SEL_MAXCHUNK (the kernels split the tiles by shifts)
}

namespace {

constexpr int CAPL = 96;          // candidates per lane (two lanes per query)

struct Cand { uint32_t key; int idx; };

// ---- feature-space metric (MFMA) --------------------------------------------------------------------------
// F16: the dot products run on the fp16 matrix pipe through the split-fp16 evaluation of split16.h (X = the row image,
// same bytes per row as the fp32 row; inv = the rows' 2^-e); otherwise exact fp32 MFMA chains on X itself.
// FAR: the k LARGEST distances (smooth_normal_matrix.py:33-40). Tiles are processed by a lambda instantiated twice: only the
// cloud's last, partly filled tile pays for the padding tests.
// CHUNK (sweep 2 at few clouds per call): gridDim.z key chunks per query block. A template parameter because the one-chunk
// instantiation needs 126 registers (4 waves per SIMD) and the chunked one 154 (3 waves: 10 % slower per wave, irrelevant
// when the chip is not full anyway).
template <int NT, int M, int PASS, bool F16, bool FAR, bool CHUNK = false>
__global__ __launch_bounds__(256, 2) void knn_sweep_kernel(const float* __restrict__ X, const float* __restrict__ xx,
                                                           const float* __restrict__ inv,
                                                           int N, int k, uint32_t* __restrict__ Tbuf,
                                                           Cand* __restrict__ lists, int* __restrict__ counts,
                                                           int* __restrict__ overflow) {
    constexpr int D = 32 * NT;
    constexpr int LDX = D + 4;
    constexpr int C4 = D / 4;
    __shared__ __attribute__((aligned(16))) float lds[2][32 * LDX];
    __shared__ __attribute__((aligned(16))) float xxs[2][32];

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, hi = lane >> 5;
    int bxi;
    const int cloud = sed_xcd_cloud_block(&bxi);
    const float* Xc = X + (size_t)cloud * N * D;
    const float* xxc = xx + (size_t)cloud * N;
    const int qrow = bxi * 128 + wave * 32 + li;
    const int qrow_c = qrow < N ? qrow : N - 1;
    const int ntiles = (N + 31) >> 5;

    float q[F16 ? 1 : NT][16];
    h16x8 qh[F16 ? 2 * NT : 1], ql[F16 ? 2 * NT : 1];
    float two_cq = 2.0f;                                   // 2 * 2^-e of the query row (F16), else 2
    if (F16) {
        split_load_query<NT>((const h16*)Xc + (size_t)qrow_c * 2 * D, hi, qh, ql);
        two_cq = 2.0f * inv[(size_t)cloud * N + qrow_c];
    } else {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v = *(const f32x4*)(Xc + (size_t)qrow_c * D + 32 * t + 8 * g + 4 * hi);
#pragma unroll
                for (int c = 0; c < 4; ++c) q[t][4 * g + c] = v[c];
            }
    }
    const float xq = xxc[qrow_c];
    const float* invc = F16 ? inv + (size_t)cloud * N : nullptr;
    __shared__ __attribute__((aligned(16))) float cks[2][32];      // 2^-e of the staged key rows (F16)

    f32x4 stage[NT];
    float stage_xx = 0.f, stage_ck = 0.f;
    auto stage_load = [&](int tile) {
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            const int i = tid + 256 * u;
            const int row = i / C4, c4 = i % C4;
            const int key = tile * 32 + row;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (key < N) v = *(const f32x4*)(Xc + (size_t)key * D + 4 * c4);
            stage[u] = v;
        }
        if (tid < 32) {
            const int key = tile * 32 + tid;
            stage_xx = key < N ? xxc[key] : 0.f;
            if (F16) stage_ck = key < N ? invc[key] : 0.f;
        }
    };
    auto stage_store = [&](int buf) {
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            const int i = tid + 256 * u;
            const int row = i / C4, c4 = i % C4;
            *(f32x4*)(&lds[buf][row * LDX + 4 * c4]) = stage[u];
        }
        if (tid < 32) { xxs[buf][tid] = stage_xx; if (F16) cks[buf][tid] = stage_ck; }
    };

    // sweep 1 keeps the bucket minima as FLOATS (v_min / v_max; converted to order-preserving keys once, for the
    // bisection): any confusion of -0 / +0 in a float min only loosens the bound T, which stays a valid upper bound of
    // the k-th smallest. Sweep 2 rejects with one float compare (a superset of key <= T for the same reason) and builds
    // the key only for the ~0.4 % of elements that survive.
    float bm[M][16];
    if (PASS == 1) {
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) bm[i][r] = 3.0e38f;
    }
    uint32_t T = 0;
    float Tf = 0.f;
    int cnt = 0;
    Cand* mylist = nullptr;
    if (PASS == 2) {
        T = Tbuf[(size_t)cloud * N + qrow_c];
        Tf = T == 0xFFFFFFFFu ? __builtin_inff() : sortable_f32(T);      // fewer than k bucket values: take everything
        mylist = CHUNK ? lists + ((((size_t)cloud * N + qrow_c) * gridDim.z + blockIdx.z) * 2 + hi) * CAPL
                       : lists + (((size_t)cloud * N + qrow_c) * 2 + hi) * CAPL;
    }
    // Few clouds per call: sweep 2 is split over gridDim.z key chunks (sed_sel_chunks: a cloud's 79 workgroups alone cannot
    // fill 256 CUs); chunk z appends to its own pair of half-lists, the finalize kernel ranks the union. Sweep 1 is launched
    // with one chunk.
    const int zsh = 31 - __builtin_clz(gridDim.z);            // chunk counts are powers of two (sed_sel_chunks): no division
    const int t0 = CHUNK ? (int)(ntiles * blockIdx.z) >> zsh : 0, t1 = CHUNK ? (int)(ntiles * (blockIdx.z + 1)) >> zsh : ntiles;

    stage_load(t0);
    stage_store(0);
    __syncthreads();
    int cur = 0;
    // Sweep 1 only needs SOME valid upper bound of the k-th smallest score: on large clouds it visits every other key
    // tile (any 32 M distinct row elements bound the k-th smallest from above); sweep 2 then collects ~2x as many
    // candidates and the selection stays exact. Only for k <= 32, where 2x the candidates (~2.5 k per query, half per
    // lane) stay far below the list capacity; larger k keep the full first sweep.
    const int tstep = (PASS == 1 && N >= 4096 && k <= 32) ? 2 : 1;
    for (int tile = t0; tile < t1; tile += tstep) {
        if (tile + tstep < t1) stage_load(tile + tstep);
        const float* xt = lds[cur];
        f32x16 s;
        if (F16) {
            s = split_tile_keys_on_rows<NT>((const uint8_t*)(xt + li * LDX), hi, qh, ql);
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 xa = *(const f32x4*)(xt + li * LDX + 32 * t + 8 * g + 4 * hi);
#pragma unroll
                    for (int c = 0; c < 4; ++c) s = mfma32(xa[c], q[t][4 * g + c], s);      // keys on rows, queries on lanes
                }
        }
        const bool ragged = (tile == ntiles - 1) && (N & 31);
        // per-key-row constants of this lane's 16 accumulator rows: rows 8 g + 4 hi .. + 3 are contiguous -> 4 + 4 vector reads
        f32x4 xk4[4], ck4[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            xk4[g] = *(const f32x4*)&xxs[cur][8 * g + 4 * hi];
            if (F16) ck4[g] = *(const f32x4*)&cks[cur][8 * g + 4 * hi];
        }
        auto select = [&](auto ragged_c) {
            constexpr bool RAGGED = decltype(ragged_c)::value;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int krow = mfma_row(r, hi);
                const float xk = xk4[r >> 2][r & 3];
                const float dot2 = F16 ? (s[r] * two_cq) * ck4[r >> 2][r & 3] : 2.0f * s[r];   // 2 x_i.x_j (exact power-of-two unscale)
                const float t1 = __fadd_rn(-xk, dot2);               // (-xx_j) - inner, inner = -2 dot   (PointNet.py:76-78)
                float dv = -__fsub_rn(t1, xq);                       // ... - xx_i ; distance = -score
                if (FAR) dv = -dv;
                const bool pad = RAGGED && tile * 32 + krow >= N;
                if (PASS == 1) {
                    float v = pad ? 3.0e38f : dv;
#pragma unroll
                    for (int i = 0; i < M; ++i) {
                        const float lo_ = sed_vmin(bm[i][r], v);
                        if (i + 1 < M) v = sed_vmax(bm[i][r], v);
                        bm[i][r] = lo_;
                    }
                } else {
                    if (dv <= Tf && !pad) {                      // one branch per value; inside it the append is predicated
                        const uint32_t key = f32_sortable(dv);
                        const bool hit = key <= T;
                        Cand c; c.key = key; c.idx = tile * 32 + krow;
                        if (hit && cnt < CAPL) mylist[cnt] = c;
                        cnt += hit ? 1 : 0;
                    }
                }
            }
        };
        if (ragged) select(std::true_type{});
        else select(std::false_type{});
        if (tile + tstep < t1) stage_store(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    if (PASS == 1) {
        // k-th smallest of this query's 32 M bucket values (this lane's + the partner lane's)
        uint32_t bk[M][16];
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) bk[i][r] = bm[i][r] >= 3.0e38f ? 0xFFFFFFFFu : f32_sortable(bm[i][r]);
        uint32_t lo = 0, hiv = 0xFFFFFFFFu;
        for (int it = 0; it < 32; ++it) {
            const uint32_t mid = lo + ((hiv - lo) >> 1);
            int c = 0;
#pragma unroll
            for (int i = 0; i < M; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) c += bk[i][r] <= mid ? 1 : 0;
            c += __shfl_xor(c, 32, 64);
            if (lo < hiv) { if (c >= k) hiv = mid; else lo = mid + 1; }
        }
        if (qrow < N && hi == 0) Tbuf[(size_t)cloud * N + qrow] = lo;
    } else {
        if (qrow < N) {
            counts[CHUNK ? (((size_t)cloud * N + qrow) * gridDim.z + blockIdx.z) * 2 + hi : ((size_t)cloud * N + qrow) * 2 + hi] = cnt;
            if (cnt > CAPL) *overflow = 1;
        }
    }
}

typedef float f32x8u __attribute__((ext_vector_type(8), aligned(4)));      // 8 consecutive floats at any dword address

// ---- first-layer metric Dp (1 + W Dn) on xyz + normals (VALU; one thread per query) -----------------------------
template <int M, int PASS>
__global__ __launch_bounds__(256) void knn_pn_sweep_kernel(const float* __restrict__ x6, int N, int k, float W,
                                                           uint32_t* __restrict__ Tbuf, Cand* __restrict__ lists,
                                                           int* __restrict__ counts, int* __restrict__ overflow) {
    // The 32 keys of a tile are the same for every lane: their coordinates and normals come through the SCALAR cache (uniform
    // addresses -> s_load), only their squared norms (computed once per tile, like the reference's xx) go through LDS, four per
    // read. Round 3 read 2 x 16 bytes of LDS per key and lane-uniformly: the CU's LDS unit was 80 % busy and bounded the kernel.
    // Every rounding of the metric is written out (hipcc contracts a * b + c into an fma wherever it likes, also through
    // __fmul_rn / __fadd_rn, and not the same way in every instantiation): contraction is off in this function and the fmas below are
    // the ones the round-3 build of this kernel and pairwise.hip's materialised form perform, so that the two keep agreeing bit for bit.
#pragma clang fp contract(off)
    __shared__ __attribute__((aligned(16))) float kxx[2][32];
    const int cloud = blockIdx.y, tid = threadIdx.x;
    const float* xc = x6 + (size_t)cloud * 6 * N;
    const int qi = blockIdx.x * 256 + tid;
    const int qc = qi < N ? qi : N - 1;
    const float p0 = xc[qc], p1 = xc[N + qc], p2 = xc[2 * N + qc];
    const float n0 = xc[3 * N + qc], n1 = xc[4 * N + qc], n2 = xc[5 * N + qc];
    auto sqnorm3 = [](float a0, float a1, float a2) { return fmaf(a2, a2, fmaf(a0, a0, a1 * a1)); };
    const float xxi = sqnorm3(p0, p1, p2);
    const int ntiles = (N + 31) >> 5;

    // bucket minima as floats, float prefilter in sweep 2, padding tests only in the last tile: see knn_sweep_kernel
    float bm[M][32];
    if (PASS == 1) {
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
            for (int r = 0; r < 32; ++r) bm[i][r] = 3.0e38f;
    }
    uint32_t T = 0;
    float Tf = 0.f;
    int cnt = 0;
    Cand* mylist = nullptr;
    if (PASS == 2) {
        T = Tbuf[(size_t)cloud * N + qc];
        Tf = T == 0xFFFFFFFFu ? __builtin_inff() : sortable_f32(T);
        mylist = lists + (((size_t)cloud * N + qc) * gridDim.z + blockIdx.z) * 2 * CAPL;       // one thread owns both halves
    }
    const int zsh = 31 - __builtin_clz(gridDim.z);            // chunk counts are powers of two (sed_sel_chunks): no division
    const int t0 = (int)(ntiles * blockIdx.z) >> zsh, t1 = (int)(ntiles * (blockIdx.z + 1)) >> zsh;
    auto stage = [&](int tile, int buf) {
        if (tid < 32) {
            int j = tile * 32 + tid;
            j = j < N ? j : N - 1;
            const float a0 = xc[j], a1 = xc[N + j], a2 = xc[2 * N + j];
            kxx[buf][tid] = sqnorm3(a0, a1, a2);
        }
    };
    stage(t0, 0);
    __syncthreads();
    // sweep 1 on every other key tile for small k on large clouds (see knn_sweep_kernel)
    const int tstep = (PASS == 1 && N >= 4096 && k <= 32) ? 2 : 1;
    int cur = 1;
    for (int tile = t0; tile < t1; tile += tstep) {
        cur ^= 1;
        if (tile + tstep < t1) stage(tile + tstep, cur ^ 1);
        const bool ragged = tile * 32 + 32 > N;
        auto values = [&](auto ragged_c) {
            constexpr bool RAGGED = decltype(ragged_c)::value;
            f32x4 xx4[8];
#pragma unroll
            for (int g = 0; g < 8; ++g) xx4[g] = *(const f32x4*)&kxx[cur][4 * g];
            f32x8u kv[6];
#pragma unroll
            for (int r = 0; r < 32; ++r) {
                const bool pad = RAGGED && tile * 32 + r >= N;
                float k0, k1, k2, k3, k4, k5;
                if (RAGGED) {
                    const int j = pad ? N - 1 : tile * 32 + r;                          // uniform: scalar loads
                    k0 = xc[j]; k1 = xc[N + j]; k2 = xc[2 * N + j]; k3 = xc[3 * N + j]; k4 = xc[4 * N + j]; k5 = xc[5 * N + j];
                } else {
                    if ((r & 7) == 0) {                                                  // 8 keys of a channel per s_load_dwordx8
#pragma unroll
                        for (int c = 0; c < 6; ++c) kv[c] = *(const f32x8u*)(xc + (size_t)c * N + tile * 32 + r);
                    }
                    k0 = kv[0][r & 7]; k1 = kv[1][r & 7]; k2 = kv[2][r & 7]; k3 = kv[3][r & 7]; k4 = kv[4][r & 7]; k5 = kv[5][r & 7];
                }
                const float dotp = fmaf(p2, k2, fmaf(p1, k1, p0 * k0));
                const float dotn = fmaf(n2, k5, fmaf(n1, k4, n0 * k3));
                const float dp = (xx4[r >> 2][r & 3] - 2.0f * dotp) + xxi;               // (xx_j - inner) + xx_i  (:109)
                const float dn = 2.0f - 2.0f * dotn;                                     // :112
                const float dv = dp * (RAGGED ? 1.0f + dn * W : fmaf(W, dn, 1.0f));      // :115 (a cloud's last, ragged tile: two roundings)
                if (PASS == 1) {
                    float v = pad ? 3.0e38f : dv;
#pragma unroll
                    for (int i = 0; i < M; ++i) {
                        const float lo_ = sed_vmin(bm[i][r], v);
                        if (i + 1 < M) v = sed_vmax(bm[i][r], v);
                        bm[i][r] = lo_;
                    }
                } else {
                    if (dv <= Tf && !pad) {
                        const uint32_t key = f32_sortable(dv);
                        const bool hit = key <= T;
                        Cand c; c.key = key; c.idx = tile * 32 + r;
                        if (hit && cnt < 2 * CAPL) mylist[cnt] = c;
                        cnt += hit ? 1 : 0;
                    }
                }
            }
        };
        if (ragged) values(std::true_type{});
        else values(std::false_type{});
        __syncthreads();
    }
    if (PASS == 1) {
        uint32_t bk[M][32];
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
            for (int r = 0; r < 32; ++r) bk[i][r] = bm[i][r] >= 3.0e38f ? 0xFFFFFFFFu : f32_sortable(bm[i][r]);
        uint32_t lo = 0, hiv = 0xFFFFFFFFu;
        for (int it = 0; it < 32; ++it) {
            const uint32_t mid = lo + ((hiv - lo) >> 1);
            int c = 0;
#pragma unroll
            for (int i = 0; i < M; ++i)
#pragma unroll
                for (int r = 0; r < 32; ++r) c += bk[i][r] <= mid ? 1 : 0;
            if (lo < hiv) { if (c >= k) hiv = mid; else lo = mid + 1; }
        }
        if (qi < N) Tbuf[(size_t)cloud * N + qi] = lo;
    } else if (qi < N) {
        counts[(((size_t)cloud * N + qi) * gridDim.z + blockIdx.z) * 2] = cnt < 2 * CAPL ? cnt : 2 * CAPL;
        counts[(((size_t)cloud * N + qi) * gridDim.z + blockIdx.z) * 2 + 1] = 0;
        if (cnt > 2 * CAPL) *overflow = 1;
    }
}

// ---- finalize: one wave per query, rank sort of the short candidate list ------------------------------------------
// The candidates live in registers (lane e of chunk t holds candidate 64 t + e; <= 3 chunks = 2 CAPL) and are broadcast with
// v_readlane: no LDS round trip per comparison (the LDS version spent its time in 60 dependent ds_read latencies per row).
// rank = number of candidates that sort before mine by (key, index); ranks are distinct, the k smallest are written in order.
constexpr int FIN_ROWS = 4;                        // rows per wave: all their loads are in flight before the first rank loop
constexpr int SEL_MAXCHUNK = 4;                    // key chunks of sweep 2 at few clouds per call (sed_sel_chunks)
// S = key chunks of sweep 2: a row owns S pairs of half-lists (pair p at ((row S + p) 2) CAPL), pair p holds a_p entries from
// its start and b_p entries from its second half (the xyz-normal kernel packs everything into the first). The union has the
// same members as the one-chunk lists (same threshold T), so the ranks -- hence the output -- do not depend on S; a union
// beyond the 2 CAPL register slots raises the overflow flag like a full list does.
// perm (round 6, may be null): the lists belong to row POSITIONS of an ordered cloud (knn_ordered.h); position j's neighbours are
// written to output row perm[j] of its cloud (N rows per cloud).
__global__ __launch_bounds__(256) void knn_finalize_kernel(const Cand* __restrict__ lists, const int* __restrict__ counts,
                                                           int k, size_t rows, int S, int* __restrict__ idx_out,
                                                           int* __restrict__ overflow, const int* __restrict__ perm, int N) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t row0 = ((size_t)blockIdx.x * 4 + wave) * FIN_ROWS;
    if (row0 >= rows) return;
    auto out_row = [&](size_t row) { return perm ? row - row % (size_t)N + (size_t)perm[row] : row; };
    constexpr int NCH = (2 * CAPL + 63) / 64;
    int pa[FIN_ROWS][SEL_MAXCHUNK], pb[FIN_ROWS][SEL_MAXCHUNK], C[FIN_ROWS];
#pragma unroll
    for (int q = 0; q < FIN_ROWS; ++q) {
        const size_t row = row0 + q < rows ? row0 + q : rows - 1;
        int tot = 0;
#pragma unroll
        for (int p = 0; p < SEL_MAXCHUNK; ++p) {
            int a = 0, b = 0;
            if (p < S) {
                a = counts[(row * S + p) * 2];
                b = counts[(row * S + p) * 2 + 1];
                a = a < CAPL * 2 ? a : CAPL * 2;        // pn kernel packs everything into the first half-list pair
                b = b < CAPL ? b : CAPL;
                if (b > 0 && a > CAPL) a = CAPL;
            }
            pa[q][p] = __builtin_amdgcn_readfirstlane(a);
            pb[q][p] = __builtin_amdgcn_readfirstlane(b);
            tot += pa[q][p] + pb[q][p];
        }
        if (tot > 2 * CAPL) { tot = 2 * CAPL; if (lane == 0 && row0 + q < rows) *overflow = 1; }
        C[q] = row0 + q < rows ? tot : 0;
    }
    uint32_t key[FIN_ROWS][NCH];                        // (key, index) compared as ONE 64-bit number: key in the high word
    int idx[FIN_ROWS][NCH];
#pragma unroll
    for (int q = 0; q < FIN_ROWS; ++q) {
        const size_t row = row0 + q < rows ? row0 + q : rows - 1;
        const Cand* base = lists + row * S * 2 * CAPL;
        const int Cu = __builtin_amdgcn_readfirstlane(C[q]);
#pragma unroll
        for (int t = 0; t < NCH; ++t) {
            // unconditional loads from clamped positions (the lists are allocated in full), selected afterwards: no divergent
            // branch, so the loads of all rows are in flight together; chunks beyond the row's count are skipped uniformly
            const int e = lane + 64 * t;
            key[q][t] = 0xFFFFFFFFu; idx[q][t] = 0x7FFFFFFF;
            if (64 * t < Cu) {
                const bool ok = e < Cu;
                int rel = ok ? e : 0, off = 0;          // entry `rel` of the union -> position in its pair
#pragma unroll
                for (int p = 0; p < SEL_MAXCHUNK; ++p) {
                    const int n = pa[q][p] + pb[q][p];
                    if (p < S - 1 && rel >= n && off == 2 * CAPL * p) { rel -= n; off = 2 * CAPL * (p + 1); }
                }
                const int p = off / (2 * CAPL);
                int ap = pa[q][0];
#pragma unroll
                for (int pp = 1; pp < SEL_MAXCHUNK; ++pp) ap = p == pp ? pa[q][pp] : ap;
                const Cand c = base[off + (rel < ap ? rel : CAPL + rel - ap)];
                key[q][t] = ok ? c.key : 0xFFFFFFFFu;
                idx[q][t] = ok ? c.idx : 0x7FFFFFFF;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < FIN_ROWS; ++q) {
        const int Cq = __builtin_amdgcn_readfirstlane(C[q]);
        if (row0 + q < rows && Cq < k) {
            // fewer candidates than neighbours (rows with NaN features compare below no threshold): the row is handed to the exact
            // path like an overflowing one, and until then its unfilled slots hold a VALID index -- with deferred flags the graph is
            // consumed before the flag is read, and a stale word of the output buffer must never become a gather address
            int* out_ = idx_out + out_row(row0 + q) * k;
            for (int e = Cq + lane; e < k; e += 64) out_[e] = 0;
            if (lane == 0) *overflow = 1;
        }
        if (Cq == 0) continue;
        int rank[NCH];
#pragma unroll
        for (int t = 0; t < NCH; ++t) rank[t] = 0;
        auto against_chunk = [&](int t2, auto nch_c) {    // ranks of the first NCT chunks against the candidates of chunk t2
            constexpr int NCT = decltype(nch_c)::value;
            const int n2 = Cq - 64 * t2 < 64 ? Cq - 64 * t2 : 64;
            for (int j = 0; j < n2; ++j) {
                const uint32_t ok = (uint32_t)__builtin_amdgcn_readlane((int)key[q][t2], j);
                const uint32_t oi = (uint32_t)__builtin_amdgcn_readlane(idx[q][t2], j);
                const unsigned long long o64 = ((unsigned long long)ok << 32) | oi;
#pragma unroll
                for (int t = 0; t < NCT; ++t)
                    rank[t] += o64 < (((unsigned long long)key[q][t] << 32) | (uint32_t)idx[q][t]) ? 1 : 0;
            }
        };
        if (Cq <= 64) {
            against_chunk(0, std::integral_constant<int, 1>{});
        } else {
#pragma unroll
            for (int t2 = 0; t2 < NCH; ++t2)
                if (Cq > 64 * t2) against_chunk(t2, std::integral_constant<int, NCH>{});
        }
        int* out = idx_out + out_row(row0 + q) * k;
#pragma unroll
        for (int t = 0; t < NCH; ++t)
            if (lane + 64 * t < Cq && rank[t] < k) out[rank[t]] = idx[q][t];
    }
}

int pick_M(int k) { return (3 * k + 63) / 64; }      // 32 M >= 1.5 k   (k = 20 -> 1, 32 -> 2, 64 -> 3, 85 -> 4)

#include "knn_ordered.h"

// the row image of an ORDERED cloud: image row j of a cloud = the split of its row perm[j] (same split: a function of the row
// alone), xxo[j] = that row's squared norm (xx is computed in the caller's order first)
template <int D>
__global__ __launch_bounds__(256) void split_rows_perm_kernel(const float* __restrict__ X, const float* __restrict__ xx,
                                                              const int* __restrict__ perm, int N, h16* __restrict__ img,
                                                              float* __restrict__ inv, float* __restrict__ xxo, size_t rows) {
    using Mp = SplitRowMap<D>;
    constexpr int LPR = Mp::LPR, PER = Mp::PER;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t row = i / LPR;
    const int j = (int)(i % LPR);
    const size_t rc = row < rows ? row : rows - 1;
    const size_t src = rc - rc % (size_t)N + (size_t)perm[rc];
    f32x4 v[PER];
    float am = 0.f;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (row < rows) v[u] = *(const f32x4*)(X + src * D + 4 * (j + LPR * u));
        am = fmaxf(am, fmaxf(fmaxf(fabsf(v[u][0]), fabsf(v[u][1])), fmaxf(fabsf(v[u][2]), fabsf(v[u][3]))));
    }
#pragma unroll
    for (int off = LPR / 2; off > 0; off >>= 1) am = fmaxf(am, __shfl_xor(am, off, 64));
    const float scale = split_row_scale(am);
    if (row >= rows) return;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        h16x4 h, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) { h16 a, b; split_pair(v[u][e], scale, a, b); h[e] = a; l[e] = b; }
        *(h16x4*)(img + row * 2 * D + 4 * (j + LPR * u)) = h;
        *(h16x4*)(img + row * 2 * D + D + 4 * (j + LPR * u)) = l;
    }
    if (j == 0) { inv[row] = 1.0f / scale; xxo[row] = xx[src]; }
}

}  // namespace

static inline size_t ord_blocks(int B, int N) { return (size_t)B * (((size_t)N + 127) / 128); }
static inline size_t ord_tiles(int B, int N) { return (size_t)B * (((size_t)N + 31) / 32); }
extern "C" size_t sed_knn_fused_workspace_bytes(int B, int N) {
    const size_t bn = (size_t)B * N;
    const size_t S = (size_t)sed_sel_chunks(B, N);
    return bn * sizeof(float) /*xx*/ + bn * sizeof(uint32_t) /*T*/ + bn * 2 * S * sizeof(int) /*counts*/ +
           bn * 2 * S * CAPL * sizeof(Cand) + 256 +
           bn * sizeof(float) /*row scales*/ + bn * 128 * sizeof(float) /*split-fp16 row image, d <= 128*/ + 256 +
           /* ordered form: xx in row order, sqrt of the tiles' largest xx, block lists + counts */
           bn * sizeof(float) + 256 + ord_tiles(B, N) * sizeof(float) + 256 +
           ord_blocks(B, N) * (((size_t)N + 31) / 32) * sizeof(uint32_t) + 256 + ord_blocks(B, N) * sizeof(int) + 256;
}
extern "C" int sed_knn_fused_max_k(void) { return 85; }

// X [B,N,d] point-major (first C of d channels real) -> idx [B,N,k] int32 sorted by (distance, index).
// *overflow (device int, zeroed here) becomes 1 if any candidate list overflowed: results are then invalid and the
// caller must use sed_pairdist_knn_f32 + sed_row_topk_idx_f32.   Replaces src/PointNet.py:62-87.
extern "C" int sed_knn_fused_f32(int B, int N, int d, int C, int k, const float* X, int* idx, void* ws,
                                 size_t ws_bytes, int* overflow, hipStream_t stream);
// x6 [B,6,N] channel-major -> idx [B,N,k]; metric Dp (1 + W Dn).   Replaces src/PointNet.py:90-137.
extern "C" int sed_knn_pn_fused_f32(int B, int N, int k, float W, const float* x6, int* idx, void* ws,
                                    size_t ws_bytes, int* overflow, hipStream_t stream);

namespace {
struct Ws { float* xx; uint32_t* T; int* counts; Cand* lists; float* inv; h16* img; float* xxo; float* tsq; int* bcount; uint32_t* blist; };
Ws carve(void* ws, int B, int N) {
    const size_t bn = (size_t)B * N;
    Ws w;
    w.xx = (float*)ws;
    w.T = (uint32_t*)(w.xx + bn);
    w.counts = (int*)(w.T + bn);
    const size_t S = (size_t)sed_sel_chunks(B, N);
    w.lists = (Cand*)(((uintptr_t)(w.counts + 2 * S * bn) + 15) & ~(uintptr_t)15);
    w.inv = (float*)(((uintptr_t)(w.lists + bn * 2 * S * CAPL) + 15) & ~(uintptr_t)15);
    w.img = (h16*)(((uintptr_t)(w.inv + bn) + 255) & ~(uintptr_t)255);
    w.xxo = (float*)(((uintptr_t)(w.img + bn * 256) + 255) & ~(uintptr_t)255);
    w.tsq = (float*)(((uintptr_t)(w.xxo + bn) + 255) & ~(uintptr_t)255);
    w.bcount = (int*)(((uintptr_t)(w.tsq + ord_tiles(B, N)) + 255) & ~(uintptr_t)255);
    w.blist = (uint32_t*)(((uintptr_t)(w.bcount + ord_blocks(B, N)) + 255) & ~(uintptr_t)255);
    return w;
}
__global__ __launch_bounds__(256) void sqnorm_kernel(const float* __restrict__ X, float* __restrict__ xx, int rows,
                                                     int D, int C) {
    __shared__ float tile[256 * 33];
    sed_row_sqnorm_block(X, xx, rows, D, C, tile);
}
// d = 64 and d = 128 (the widths SED-Net uses) take the split-fp16 products; the row image is built once per call
// the ordered kernels' tile rings are dynamic LDS (d = 128: 66 KiB + the tile maps, beyond the default limit)
template <int NT, int M>
int ord_raise_lds_limit() {
    static std::atomic<unsigned long long> attr{0};          // devices whose limit has been raised (common.h)
    int attr_err = 0;
    if (sed_first_on_device(attr, &attr_err)) {
        constexpr int SM = OrdRing<NT>::NBUF * OrdRing<NT>::IMG + 4 * (ORD_MAXTILES / 32) * 4 + 64;
        hipError_t e = hipFuncSetAttribute((const void*)knn_ord_bound_kernel<NT, M>, hipFuncAttributeMaxDynamicSharedMemorySize, SM);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)knn_ord_collect_kernel<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, SM);
        if (e != hipSuccess) return (int)e;
        sed_mark_device(attr);
    } else if (attr_err) return attr_err;
    return 0;
}

template <int NT, int M>
void launch_sweeps(dim3 grid, const float* X, const Ws& w, int N, int k, int* overflow, int far, const int* perm, hipStream_t s) {
    constexpr bool F16 = NT == 2 || NT == 4;
    if constexpr (F16) {
        constexpr int D = 32 * NT;
        const size_t rows = (size_t)grid.y * N;
        if (perm && ord_raise_lds_limit<NT, M>() == 0) {      // ordered form (knn_ordered.h; the caller has checked its range)
            const int ntiles = (N + 31) / 32;
            const size_t tt = (size_t)grid.y * ntiles;
            split_rows_perm_kernel<D><<<(unsigned)((rows * SplitRowMap<D>::LPR + 255) / 256), 256, 0, s>>>(X, w.xx, perm, N, w.img, w.inv,
                                                                                                           w.xxo, rows);
            ord_tilemax_kernel<<<(unsigned)((tt + 255) / 256), 256, 0, s>>>(w.xxo, N, ntiles, tt, w.tsq);
            const float* Xi = (const float*)w.img;
            constexpr int SM_C = OrdRing<NT>::NBUF * OrdRing<NT>::IMG, SM_B = SM_C + 4 * (ORD_MAXTILES / 32) * 4 + 64;
            knn_ord_bound_kernel<NT, M><<<grid, 256, SM_B, s>>>(Xi, w.xxo, w.inv, w.tsq, perm, N, k, w.T, w.blist, w.bcount);
            knn_ord_collect_kernel<NT><<<grid, 256, SM_C, s>>>(Xi, w.xxo, w.inv, N, w.T, w.lists, w.counts, overflow, perm, w.blist, w.bcount);
            return;
        }
        split_rows_launch<D>(X, w.img, w.inv, rows, s);
        X = (const float*)w.img;
    }
    const dim3 grid2(grid.x, grid.y, sed_sel_chunks((int)grid.y, N));        // sweep 2: key chunks at few clouds per call
    if (far) {
        knn_sweep_kernel<NT, M, 1, F16, true><<<grid, 256, 0, s>>>(X, w.xx, w.inv, N, k, w.T, w.lists, w.counts, overflow);
        if (grid2.z > 1)
            knn_sweep_kernel<NT, M, 2, F16, true, true><<<grid2, 256, 0, s>>>(X, w.xx, w.inv, N, k, w.T, w.lists, w.counts, overflow);
        else
            knn_sweep_kernel<NT, M, 2, F16, true><<<grid, 256, 0, s>>>(X, w.xx, w.inv, N, k, w.T, w.lists, w.counts, overflow);
    } else {
        knn_sweep_kernel<NT, M, 1, F16, false><<<grid, 256, 0, s>>>(X, w.xx, w.inv, N, k, w.T, w.lists, w.counts, overflow);
        if (grid2.z > 1)
            knn_sweep_kernel<NT, M, 2, F16, false, true><<<grid2, 256, 0, s>>>(X, w.xx, w.inv, N, k, w.T, w.lists, w.counts, overflow);
        else
            knn_sweep_kernel<NT, M, 2, F16, false><<<grid, 256, 0, s>>>(X, w.xx, w.inv, N, k, w.T, w.lists, w.counts, overflow);
    }
}
template <int NT>
int launch_nt(dim3 grid, int M, const float* X, const Ws& w, int N, int k, int* overflow, int far, const int* perm, hipStream_t s) {
    switch (M) {
        case 1: launch_sweeps<NT, 1>(grid, X, w, N, k, overflow, far, perm, s); break;
        case 2: launch_sweeps<NT, 2>(grid, X, w, N, k, overflow, far, perm, s); break;
        case 3: launch_sweeps<NT, 3>(grid, X, w, N, k, overflow, far, perm, s); break;
        case 4: launch_sweeps<NT, 4>(grid, X, w, N, k, overflow, far, perm, s); break;
        default: return SED_EUNSUPPORTED;
    }
    return SED_OK;
}
}  // namespace

static int knn_fused_impl(int B, int N, int d, int C, int k, const float* X, int* idx, void* ws, size_t ws_bytes,
                          int* overflow, int far, const int* perm, hipStream_t stream);

extern "C" int sed_knn_fused_f32(int B, int N, int d, int C, int k, const float* X, int* idx, void* ws,
                                 size_t ws_bytes, int* overflow, hipStream_t stream) {
    return knn_fused_impl(B, N, d, C, k, X, idx, ws, ws_bytes, overflow, 0, nullptr, stream);
}

// The same graph computed on an ORDERED copy of the rows (knn_ordered.h): perm [B,N] int32 (a permutation of 0 .. N-1 per cloud,
// e.g. sed_spatial_order_f32 of the network's input; null = the caller's order). idx is bit-identical to sed_knn_fused_f32's for
// every permutation -- a tile-coherent order only decides how many key tiles a wave can dismiss early. The ordered form runs at
// d = 64 / 128, N <= 16384, for calls large enough to fill the chip with one key chunk; anything else ignores perm.
extern "C" int sed_knn_fused_order_f32(int B, int N, int d, int C, int k, const float* X, const int* perm, int* idx, void* ws,
                                       size_t ws_bytes, int* overflow, hipStream_t stream) {
    return knn_fused_impl(B, N, d, C, k, X, idx, ws, ws_bytes, overflow, 0, perm, stream);
}

// Same selection on the NEGATED distances: idx [B,N,k] = the k FARTHEST points of every row, farthest first (ties ->
// lowest index). Replaces knn_idx of src/smooth_normal_matrix.py:33-40 (square_distance(...).topk(k) -- largest).
extern "C" int sed_knn_fused_far_f32(int B, int N, int d, int C, int k, const float* X, int* idx, void* ws,
                                     size_t ws_bytes, int* overflow, hipStream_t stream) {
    return knn_fused_impl(B, N, d, C, k, X, idx, ws, ws_bytes, overflow, 1, nullptr, stream);
}

static int knn_fused_impl(int B, int N, int d, int C, int k, const float* X, int* idx, void* ws, size_t ws_bytes,
                          int* overflow, int far, const int* perm, hipStream_t stream) {
    if (B <= 0 || N <= 0 || k <= 0 || k > N || !X || !idx || !ws || !overflow || C > d) return SED_EINVAL;
    if (d % 32 != 0 || d < 32 || d > 128 || k > 85) return SED_EUNSUPPORTED;
    if (ws_bytes < sed_knn_fused_workspace_bytes(B, N)) return SED_EINVAL;
    const Ws w = carve(ws, B, N);
    hipError_t e = hipMemsetAsync(overflow, 0, sizeof(int), stream);
    if (e != hipSuccess) return (int)e;
    const int rows = B * N;
    sqnorm_kernel<<<(rows + 255) / 256, 256, 0, stream>>>(X, w.xx, rows, d, C);
    SED_LAUNCH_CHECK();
    dim3 grid((N + 127) / 128, B);
    const int M = pick_M(k);
    // the ordered form: nearest neighbours at the split-fp16 widths, one key chunk, tile maps of <= 512 tiles
    if (!(perm && !far && (d == 64 || d == 128) && sed_sel_chunks(B, N) == 1 && N <= 32 * ORD_MAXTILES && N % 4 == 0)) perm = nullptr;
    int rc;
    switch (d / 32) {
        case 1: rc = launch_nt<1>(grid, M, X, w, N, k, overflow, far, perm, stream); break;
        case 2: rc = launch_nt<2>(grid, M, X, w, N, k, overflow, far, perm, stream); break;
        case 3: rc = launch_nt<3>(grid, M, X, w, N, k, overflow, far, perm, stream); break;
        default: rc = launch_nt<4>(grid, M, X, w, N, k, overflow, far, perm, stream); break;
    }
    if (rc != SED_OK) return rc;
    SED_LAUNCH_CHECK();
    knn_finalize_kernel<<<(unsigned)((rows + 4 * FIN_ROWS - 1) / (4 * FIN_ROWS)), 256, 0, stream>>>(w.lists, w.counts, k, (size_t)rows,
                                                                                                  sed_sel_chunks(B, N), idx, overflow,
                                                                                                  perm, N);
    SED_LAUNCH_CHECK();
    return SED_OK;
}

extern "C" int sed_knn_pn_fused_f32(int B, int N, int k, float W, const float* x6, int* idx, void* ws,
                                    size_t ws_bytes, int* overflow, hipStream_t stream) {
    if (B <= 0 || N <= 0 || k <= 0 || k > N || !x6 || !idx || !ws || !overflow) return SED_EINVAL;
    if (k > 85) return SED_EUNSUPPORTED;                       // M <= 4: 128 bucket registers per thread
    if (ws_bytes < sed_knn_fused_workspace_bytes(B, N)) return SED_EINVAL;
    const Ws w = carve(ws, B, N);
    hipError_t e = hipMemsetAsync(overflow, 0, sizeof(int), stream);
    if (e != hipSuccess) return (int)e;
    dim3 grid((N + 255) / 256, B);
    const dim3 grid2(grid.x, grid.y, sed_sel_chunks(B, N));
    switch (pick_M(k)) {
        case 1:
            knn_pn_sweep_kernel<1, 1><<<grid, 256, 0, stream>>>(x6, N, k, W, w.T, w.lists, w.counts, overflow);
            knn_pn_sweep_kernel<1, 2><<<grid2, 256, 0, stream>>>(x6, N, k, W, w.T, w.lists, w.counts, overflow);
            break;
        case 2:
            knn_pn_sweep_kernel<2, 1><<<grid, 256, 0, stream>>>(x6, N, k, W, w.T, w.lists, w.counts, overflow);
            knn_pn_sweep_kernel<2, 2><<<grid2, 256, 0, stream>>>(x6, N, k, W, w.T, w.lists, w.counts, overflow);
            break;
        case 3:                                                // k = 64, the reference script's default (round 2)
            knn_pn_sweep_kernel<3, 1><<<grid, 256, 0, stream>>>(x6, N, k, W, w.T, w.lists, w.counts, overflow);
            knn_pn_sweep_kernel<3, 2><<<grid2, 256, 0, stream>>>(x6, N, k, W, w.T, w.lists, w.counts, overflow);
            break;
        default:
            knn_pn_sweep_kernel<4, 1><<<grid, 256, 0, stream>>>(x6, N, k, W, w.T, w.lists, w.counts, overflow);
            knn_pn_sweep_kernel<4, 2><<<grid2, 256, 0, stream>>>(x6, N, k, W, w.T, w.lists, w.counts, overflow);
            break;
    }
    SED_LAUNCH_CHECK();
    const size_t rows = (size_t)B * N;
    knn_finalize_kernel<<<(unsigned)((rows + 4 * FIN_ROWS - 1) / (4 * FIN_ROWS)), 256, 0, stream>>>(w.lists, w.counts, k, rows,
                                                                                                  sed_sel_chunks(B, N), idx, overflow,
                                                                                                  nullptr, N);
    SED_LAUNCH_CHECK();
    return SED_OK;
}
