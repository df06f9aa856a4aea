// Opt-in block-sparse mean-shift iterations with geometric bounds (d = 128): skips BOTH products of a 32 x 32
// (keys x queries) block that provably carries only negligible kernel weights, and does not even stage a 64-key slab that
// none of the workgroup's four waves needs.
//
// Same update as /root/reference/src/mean_shift.py:45-79 and ms_iterate_d128_kernel (ms_iterate.hip); what is dropped:
// blocks in which every weight exp(-dist / (2 b^2)) is <= e^skip (skip = -30: 9.4e-14). A row sum is >= 1 (the self weight),
// so dropping such blocks changes it by <= N e^skip relative (1e-9 at N = 10 000, 60 x below fp32 resolution).
//
// The caller (sednet_hip.ops.ms_iterate_sparse) sorts the rows by nearest of P pivot rows and passes the nearest pivot of
// every row, per 32-row tile t a reference pivot rp[t] with alpha[t] = max angle between a row of the tile and that
// pivot, and the table of pivot-pivot angles. Every iteration a query row measures beta = its current angle to its own
// pivot P_a; by the triangle inequality on the unit sphere
//     angle(q, x) >= angle(P_a, P_rp[t]) - beta - alpha[t]      for every key x of tile t,
// so the block is skipped when that bound is >= theta = acos(1 - D/2) + margin, D = -2 skip b^2 (dist = 2 - 2 cos).
// Blocks that survive the bound still get the exact per-element test of the first-level sparse kernel.
#include "common.h"

namespace {

__device__ __forceinline__ float exp_comp(float a) {          // same as ms_iterate.hip: ~1 ulp v_exp_f32
    const float L2E_HI = 1.44269502162933349609375f;
    const float L2E_LO = 1.925963033500011e-08f;
    const float LN2 = 0.693147182464599609375f;
    const float t = a * L2E_HI;
    const float e = fmaf(a, L2E_LO, fmaf(a, L2E_HI, -t));
    const float r = __builtin_amdgcn_exp2f(t);
    return fmaf(r, e * LN2, r);
}

constexpr int MAXW = 8;          // 64-bit words of the tile mask: up to 512 tiles = 16 384 points

__global__ __launch_bounds__(256, 2) void ms_iterate_d128_bounds_kernel(
    const float* __restrict__ X, float* __restrict__ newX, const float* __restrict__ bw, int N, int iters,
    float skip_below, const int* __restrict__ row_piv, const int* __restrict__ tile_rp,
    const float* __restrict__ tile_alpha, const float* __restrict__ piv, const float* __restrict__ pang, int P,
    float margin) {
    constexpr int D = 128, LDX = 132, C4 = 32, KT = 64;
    extern __shared__ __attribute__((aligned(16))) float lds_dyn[];      // [2][KT * LDX]
    __shared__ unsigned long long wmask[4][MAXW];
    __shared__ int slist[512 / 2];
    __shared__ int scount;
    __shared__ int wcount[4];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;      // wave id in a scalar register
    const int li = lane & 31, hi = lane >> 5;
    int bx;
    const int cloud = sed_xcd_cloud_block(&bx);
    const float* Xc = X + (size_t)cloud * N * D;
    const int qrow = bx * 128 + wave * 32 + li;
    const int qrow_c = qrow < N ? qrow : N - 1;
    const float b = bw[cloud];
    const float neg_half_inv_b2 = -0.5f / (b * b);
    const int nst = (N + KT - 1) / KT;
    const int ntile = (N + 31) >> 5;
    const int nword = (ntile + 63) >> 6;
    // angular threshold: dist >= Dthr  <=>  angle >= acos(1 - Dthr / 2)
    const float Dthr = -2.0f * skip_below * b * b;
    const float theta = Dthr < 3.99f ? acosf(1.0f - 0.5f * Dthr) + margin : 1.0e9f;     // 1e9: never skip
    const int* rpc = tile_rp + (size_t)cloud * ntile;
    const float* alc = tile_alpha + (size_t)cloud * ntile;
    // every query row is bounded against ITS OWN nearest pivot (fixed by the sort), so a wave whose 32 rows straddle two
    // clusters needs the tiles of those two clusters only, not everything
    const int myp = row_piv[(size_t)cloud * N + qrow_c];
    const float* mypiv = piv + ((size_t)cloud * P + myp) * D;

    float q[4][16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const f32x4 v = *(const f32x4*)(Xc + (size_t)qrow_c * D + 4 * mfma_row(r, hi));
#pragma unroll
        for (int c = 0; c < 4; ++c) q[c][r] = v[c];
    }
    f32x4 stage[8];
    auto stage_load = [&](int tile) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = tid + 256 * u;
            const int row = i / C4, c4 = i % C4;
            const int key = tile * KT + row;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (key < N) v = *(const f32x4*)(Xc + (size_t)key * D + 4 * c4);
            stage[u] = v;
        }
    };
    auto stage_store = [&](int buf) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = tid + 256 * u;
            const int row = i / C4, c4 = i % C4;
            *(f32x4*)(lds_dyn + buf * KT * LDX + row * LDX + 4 * c4) = stage[u];
        }
    };

    for (int it = 0; it < iters; ++it) {
        // ---- (1) beta: widest angle between this wave's current queries and the pivot of its tile
        float dp = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const f32x4 pv = *(const f32x4*)(mypiv + 4 * mfma_row(r, hi));
#pragma unroll
            for (int c = 0; c < 4; ++c) dp = fmaf(q[c][r], pv[c], dp);
        }
        dp += xor32(dp);
        const float beta_i = qrow < N ? acosf(fminf(fmaxf(dp, -1.0f), 1.0f)) : -1.0e9f;       // this row's angle to its pivot
        // ---- (2) per-wave tile mask (bit = block may carry weight), then the workgroup's list of slabs to stage
        // one pass per DISTINCT pivot among the wave's rows (1 for a cluster-pure wave, 2 at a group boundary): tiles
        // across lanes, beta = widest angle among the rows of that pivot
        if (lane < MAXW) wmask[wave][lane] = 0ull;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        unsigned long long todo = __builtin_amdgcn_ballot_w64(qrow < N);
        while (todo) {
            const int leader = __builtin_ctzll(todo);
            const int a = __builtin_amdgcn_readlane(myp, leader);
            const bool mine = myp == a && qrow < N;
            float beta = mine ? beta_i : -1.0e9f;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) beta = fmaxf(beta, __shfl_xor(beta, off, 64));
            const float* ang = pang + ((size_t)cloud * P + a) * P;
            unsigned long long mw[MAXW];
#pragma unroll
            for (int w = 0; w < MAXW; ++w) {                    // all words' dependent loads in flight together
                const int t = w * 64 + lane;
                bool need = false;
                if (w < nword && t < ntile) need = !(ang[rpc[t]] - beta - alc[t] >= theta);
                mw[w] = __builtin_amdgcn_ballot_w64(need);
            }
#pragma unroll
            for (int w = 0; w < MAXW; ++w)
                if (lane == w) wmask[wave][w] |= mw[w];          // a wave only ever touches its own row of wmask
            todo &= ~__builtin_amdgcn_ballot_w64(mine);
        }
        __syncthreads();
        {   // compact the needed slabs into slist in ascending order: thread s owns slab s (nst <= 256)
            bool need = false;
            if (tid < nst) {
                const int t0 = 2 * tid;
                const int w = t0 >> 6, sh = t0 & 63;               // tiles 2s, 2s+1 live in the same word
                const unsigned long long any = wmask[0][w] | wmask[1][w] | wmask[2][w] | wmask[3][w];
                need = ((any >> sh) & 3ull) != 0ull;
            }
            const unsigned long long bal = __builtin_amdgcn_ballot_w64(need);
            if (lane == 0) wcount[wave] = __builtin_popcountll(bal);
            __syncthreads();
            int base = 0;
            for (int w = 0; w < wave; ++w) base += wcount[w];
            if (need) slist[base + __builtin_popcountll(bal & ((1ull << lane) - 1ull))] = tid;
            if (tid == 0) scount = wcount[0] + wcount[1] + wcount[2] + wcount[3];
        }
        __syncthreads();
        const int ns = scount;

        f32x16 o[4];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[c][r] = 0.f;
        float rsum = 0.f;
        const bool fwd = (it & 1) == 0;                              // ping-pong over the list (L2 re-use)
        int cur = 0;
        if (ns > 0) {
            stage_load(slist[fwd ? 0 : ns - 1]);
            stage_store(0);
        }
        __syncthreads();
        for (int j = 0; j < ns; ++j) {
            const int tile = slist[fwd ? j : ns - 1 - j];
            if (j + 1 < ns) stage_load(slist[fwd ? j + 1 : ns - 2 - j]);
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                const float* xt = lds_dyn + cur * KT * LDX + sub * 32 * LDX;
                const int key0 = tile * KT + sub * 32;
                const int kt = 2 * tile + sub;
                if (key0 < N && ((wmask[wave][kt >> 6] >> (kt & 63)) & 1ull)) {      // wave-uniform
                    f32x16 s;
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const f32x4 xa = *(const f32x4*)(xt + li * LDX + 4 * mfma_row(r, hi));
#pragma unroll
                        for (int c = 0; c < 4; ++c) s = mfma32(xa[c], q[c][r], s);
                    }
                    float p[16];
                    float amax = -3.0e38f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float dist = 2.0f - 2.0f * s[r];
                        float a = dist * neg_half_inv_b2;
                        a = fminf(fmaxf(a, -75.0f), 75.0f);
                        p[r] = a;
                        amax = fmaxf(amax, a);
                    }
                    if (__builtin_amdgcn_ballot_w64(amax >= skip_below) == 0) continue;      // exact per-element test
#pragma unroll
                    for (int r = 0; r < 16; ++r) p[r] = exp_comp(p[r]);
                    if (key0 + 32 > N) {
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            if (key0 + mfma_row(r, hi) >= N) p[r] = 0.f;
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) rsum += p[r];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const f32x4 xb = *(const f32x4*)(xt + mfma_row(r, hi) * LDX + 4 * li);
#pragma unroll
                        for (int c = 0; c < 4; ++c) o[c] = mfma32(xb[c], p[r], o[c]);
                    }
                }
            }
            if (j + 1 < ns) stage_store(cur ^ 1);
            __syncthreads();
            cur ^= 1;
        }
        const float rs = rsum + xor32(rsum);
        const float Dinv = 1.0f / rs;
        float n2 = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float m = o[c][r] * Dinv - q[c][r];
                const float nq = q[c][r] + m;
                q[c][r] = nq;
                n2 += nq * nq;
            }
        n2 += xor32(n2);
        const float nrm = sqrtf(n2);
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) q[c][r] = q[c][r] / nrm;
    }
    if (qrow < N) {
        float* out = newX + ((size_t)cloud * N + qrow) * D;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            f32x4 v = {q[0][r], q[1][r], q[2][r], q[3][r]};
            *(f32x4*)(out + 4 * mfma_row(r, hi)) = v;
        }
    }
}

}  // namespace

// X [B,N,128] unit rows sorted so that 32-row tiles are cluster-pure; row_piv [B,N] nearest pivot of every row;
// tile_rp [B,ceil(N/32)] reference pivot of each tile; tile_alpha [B,ceil(N/32)] max angle (rad) between a row of the tile and that pivot; piv [B,P,128] unit pivot
// rows; pang [B,P,P] pivot-pivot angles (rad). skip_below < 0; margin >= 0 is added to the angular threshold.
extern "C" int sed_ms_iterate_bounds_f32(int B, int N, int d, int iters, const float* bw, const float* X, float* newX,
                                         float skip_below, const int* row_piv, const int* tile_rp,
                                         const float* tile_alpha, const float* piv, const float* pang, int P,
                                         float margin, hipStream_t stream) {
    if (B <= 0 || N <= 0 || iters < 0 || !bw || !X || !newX || !(skip_below < 0.f) || !row_piv || !tile_rp ||
        !tile_alpha || !piv || !pang || P <= 0 || margin < 0.f)
        return SED_EINVAL;
    if (d != 128 || (N + 31) / 32 > 64 * MAXW) return SED_EUNSUPPORTED;
    constexpr int sm = 2 * 64 * 132 * (int)sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)ms_iterate_d128_bounds_kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, sm);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    ms_iterate_d128_bounds_kernel<<<dim3((N + 127) / 128, B), 256, sm, stream>>>(
        X, newX, bw, N, iters, skip_below, row_piv, tile_rp, tile_alpha, piv, pang, P, margin);
    SED_LAUNCH_CHECK();
    return SED_OK;
}

// ------------------------------------------------------------------------------------------------------------
// Farthest-point pivots for the row order of the block-sparse schedules (greedy k-centre on the unit sphere): P dependent
// steps, each = the dot products of every candidate row with the newest pivot, a running maximum per row ("how close is my
// nearest pivot"), and the arg-min of that maximum. One 1024-thread workgroup per cloud keeps the running maxima in LDS and
// walks all P steps in one launch (the host version: 5 launches per step). Candidates = every `stride`-th row.
namespace {

__global__ __launch_bounds__(1024) void fps_pivots_kernel(const float* __restrict__ X, int N, int stride, int P,
                                                          int* __restrict__ picks, float* __restrict__ picked) {
    constexpr int D = 128, MAXC = 4096;
    __shared__ float closest[MAXC];
    __shared__ __attribute__((aligned(16))) float pv[D];
    __shared__ float red_v[16];
    __shared__ int red_i[16];
    __shared__ int cur;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = tid >> 3, sub = tid & 7;                  // 8 lanes per candidate row, 16 features each
    const int cloud = blockIdx.x;
    const float* Xc = X + (size_t)cloud * N * D;
    const int Ns = (N + stride - 1) / stride;
    if (tid == 0) cur = 0;
    __syncthreads();
    for (int j = 0; j < P; ++j) {
        const int c = cur;
        if (tid < D) {
            const float v = Xc[(size_t)c * stride * D + tid];
            pv[tid] = v;
            picked[((size_t)cloud * P + j) * D + tid] = v;
        }
        if (tid == 0) picks[(size_t)cloud * P + j] = c * stride;
        __syncthreads();
        f32x4 p4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) p4[u] = *(const f32x4*)(pv + 16 * sub + 4 * u);
        float best_v = 3.0e38f;
        int best_i = 0x7fffffff;
        for (int r = grp; r < Ns; r += 128) {
            const float* row = Xc + (size_t)r * stride * D + 16 * sub;
            float d = 0.f;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const f32x4 x = *(const f32x4*)(row + 4 * u);
                d = fmaf(x[0], p4[u][0], fmaf(x[1], p4[u][1], fmaf(x[2], p4[u][2], fmaf(x[3], p4[u][3], d))));
            }
            d += __shfl_xor(d, 1, 64);
            d += __shfl_xor(d, 2, 64);
            d += __shfl_xor(d, 4, 64);
            if (sub == 0) {
                const float cl = j == 0 ? d : fmaxf(closest[r], d);
                closest[r] = cl;
                if (cl < best_v) { best_v = cl; best_i = r; }
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(best_v, off, 64);
            const int oi = __shfl_xor(best_i, off, 64);
            if (ov < best_v || (ov == best_v && oi < best_i)) { best_v = ov; best_i = oi; }
        }
        if (lane == 0) { red_v[wave] = best_v; red_i[wave] = best_i; }
        __syncthreads();
        if (tid == 0) {
            float bv = red_v[0];
            int bi = red_i[0];
            for (int w = 1; w < 16; ++w)
                if (red_v[w] < bv || (red_v[w] == bv && red_i[w] < bi)) { bv = red_v[w]; bi = red_i[w]; }
            cur = bi < Ns ? bi : 0;                        // rows with NaN never compare smaller: stay inside the cloud
        }
        __syncthreads();
    }
}

}  // namespace

// X [B,N,128] unit rows -> picks [B,P] (row indices, multiples of stride; the first is row 0) and picked [B,P,128] (those
// rows): greedy farthest-point selection among rows 0, stride, 2 stride, ... (at most 4096 candidates per cloud).
extern "C" int sed_fps_pivots_f32(int B, int N, int d, int stride, int P, const float* X, int* picks, float* picked,
                                  hipStream_t stream) {
    if (B <= 0 || N <= 0 || stride <= 0 || P <= 0 || !X || !picks || !picked) return SED_EINVAL;
    if (d != 128 || (N + stride - 1) / stride > 4096 || P > (N + stride - 1) / stride) return SED_EUNSUPPORTED;
    fps_pivots_kernel<<<B, 1024, 0, stream>>>(X, N, stride, P, picks, picked);
    SED_LAUNCH_CHECK();
    return SED_OK;
}
