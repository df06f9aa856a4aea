// First-layer kNN graph (metric Dp (1 + W Dn) on xyz + normals, /root/reference/src/PointNet.py:90-137) on a Morton-ORDERED
// copy of the cloud (round 6). Included by knn_fused.hip inside its anonymous namespace; same candidate lists and finalize kernel
// as knn_pn_sweep_kernel, same metric instruction for instruction => the same neighbours bit for bit, for every permutation.
//
// In 3-d the order does what it cannot do for the 64-d feature graphs: a 32-row tile of a Morton order is
// a small box, Dp (1 + W Dn) >= Dp f with f = 1 + W (2 - 2 |n_i||n_j|) (= 1 for unit normals), and Dp >= the squared distance
// between the boxes of a wave's 64 queries and of the key tile. So
//   sweep 1: the threshold of a query = the k-th smallest metric value among the 256 rows of its OWN block (its spatial
//            neighbourhood: 8 tiles instead of every other tile of the cloud), exact over 4 x 32 bucket slots;
//   sweep 2: a wave visits only the key tiles whose box can reach its largest threshold -- a few per cent of the cloud.
// No LDS and no barrier: the keys' coordinates, normals, squared norms and original indices come through the scalar cache
// (uniform addresses), a wave is independent of the others of its workgroup.
#pragma once

// xs [B,8,N]: channels 0-5 of the rows in order, 6 = |xyz|^2 (the sweeps' own formula), 7 = the original index (as bits);
// tmeta [B,ntiles,8]: box min xyz, box max xyz, largest |xyz|^2, largest |n|^2 of every 32-row tile
__global__ __launch_bounds__(256) void pn_gather_kernel(const float* __restrict__ x6, const int* __restrict__ perm, int N,
                                                        float* __restrict__ xs) {
#pragma clang fp contract(off)
    const int cloud = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
    if (j >= N) return;
    const float* xc = x6 + (size_t)cloud * 6 * N;
    float* xo = xs + (size_t)cloud * 8 * N;
    const int src = perm[(size_t)cloud * N + j];
    const float a0 = xc[src], a1 = xc[N + src], a2 = xc[2 * N + src];
    xo[j] = a0; xo[N + j] = a1; xo[2 * N + j] = a2;
    xo[3 * N + j] = xc[3 * N + src]; xo[4 * N + j] = xc[4 * N + src]; xo[5 * N + j] = xc[5 * N + src];
    xo[6 * N + j] = fmaf(a2, a2, fmaf(a0, a0, a1 * a1));
    xo[7 * N + j] = __int_as_float(src);
}
__global__ __launch_bounds__(256) void pn_tilemeta_kernel(const float* __restrict__ xs, int N, int ntiles, size_t total,
                                                          float* __restrict__ tmeta) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const size_t cloud = i / ntiles;
    const int tile = (int)(i % ntiles);
    const float* xo = xs + cloud * 8 * N;
    const int n = N - tile * 32 < 32 ? N - tile * 32 : 32;
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f}, xm = 0.f, nm = 0.f;
    bool bad = false;
    for (int r = 0; r < n; ++r) {
        const int j = tile * 32 + r;
        float n2 = 0.f;
        for (int c = 0; c < 3; ++c) {
            const float v = xo[(size_t)c * N + j], w = xo[(size_t)(3 + c) * N + j];
            bad |= !(fabsf(v) < 1.0e18f) || !(fabsf(w) < 1.0e18f);
            lo[c] = fminf(lo[c], v); hi[c] = fmaxf(hi[c], v);
            n2 += w * w;
        }
        xm = fmaxf(xm, xo[(size_t)6 * N + j]);
        nm = fmaxf(nm, n2);
    }
    float* o = tmeta + i * 8;
    if (bad) {                                                 // a non-finite row: the tile's box is everything, it is never skipped
        for (int c = 0; c < 3; ++c) { o[c] = -3.0e38f; o[3 + c] = 3.0e38f; }
        o[6] = 0.f; o[7] = 0.f;
    } else {
        for (int c = 0; c < 3; ++c) { o[c] = lo[c]; o[3 + c] = hi[c]; }
        o[6] = xm; o[7] = nm;
    }
}

// PASS 1: thresholds from the block's own tiles; PASS 2: candidates from the tiles a wave's box can reach. One thread per query
// POSITION; Tbuf / lists / counts are indexed by position (knn_finalize_kernel writes row perm[position]).
// nfull = 32 (N / 32): keys whose ORIGINAL index is >= nfull sat in the ragged last tile of the unordered sweep, which evaluates
// 1 + W Dn with two roundings instead of one fma (knn_pn_sweep_kernel) -- the factor follows the key's original index here.
template <int PASS>
__global__ __launch_bounds__(256) void knn_pn_ord_kernel(const float* __restrict__ xs, const float* __restrict__ tmeta, int N,
                                                         int k, float W, uint32_t* __restrict__ Tbuf, Cand* __restrict__ lists,
                                                         int* __restrict__ counts, int* __restrict__ overflow) {
#pragma clang fp contract(off)
    constexpr int M = 4;                                       // 4 x 32 bucket slots >= the 256 rows of a block, two per slot
    const int cloud = blockIdx.y, tid = threadIdx.x;
    const float* xc = xs + (size_t)cloud * 8 * N;
    const int qi = blockIdx.x * 256 + tid;
    const int qc = qi < N ? qi : N - 1;
    const float p0 = xc[qc], p1 = xc[N + qc], p2 = xc[2 * N + qc];
    const float n0 = xc[3 * N + qc], n1 = xc[4 * N + qc], n2 = xc[5 * N + qc];
    const float xxi = fmaf(p2, p2, fmaf(p0, p0, p1 * p1));
    const int ntiles = (N + 31) >> 5;
    const int nfull = N & ~31;

    // the metric of this lane's query against the 32 keys of a tile (uniform addresses: 8 keys of a channel per s_load_dwordx8) --
    // knn_pn_sweep_kernel's lines; fn(r, dv, orig) for every key r of the tile that exists
    auto for_keys = [&](int tile, auto&& fn) {
#pragma unroll
        for (int r0 = 0; r0 < 32; r0 += 8) {
            const int j0 = tile * 32 + r0;
            if (j0 >= N) break;
            const bool whole = j0 + 8 <= N;
            f32x8u kv[8];
            if (whole) {
#pragma unroll
                for (int c = 0; c < 8; ++c) kv[c] = *(const f32x8u*)(xc + (size_t)c * N + j0);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                float k0, k1, k2, k3, k4, k5, kx, ko;
                if (whole) {
                    k0 = kv[0][u]; k1 = kv[1][u]; k2 = kv[2][u]; k3 = kv[3][u]; k4 = kv[4][u]; k5 = kv[5][u]; kx = kv[6][u]; ko = kv[7][u];
                } else {
                    if (j0 + u >= N) break;
                    const int j = j0 + u;
                    k0 = xc[j]; k1 = xc[N + j]; k2 = xc[2 * N + j]; k3 = xc[3 * N + j]; k4 = xc[4 * N + j]; k5 = xc[5 * N + j];
                    kx = xc[6 * N + j]; ko = xc[7 * N + j];
                }
                const int orig = __float_as_int(ko);
                const float dotp = fmaf(p2, k2, fmaf(p1, k1, p0 * k0));
                const float dotn = fmaf(n2, k5, fmaf(n1, k4, n0 * k3));
                const float dp = (kx - 2.0f * dotp) + xxi;               // (xx_j - inner) + xx_i  (:109)
                const float dn = 2.0f - 2.0f * dotn;                     // :112
                fn(r0 + u, dp * (orig >= nfull ? 1.0f + dn * W : fmaf(W, dn, 1.0f)), orig);      // :115
            }
        }
    };

    if (PASS == 1) {
        float bm[M][32];                                       // per slot r: the M smallest values of the keys at tile position r
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
            for (int r = 0; r < 32; ++r) bm[i][r] = 3.0e38f;
        int t0 = 8 * (int)blockIdx.x;
        if (t0 + 8 > ntiles) t0 = ntiles - 8 > 0 ? ntiles - 8 : 0;
        const int t1 = t0 + 8 < ntiles ? t0 + 8 : ntiles;
        for (int tile = t0; tile < t1; ++tile)
            for_keys(tile, [&](int r, float dv, int) {
                float v = dv;
#pragma unroll
                for (int i = 0; i < M; ++i) {
                    const float lo_ = sed_vmin(bm[i][r], v);
                    if (i + 1 < M) v = sed_vmax(bm[i][r], v);
                    bm[i][r] = lo_;
                }
            });
        // k-th smallest of the 128 slots by bisection on order-preserving keys (kept in the same registers)
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
            for (int r = 0; r < 32; ++r) bm[i][r] = __uint_as_float(bm[i][r] >= 3.0e38f ? 0xFFFFFFFFu : f32_sortable(bm[i][r]));
        uint32_t lo = 0, hiv = 0xFFFFFFFFu;
        for (int it = 0; it < 32; ++it) {
            const uint32_t mid = lo + ((hiv - lo) >> 1);
            int c = 0;
#pragma unroll
            for (int i = 0; i < M; ++i)
#pragma unroll
                for (int r = 0; r < 32; ++r) c += __float_as_uint(bm[i][r]) <= mid ? 1 : 0;
            if (lo < hiv) { if (c >= k) hiv = mid; else lo = mid + 1; }
        }
        if (qi < N) Tbuf[(size_t)cloud * N + qi] = lo;
        return;
    }

    const uint32_t T = Tbuf[(size_t)cloud * N + qc];
    const float Tf = T == 0xFFFFFFFFu ? __builtin_inff() : sortable_f32(T);
    Cand* mylist = lists + ((size_t)cloud * N + qc) * 2 * CAPL;          // one thread owns both halves
    int cnt = 0;
    // the wave's box, largest |xyz|^2, largest |n|^2 and largest threshold (out-of-range lanes repeat the cloud's last row)
    auto wmax = [&](float v) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
        return v;
    };
    const bool qfinite = fabsf(p0) < 1.0e18f && fabsf(p1) < 1.0e18f && fabsf(p2) < 1.0e18f;
    const float blo0 = -wmax(-p0), blo1 = -wmax(-p1), blo2 = -wmax(-p2);
    const float bhi0 = wmax(p0), bhi1 = wmax(p1), bhi2 = wmax(p2);
    const float qxm = wmax(xxi), qnm = wmax(fmaf(n2, n2, fmaf(n0, n0, n1 * n1)));
    const float Tmax = wmax(Tf);
    // a wave with a non-finite query or threshold, or a negative W (the factor is then no lower bound), skips nothing
    const bool can_skip = __builtin_amdgcn_ballot_w64(!qfinite || !(Tf < 3.0e38f)) == 0ull && W >= 0.f && qnm < 1.0e18f;
    const float* tm = tmeta + (size_t)cloud * ntiles * 8;
    for (int tile = 0; tile < ntiles; ++tile) {
        if (can_skip) {
            const float* b = tm + (size_t)tile * 8;
            const float g0 = fmaxf(fmaxf(b[0] - bhi0, blo0 - b[3]), 0.f), g1 = fmaxf(fmaxf(b[1] - bhi1, blo1 - b[4]), 0.f),
                        g2 = fmaxf(fmaxf(b[2] - bhi2, blo2 - b[5]), 0.f);
            const float lb = g0 * g0 + g1 * g1 + g2 * g2;        // squared distance between the boxes (>= 0; 0 for an unbounded box)
            const float f = 1.0f + W * (2.0f - 2.0f * sqrtf(qnm * b[7]));
            // computed values may undercut the true Dp f by the cancellation error of (xx_j - 2 x.y) + xx_i: a few ulps of the norms
            const float reach = Tmax * 1.00001f + 1.0e-5f * (1.0f + W * (2.0f + 2.0f * sqrtf(qnm * b[7]))) * (qxm + b[6]);
            if (f > 0.f && lb * f * 0.99999f > reach) continue;
        }
        for_keys(tile, [&](int, float dv, int orig) {
            if (dv <= Tf) {
                const uint32_t key = f32_sortable(dv);
                const bool hit = key <= T;
                Cand c; c.key = key; c.idx = orig;
                if (hit && cnt < 2 * CAPL) mylist[cnt] = c;
                cnt += hit ? 1 : 0;
            }
        });
    }
    if (qi < N) {
        counts[((size_t)cloud * N + qi) * 2] = cnt < 2 * CAPL ? cnt : 2 * CAPL;
        counts[((size_t)cloud * N + qi) * 2 + 1] = 0;
        if (cnt > 2 * CAPL) *overflow = 1;
    }
}
