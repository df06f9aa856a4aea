// Feature-space kNN on an ORDERED copy of the rows (round 6). Included by knn_fused.hip inside its anonymous namespace (shares Cand,
// CAPL and the candidate-list layout with the unordered sweeps; knn_finalize_kernel ranks the lists of either form).
//
// Replaces /root/reference/src/PointNet.py:62-87 like the unordered form and returns the same neighbours bit for bit: scores
// belong to (query, key) pairs and are computed by the same instruction sequence (split16.h), candidates carry their ORIGINAL
// index, ties go by it. What the order buys is that almost all of a wave's work can be dismissed early:
//
// In the caller's order the ~30 keys within a query's threshold are spread over all 313 key tiles of a 10 000-point cloud, so
// every (32-query wave, 32-key tile) pair must be scored exactly (3 fp16 MFMAs per 16 features) and tested element by element,
// twice (threshold sweep, candidate sweep). In a Morton order of the INPUT cloud (spatial_order.hip) rows that are close in
// xyz + normals are close in the EdgeConv features too: on the trained network's layer-2 / layer-3 features only ~12 % of the
// (wave, tile) pairs hold a score within the wave's thresholds (CPU emulation and device counters).
//
//   bound sweep  (knn_ord_bound_kernel): every key tile is scored with the HEAD product only (h_x . h_y: 1 MFMA per 16 features
//       instead of 3) -- good to 2^-9 sqrt(|x_i|^2 |x_j|^2) -- and dismissed when even score + that margin cannot reach the RUNNING
//       bound R of any of the wave's queries (R = the threshold the buckets seen so far give, an upper bound of the final one; the
//       block's own four tiles are visited first, after which R is already close). Only the remaining "listed" tiles are scored
//       exactly; only they feed the buckets (any subset of a row's scores bounds its k-th smallest from above, and the small
//       scores are exactly the listed ones) and get a bit in the wave's tile map. At the end: the thresholds T and the block's
//       list of (tile, 4-bit wave mask) entries.
//   collect sweep (knn_ord_collect_kernel): the block walks ITS list (~20-25 % of the tiles), a wave scores a staged tile only
//       if its bit is set, and appends the scores <= T to the lane-private candidate lists exactly like the unordered sweep 2.
//
// Every dismissal is a proven superset test (the element-wise test of the listed tiles is the unordered one), so T is a valid
// bound and the candidate sets contain every score <= T: knn_finalize_kernel's (score, index) ranks -- the output -- are those
// of the unordered form.
#pragma once

constexpr int ORD_MAXTILES = 512;                 // N <= 16384 (sed_spatial_order_max_points)

// sqrt of the largest squared norm of every 32-row tile (the key side of the head product's error margin)
__global__ __launch_bounds__(256) void ord_tilemax_kernel(const float* __restrict__ xxo, int N, int ntiles, size_t total,
                                                          float* __restrict__ tsq) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const size_t cloud = i / ntiles;
    const int tile = (int)(i % ntiles);
    const float* p = xxo + cloud * N + (size_t)tile * 32;
    const int n = N - tile * 32 < 32 ? N - tile * 32 : 32;
    float m = 0.f;
    bool bad = false;
    for (int j = 0; j < n; ++j) {
        const float v = p[j];
        bad |= !(v >= 0.f && v < 3.0e38f);
        m = fmaxf(m, v);
    }
    tsq[i] = bad ? __builtin_inff() : sqrtf(m);                 // a non-finite row: the tile is never dismissed
}

// shared by the two sweeps: stage one 32-row tile of the row image (+ the rows' squared norms and 2^-e) through registers
template <int NT>
struct OrdTile {
    static constexpr int D = 32 * NT, LDX = D + 4, C4 = D / 4;
    f32x4 v[NT];
    float sxx, sck;
    __device__ __forceinline__ void load(const float* Xc, const float* xxc, const float* invc, int N, int tile, int tid) {
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            const int i = tid + 256 * u;
            const int row = i / C4, c4 = i % C4;
            const int key = tile * 32 + row;
            v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (key < N) v[u] = *(const f32x4*)(Xc + (size_t)key * D + 4 * c4);
        }
        if (tid < 32) {
            const int key = tile * 32 + tid;
            sxx = key < N ? xxc[key] : 0.f;
            sck = key < N ? invc[key] : 0.f;
        }
    }
    __device__ __forceinline__ void store(float* lds, float* xxs, float* cks, int tid) const {
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            const int i = tid + 256 * u;
            const int row = i / C4, c4 = i % C4;
            *(f32x4*)(&lds[row * LDX + 4 * c4]) = v[u];
        }
        if (tid < 32) { xxs[tid] = sxx; cks[tid] = sck; }
    }
};

// t[r] = (-xx_j) - inner, inner = -2 x_i.x_j (PointNet.py:76-78) for this lane's 16 accumulator rows; the unscale by the two rows'
// powers of two is exact, so the line is ONE rounding however it is contracted -- the reference's addition
__device__ __forceinline__ void ord_scores(f32x16& s, float two_cq, const float* xxs, const float* cks, int hi) {
    f32x4 xk4[4], ck4[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        xk4[g] = *(const f32x4*)&xxs[8 * g + 4 * hi];
        ck4[g] = *(const f32x4*)&cks[8 * g + 4 * hi];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float dot2 = (s[r] * two_cq) * ck4[r >> 2][r & 3];
        s[r] = __fadd_rn(-xk4[r >> 2][r & 3], dot2);
    }
}
__device__ __forceinline__ float ord_max16(const f32x16& s) {
    float m = sed_vmax_acc(s[0], s[1]);
#pragma unroll
    for (int r = 2; r < 16; ++r) m = sed_vmax_acc(m, s[r]);
    return m;
}
// a score t can belong to a distance -fl(t - xq) <= bound only if t >= this (fl is monotone; the slack is 16 roundings of the
// operands' magnitudes, the three roundings involved stay far inside it). A non-finite bound dismisses nothing.
__device__ __forceinline__ float ord_reach(float xq, float bound) {
    return bound < __builtin_inff() ? (xq - bound) - 1.0e-6f * (fabsf(xq) + fabsf(bound)) : -__builtin_inff();
}

template <int NT, int M>
__global__ __launch_bounds__(256, 2) void knn_ord_bound_kernel(const float* __restrict__ X, const float* __restrict__ xx,
                                                               const float* __restrict__ inv, const float* __restrict__ tsq,
                                                               int N, int k, uint32_t* __restrict__ Tbuf,
                                                               unsigned short* __restrict__ blist, int* __restrict__ bcount) {
    using St = OrdTile<NT>;
    constexpr int D = St::D, LDX = St::LDX;
    __shared__ __attribute__((aligned(16))) float lds[2][32 * LDX];
    __shared__ __attribute__((aligned(16))) float xxs[2][32];
    __shared__ __attribute__((aligned(16))) float cks[2][32];
    __shared__ uint32_t fl[4][ORD_MAXTILES / 32];
    __shared__ int wsum[4];

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, hi = lane >> 5;
    int bxi;
    const int cloud = sed_xcd_cloud_block(&bxi);
    const float* Xc = X + (size_t)cloud * N * D;
    const float* xxc = xx + (size_t)cloud * N;
    const float* invc = inv + (size_t)cloud * N;
    const int ntiles = (N + 31) >> 5;
    const float* tsqc = tsq + (size_t)cloud * ntiles;
    const int qrow = bxi * 128 + wave * 32 + li;
    const int qrow_c = qrow < N ? qrow : N - 1;
    for (int i = tid; i < 4 * (ORD_MAXTILES / 32); i += 256) (&fl[0][0])[i] = 0u;

    h16x8 qh[2 * NT], ql[2 * NT];
    split_load_query<NT>((const h16*)Xc + (size_t)qrow_c * 2 * D, hi, qh, ql);
    const float two_cq = 2.0f * invc[qrow_c];
    const float xq = xxc[qrow_c];
    // margin of the head product: the dropped terms l.h + h.l + l.l are <= 2^-10 (1 + 2^-10) |x_i||x_j| per dot product (|l| <=
    // 2^-11 |h| element by element, Cauchy-Schwarz), twice that in 2 x_i.x_j; 5 % on top covers the accumulation roundings of
    // both products and of xx (each ~1e-6 of the same magnitude)
    const float emul = 1.05f * 0.001953125f * sqrtf(xq);

    float bm[M][16];                                            // the M largest t = 2 x_i.x_j - xx_j per bucket (see knn_sweep_kernel)
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) bm[i][r] = -3.0e38f;
    auto kth_bound = [&](int steps) -> uint32_t {               // upper end of the bisection: a valid bound after any number of steps
        uint32_t bk[M][16];
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) bk[i][r] = bm[i][r] <= -3.0e38f ? 0xFFFFFFFFu : f32_sortable(-__fsub_rn(bm[i][r], xq));
        uint32_t lo = 0, hiv = 0xFFFFFFFFu;
        for (int it = 0; it < steps; ++it) {
            const uint32_t mid = lo + ((hiv - lo) >> 1);
            int c = 0;
#pragma unroll
            for (int i = 0; i < M; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) c += bk[i][r] <= mid ? 1 : 0;
            c += __shfl_xor(c, 32, 64);
            if (lo < hiv) { if (c >= k) hiv = mid; else lo = mid + 1; }
        }
        return steps >= 32 ? lo : hiv;
    };

    // every tile once, the block's own four first
    const int own0 = 4 * bxi, nown = ntiles - own0 < 4 ? ntiles - own0 : 4;
    auto tile_at = [&](int i) { return i < nown ? own0 + i : (i - nown < own0 ? i - nown : i); };
    float Rthr = -__builtin_inff();
    St st;
    st.load(Xc, xxc, invc, N, tile_at(0), tid);
    st.store(lds[0], xxs[0], cks[0], tid);
    __syncthreads();
    int cur = 0;
    for (int vi = 0; vi < ntiles; ++vi) {
        const int tile = tile_at(vi);
        if (vi + 1 < ntiles) st.load(Xc, xxc, invc, N, tile_at(vi + 1), tid);
        const uint8_t* krow = (const uint8_t*)(lds[cur] + li * LDX);
        const bool ragged = (tile == ntiles - 1) && (N & 31);
        bool listed = ragged || vi < nown;
        if (!listed) {
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 2 * NT; ++ks) s = mfma16(*(const h16x8*)(krow + ks * 32 + hi * 16), qh[ks], s);
            ord_scores(s, two_cq, xxs[cur], cks[cur], hi);
            const float reach = fmaf(emul, tsqc[tile], ord_max16(s));
            listed = __builtin_amdgcn_ballot_w64(reach >= Rthr) != 0ull;
        }
        if (listed) {
            f32x16 s = split_tile_keys_on_rows<NT>(krow, hi, qh, ql);
            ord_scores(s, two_cq, xxs[cur], cks[cur], hi);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const bool pad = ragged && tile * 32 + mfma_row(r, hi) >= N;
                float v = pad ? -3.0e38f : s[r];
#pragma unroll
                for (int i = 0; i < M; ++i) {
                    const float keep = sed_vmax(bm[i][r], v);
                    if (i + 1 < M) v = sed_vmin(bm[i][r], v);
                    bm[i][r] = keep;
                }
            }
            if (lane == 0) fl[wave][tile >> 5] |= 1u << (tile & 31);
        }
        if (vi == nown - 1 || (vi >= nown && ((vi - nown) & 63) == 63)) {
            const uint32_t R = kth_bound(16);                   // 16 steps: within 2^-7 of the exact k-th bucket value
            if (R != 0xFFFFFFFFu) Rthr = ord_reach(xq, sortable_f32(R));
        }
        if (vi + 1 < ntiles) st.store(lds[cur ^ 1], xxs[cur ^ 1], cks[cur ^ 1], tid);
        __syncthreads();
        cur ^= 1;
    }
    const uint32_t T = kth_bound(32);
    if (qrow < N && hi == 0) Tbuf[(size_t)cloud * N + qrow] = T;

    // the block's list: (tile << 4 | wave mask) in tile order
    const size_t block = (size_t)cloud * gridDim.x + bxi;
    unsigned short* bl = blist + block * ntiles;
    int base = 0;
    for (int t0 = 0; t0 < ntiles; t0 += 256) {
        const int t = t0 + tid;
        uint32_t m4 = 0;
        if (t < ntiles)
#pragma unroll
            for (int w = 0; w < 4; ++w) m4 |= ((fl[w][t >> 5] >> (t & 31)) & 1u) << w;
        const unsigned long long b = __builtin_amdgcn_ballot_w64(m4 != 0);
        if (lane == 0) wsum[wave] = __builtin_popcountll(b);
        __syncthreads();
        int pos = base + __builtin_popcountll(b & ((1ull << lane) - 1ull));
        for (int w = 0; w < wave; ++w) pos += wsum[w];
        if (m4) bl[pos] = (unsigned short)((t << 4) | m4);
        base += wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
    if (tid == 0) bcount[block] = base;
}

template <int NT>
__global__ __launch_bounds__(256, 2) void knn_ord_collect_kernel(const float* __restrict__ X, const float* __restrict__ xx,
                                                                 const float* __restrict__ inv, int N,
                                                                 const uint32_t* __restrict__ Tbuf, Cand* __restrict__ lists,
                                                                 int* __restrict__ counts, int* __restrict__ overflow,
                                                                 const int* __restrict__ perm,
                                                                 const unsigned short* __restrict__ blist,
                                                                 const int* __restrict__ bcount) {
    using St = OrdTile<NT>;
    constexpr int D = St::D, LDX = St::LDX;
    __shared__ __attribute__((aligned(16))) float lds[2][32 * LDX];
    __shared__ __attribute__((aligned(16))) float xxs[2][32];
    __shared__ __attribute__((aligned(16))) float cks[2][32];

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, hi = lane >> 5;
    int bxi;
    const int cloud = sed_xcd_cloud_block(&bxi);
    const float* Xc = X + (size_t)cloud * N * D;
    const float* xxc = xx + (size_t)cloud * N;
    const float* invc = inv + (size_t)cloud * N;
    const int* permc = perm + (size_t)cloud * N;
    const int ntiles = (N + 31) >> 5;
    const int qrow = bxi * 128 + wave * 32 + li;
    const int qrow_c = qrow < N ? qrow : N - 1;
    const size_t block = (size_t)cloud * gridDim.x + bxi;
    const unsigned short* bl = blist + block * ntiles;
    const int n = bcount[block];

    h16x8 qh[2 * NT], ql[2 * NT];
    split_load_query<NT>((const h16*)Xc + (size_t)qrow_c * 2 * D, hi, qh, ql);
    const float two_cq = 2.0f * invc[qrow_c];
    const float xq = xxc[qrow_c];
    const uint32_t T = Tbuf[(size_t)cloud * N + qrow_c];
    const float Tf = T == 0xFFFFFFFFu ? __builtin_inff() : sortable_f32(T);      // fewer than k bucket values: take everything
    const float thr = ord_reach(xq, Tf);
    Cand* mylist = lists + (((size_t)cloud * N + qrow_c) * 2 + hi) * CAPL;
    int cnt = 0;

    if (n > 0) {
        St st;
        st.load(Xc, xxc, invc, N, (int)bl[0] >> 4, tid);
        st.store(lds[0], xxs[0], cks[0], tid);
    }
    __syncthreads();
    int cur = 0;
    for (int i = 0; i < n; ++i) {
        const int e = (int)bl[i];
        const int tile = e >> 4;
        St st;
        if (i + 1 < n) st.load(Xc, xxc, invc, N, (int)bl[i + 1] >> 4, tid);
        if ((e >> wave) & 1) {
            f32x16 s = split_tile_keys_on_rows<NT>((const uint8_t*)(lds[cur] + li * LDX), hi, qh, ql);
            ord_scores(s, two_cq, xxs[cur], cks[cur], hi);
            const bool ragged = (tile == ntiles - 1) && (N & 31);
            if (ragged || __builtin_amdgcn_ballot_w64(ord_max16(s) >= thr) != 0ull) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int krow = mfma_row(r, hi);
                    const bool pad = ragged && tile * 32 + krow >= N;
                    const float dv = -__fsub_rn(s[r], xq);          // ... - xx_i ; distance = -score
                    if (dv <= Tf && !pad) {                          // one branch per value; inside it the append is predicated
                        const uint32_t key = f32_sortable(dv);
                        const bool hit = key <= T;
                        Cand c; c.key = key; c.idx = permc[tile * 32 + krow];
                        if (hit && cnt < CAPL) mylist[cnt] = c;
                        cnt += hit ? 1 : 0;
                    }
                }
            }
        }
        if (i + 1 < n) st.store(lds[cur ^ 1], xxs[cur ^ 1], cks[cur ^ 1], tid);
        __syncthreads();
        cur ^= 1;
    }
    if (qrow < N) {
        counts[((size_t)cloud * N + qrow) * 2 + hi] = cnt;
        if (cnt > CAPL) *overflow = 1;
    }
}
