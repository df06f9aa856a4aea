// Row order of the block-sparse mean-shift stage, round 5: a SPLIT TREE over the rows of a cloud instead of pivot groups.
//
// What the order is for (ms_sparse_f16.hip, ms_tiles.hip; mathematics /root/reference/src/mean_shift.py:56-77, :115-179): the iteration
// kernel, the bandwidth sweep and the NMS membership sweep all work on 32-row tiles of the sorted rows and skip a (query tile, key
// tile) block when caps around the tiles' reference directions prove that no pair in it is within the kernel's reach. How much is
// skipped is decided by how COMPACT the tiles are. Rounds 2-4 sorted rows by (super-group, nearest of 64 farthest-point pivots): tiles
// were cluster-pure, but inside a pivot group rows kept their arbitrary input order, so a tile sampled its whole group. On the
// reference's own embeddings of three bench clouds (tests/golden/f_64_emb.npz, CPU emulation of the kernel's cap test) 44-65 % of the
// blocks survived the test where 24-35 % of the PAIRS are within reach; recursive bisection of ALL rows along the direction towards
// the row farthest from a node's first row brings that to 30-45 % -- and more pivots made it worse (47-67 % at 128-256 pivots).
//
// The tree. A node is a range [s, e) of the current order, s a multiple of 32. A node of more than 32 rows is split: with r0 = its
// first row and a = the row of the node with the smallest dot product with r0 (farthest from it on the unit sphere; ties: the
// earliest position), rows are sorted by x . (a - r0) (ties: earlier position first) and the node is cut at a TILE BOUNDARY: among the
// boundaries between 1/8 and 7/8 of the node's tiles the one with the LARGEST GAP between the keys on its two sides (ties: the
// earliest). On a manifold-like embedding (what the bench's trained network produces) the gaps are all alike and the cut is as good as
// the median -- 30-45 % of the blocks survive either way --; on tight, well separated clusters (what a fully trained network's
// triplet loss aims for: tools/CPU emulation on synth.clustered_embedding) the median cuts through clusters and leaves tiles that
// mix two of them (15-20 % of the blocks survive where the pivot order keeps 11-13 %), while the largest gap falls BETWEEN clusters
// (11-15 %). Node boundaries therefore depend on the cloud: a per-tile node table (start, end of the node holding the tile) is carried
// from level to level. Levels: ceil(log2(tiles)) + 5 (14 at N = 10 000); a cloud whose nodes are all leaves already skips the
// level's work, a node still larger than a tile after the last level stays sorted along its last direction (any order is correct).
// Per level
//   tree_far_kernel   one workgroup per 32-row tile: the tile's candidate for a -> one 64-bit atomicMin per tile on the node's slot
//                     (order-free: the minimum of (dot, position) is the same whatever order the atomics arrive in),
//   tree_key_kernel   the sort keys x . (a - r0),
//   tree_sort_kernel  one workgroup per cloud: bitonic sort of (node, key, position) in LDS -> the new order; the cuts -> the new table.
// Everything is a deterministic function of the cloud alone (no floating-point atomics, no dependence on the batch). Leaves (<= 32
// rows) end up sorted along their parent's direction; prep_tile_refs_kernel (ms_sparse_prep.hip) then gives every tile the
// normalised means of its two 16-row halves as references.
#include "common.h"

namespace {

// node table of a cloud: nodes[2 t], nodes[2 t + 1] = start and end (row positions) of the node that holds tile t at the current level
// x_row . v for one row per 8 lanes (F = D / 8 features per lane); v given as the lane's slice
template <int D>
__device__ __forceinline__ float slice_dot(const float* __restrict__ row, const f32x4* v) {
    constexpr int F4 = D / 32;                                // float4s per lane
    float d = 0.f;
#pragma unroll
    for (int u = 0; u < F4; ++u) {
        const f32x4 x = *(const f32x4*)(row + 4 * u);
        d = fmaf(x[0], v[u][0], fmaf(x[1], v[u][1], fmaf(x[2], v[u][2], fmaf(x[3], v[u][3], d))));
    }
    d += __shfl_xor(d, 1, 64);
    d += __shfl_xor(d, 2, 64);
    d += __shfl_xor(d, 4, 64);
    return d;
}

// per 32-position tile: min over its rows of (x . x_first_of_node, position) -> atomicMin on far[cloud][s / 32]
template <int D>
__global__ __launch_bounds__(256) void tree_far_kernel(const float* __restrict__ X, const int* __restrict__ perm, int N,
                                                       const int* __restrict__ nodes, unsigned long long* __restrict__ far, int ntiles) {
    constexpr int F = D / 8, F4 = D / 32;
    __shared__ unsigned long long wmin[4];
    const int tile = blockIdx.x, cloud = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s = nodes[((size_t)cloud * ntiles + tile) * 2], e = nodes[((size_t)cloud * ntiles + tile) * 2 + 1];
    if (e - s <= 32) return;                                  // a leaf: nothing left to split
    const float* Xc = X + (size_t)cloud * N * D;
    const int* pc = perm + (size_t)cloud * N;
    const int i = tile * 32 + (tid >> 3), sub = tid & 7;
    const float* anchor = Xc + (size_t)pc[s] * D + F * sub;
    f32x4 v[F4];
#pragma unroll
    for (int u = 0; u < F4; ++u) v[u] = *(const f32x4*)(anchor + 4 * u);
    unsigned long long key = ~0ull;
    const float d = slice_dot<D>(Xc + (size_t)pc[i < N ? i : N - 1] * D + F * sub, v);
    if (i < N) key = ((unsigned long long)f32_sortable(d) << 32) | (unsigned)i;
#pragma unroll
    for (int off = 8; off < 64; off <<= 1) {                  // (the 8 lanes of a row agree; reduce over the wave's 8 rows)
        const unsigned long long o = __shfl_xor(key, off, 64);
        key = o < key ? o : key;
    }
    if (lane == 0) wmin[wave] = key;
    __syncthreads();
    if (tid == 0) {
        unsigned long long m = wmin[0];
        for (int w = 1; w < 4; ++w) m = wmin[w] < m ? wmin[w] : m;
        atomicMin(far + (size_t)cloud * ntiles + (s >> 5), m);
    }
}

// keys[i] = x_i . (x_a - x_first) for the rows of nodes that split at this level, 0 for rows of leaves (they keep their order)
template <int D>
__global__ __launch_bounds__(256) void tree_key_kernel(const float* __restrict__ X, const int* __restrict__ perm, int N,
                                                       const int* __restrict__ nodes, const unsigned long long* __restrict__ far,
                                                       int ntiles, float* __restrict__ keys) {
    constexpr int F = D / 8, F4 = D / 32;
    const int tile = blockIdx.x, cloud = blockIdx.y, tid = threadIdx.x;
    const int i = tile * 32 + (tid >> 3), sub = tid & 7;
    const int s = nodes[((size_t)cloud * ntiles + tile) * 2], e = nodes[((size_t)cloud * ntiles + tile) * 2 + 1];
    if (e - s <= 32) {
        if (i < N && sub == 0) keys[(size_t)cloud * N + i] = 0.f;
        return;
    }
    const float* Xc = X + (size_t)cloud * N * D;
    const int* pc = perm + (size_t)cloud * N;
    const int apos = (int)(far[(size_t)cloud * ntiles + (s >> 5)] & 0xffffffffull);
    const float* anchor = Xc + (size_t)pc[s] * D + F * sub;
    const float* arow = Xc + (size_t)pc[apos] * D + F * sub;
    f32x4 v[F4];
#pragma unroll
    for (int u = 0; u < F4; ++u) {
        const f32x4 a = *(const f32x4*)(arow + 4 * u), r0 = *(const f32x4*)(anchor + 4 * u);
        v[u] = f32x4{a[0] - r0[0], a[1] - r0[1], a[2] - r0[2], a[3] - r0[3]};
    }
    const float d = slice_dot<D>(Xc + (size_t)pc[i < N ? i : N - 1] * D + F * sub, v);
    if (i < N && sub == 0) keys[(size_t)cloud * N + i] = d;
}

// one workgroup per cloud: sort positions by (node start, key, position) -> perm_out[j] = perm_in[position that comes j-th]; then
// every node of more than one tile is cut at the tile boundary with the largest key gap between 1/8 and 7/8 of its tiles -> new table
__global__ __launch_bounds__(1024) void tree_sort_kernel(const float* __restrict__ keys, const int* __restrict__ perm_in, int N,
                                                         int ntiles, int M /* power of two >= N */, int* __restrict__ nodes,
                                                         int* __restrict__ perm_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long el[];      // [M]
    __shared__ int ns[512], ne[512];                          // the cloud's node table (ntiles <= 512)
    __shared__ unsigned long long best[512];                  // per node (slot = its start tile): (gap, 0xffff - boundary tile), largest wins
    __shared__ int any_split;
    const int cloud = blockIdx.x, tid = threadIdx.x;
    int* nc = nodes + (size_t)cloud * ntiles * 2;
    const int* pi = perm_in + (size_t)cloud * N;
    int* po = perm_out + (size_t)cloud * N;
    if (tid == 0) any_split = 0;
    __syncthreads();
    for (int t = tid; t < ntiles; t += 1024) {
        ns[t] = nc[2 * t];
        ne[t] = nc[2 * t + 1];
        best[t] = 0ull;
        if (ne[t] - ns[t] > 32) any_split = 1;
    }
    __syncthreads();
    if (!any_split) {                                         // every node of this cloud is a leaf already: the order stands
        for (int j = tid; j < N; j += 1024) po[j] = pi[j];
        return;
    }
    const float* kc = keys + (size_t)cloud * N;
    // Two ways to sort inside the nodes. BITONIC over all M words: 105 passes at M = 16 384, bound by the CU's LDS bandwidth (0.145 ms per
    // level whatever the nodes look like). RANK: every row counts the rows of its node that sort before it -- sum over nodes of n^2
    // comparisons on LDS keys that the lanes of a wave mostly share (broadcast reads) -- which is far less once the nodes are small
    // (from the 4th or 5th level on: ~10 us). Same result either way (the order (key, position) is total: keys compared as order-
    // preserving integers, NaNs included); the choice is made per cloud and level from the node table.
    __shared__ unsigned long long cost_s;
    if (tid == 0) cost_s = 0ull;
    __syncthreads();
    {
        unsigned long long c = 0ull;
        for (int t = tid; t < ntiles; t += 1024) c += (unsigned long long)(ne[t] - ns[t] > 32 ? 32 * (ne[t] - ns[t]) : 0);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
        if ((tid & 63) == 0 && c) atomicAdd(&cost_s, c);      // integer: order-free
    }
    __syncthreads();
    const bool by_rank = cost_s <= 4000000ull;
    uint32_t* ku = (uint32_t*)el;                            // RANK: [M] keys as order-preserving integers | [M] the same, sorted per node
    uint32_t* sk = ku + M;
    if (by_rank) {
        for (int i = tid; i < M; i += 1024) ku[i] = i < N ? f32_sortable(kc[i]) : 0xffffffffu;
        __syncthreads();
        for (int i = tid; i < N; i += 1024) {
            const int s = ns[i >> 5], e = ne[i >> 5];
            int r = i - s;                                    // rows of leaves keep their places
            const uint32_t ki = ku[i];
            if (e - s > 32) {
                // (key, position) as ONE 64-bit number: one comparison per row; 8 reads in flight (s is a multiple of 32, the tail of
                // the cloud's last node is padded with the largest key: never counted)
                const unsigned long long me = ((unsigned long long)ki << 32) | (unsigned)i;
                r = 0;
                for (int j0 = s; j0 < e; j0 += 8) {
                    uint32_t kj[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) kj[u] = ku[j0 + u];
#pragma unroll
                    for (int u = 0; u < 8; ++u) r += ((((unsigned long long)kj[u] << 32) | (unsigned)(j0 + u)) < me) ? 1 : 0;
                }
            }
            sk[s + r] = ki;
            po[s + r] = pi[i];
        }
        __syncthreads();
    } else {
        for (int i = tid; i < M; i += 1024) {
            unsigned long long v = ~0ull;
            if (i < N) v = ((unsigned long long)(ns[i >> 5] >> 5) << 46) | ((unsigned long long)f32_sortable(kc[i]) << 14) | (unsigned)i;
            el[i] = v;
        }
        __syncthreads();
        for (int k = 2; k <= M; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int idx = tid; idx < (M >> 1); idx += 1024) {
                    const int lo = idx & (j - 1);
                    const int i = ((idx - lo) << 1) | lo, p = i | j;
                    const unsigned long long a = el[i], b = el[p];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { el[i] = b; el[p] = a; }
                }
                __syncthreads();
            }
        for (int j = tid; j < N; j += 1024) po[j] = pi[(int)(el[j] & 0x3fffull)];
    }
    auto sorted_key = [&](int j) { return sortable_f32(by_rank ? sk[j] : (uint32_t)(el[j] >> 14)); };
    // the cuts: boundary in front of tile g, for every g inside a node of more than one tile
    for (int g = tid + 1; g < ntiles; g += 1024) {
        const int s = ns[g], e = ne[g];
        if (ns[g - 1] != s || e - s <= 32) continue;
        const int nt = (e - s + 31) >> 5, t = g - (s >> 5);
        const int lo = (nt + 7) >> 3, hi = (7 * nt) >> 3;      // ceil(nt / 8) .. floor(7 nt / 8); nt = 2: 1 .. 1
        if (t < (lo > 1 ? lo : 1) || t > (hi < nt - 1 ? hi : nt - 1)) continue;
        const float gap = sorted_key(32 * g) - sorted_key(32 * g - 1);      // >= 0: the keys are sorted inside the node
        atomicMax(&best[s >> 5], ((unsigned long long)f32_sortable(gap == gap ? gap : 0.f) << 16) | (unsigned)(0xffff - g));
    }
    __syncthreads();
    for (int g = tid; g < ntiles; g += 1024) {
        const int s = ns[g], e = ne[g];
        if (e - s <= 32) continue;
        int cut = 32 * (0xffff - (int)(best[s >> 5] & 0xffffull));
        //  the [lo, hi] boundary arithmetic above posts at least one candidate for every node that splits; should `best` ever
        // stay 0 the cut falls outside the node -- take the median tile boundary then instead of corrupting the node table
        if (cut <= s || cut >= e) cut = s + 32 * (((e - s + 31) >> 5) >> 1);
        if (32 * g < cut) { nc[2 * g] = s; nc[2 * g + 1] = cut; }
        else { nc[2 * g] = cut; nc[2 * g + 1] = e; }
    }
}

__global__ void tree_nodes_init_kernel(int* __restrict__ nodes, size_t tiles, int N) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < tiles) { nodes[2 * i] = 0; nodes[2 * i + 1] = N; }
}

__global__ void tree_iota_kernel(int* __restrict__ perm, size_t rows, int N) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < rows) perm[i] = (int)(i % (size_t)N);
}

// The two GROUPS of a tile's rows whose normalised means become its two references (prep_tile_refs_kernel: group A = the rows whose
// flag equals the first row's). A tile that straddles two clusters -- the cuts avoid that where they can, a 32-row grid cannot always
// -- would get two wide caps from its two 16-row halves (1.4 rad measured on separated blobs) and be visited by every query within
// reach of either cluster's far side; split where its rows really part, it gets two tight caps like the pivot order's border tiles.
// Per tile: r0 = its first row, a = its row farthest from r0, keys x . (a - r0); the 32 keys are ranked and the rows are divided at
// the largest gap between consecutive keys (a compact tile is divided somewhere, both caps stay inside the tile's own).
template <int D>
__global__ __launch_bounds__(256) void tree_tile_groups_kernel(const float* __restrict__ Xs, int N, int* __restrict__ scomp) {
    constexpr int F = D / 8, F4 = D / 32;
    __shared__ float kd[32], ks[32], srt[32];
    __shared__ int apos_s, cut_s;
    const int tile = blockIdx.x, cloud = blockIdx.y, tid = threadIdx.x;
    const int r = tid >> 3, sub = tid & 7;
    const float* Xc = Xs + (size_t)cloud * N * D;
    const int n = min(32, N - 32 * tile);                     // rows of this tile
    const int row = 32 * tile + (r < n ? r : n - 1);
    f32x4 v[F4];
#pragma unroll
    for (int u = 0; u < F4; ++u) v[u] = *(const f32x4*)(Xc + (size_t)(32 * tile) * D + F * sub + 4 * u);
    const float d0 = slice_dot<D>(Xc + (size_t)row * D + F * sub, v);
    if (sub == 0) kd[r] = d0;
    __syncthreads();
    if (tid == 0) {                                           // farthest from the first row: smallest dot product, earliest on ties
        int a = 0;
        for (int j = 1; j < n; ++j)
            if (kd[j] < kd[a]) a = j;
        apos_s = a;
    }
    __syncthreads();
    const float* arow = Xc + (size_t)(32 * tile + apos_s) * D + F * sub;
#pragma unroll
    for (int u = 0; u < F4; ++u) {
        const f32x4 a = *(const f32x4*)(arow + 4 * u);
        v[u] = f32x4{a[0] - v[u][0], a[1] - v[u][1], a[2] - v[u][2], a[3] - v[u][3]};
    }
    const float k = slice_dot<D>(Xc + (size_t)row * D + F * sub, v);
    if (sub == 0) ks[r] = k;
    __syncthreads();
    if (tid < 32 && tid < n) {                                // rank sort of the n keys (ties: earlier row first)
        int rank = 0;
        for (int j = 0; j < n; ++j) rank += (ks[j] < ks[tid] || (ks[j] == ks[tid] && j < tid)) ? 1 : 0;
        srt[rank] = ks[tid];
    }
    __syncthreads();
    if (tid == 0) {
        int cut = 0;                                          // divide between sorted keys cut and cut + 1
        float best = -1.f;
        for (int j = 0; j + 1 < n; ++j) {
            const float gap = srt[j + 1] - srt[j];
            if (gap > best) { best = gap; cut = j; }
        }
        cut_s = cut;
    }
    __syncthreads();
    if (tid < 32 && tid < n) scomp[(size_t)cloud * N + 32 * tile + tid] = ks[tid] > srt[cut_s] ? 1 : 0;
}

// Xs[i] = X[order[i]]
template <int D>
__global__ __launch_bounds__(256) void tree_gather_kernel(const float* __restrict__ X, const int* __restrict__ order, size_t rows, int N,
                                                          float* __restrict__ Xs) {
    constexpr int TPR = D / 4, RPB = 256 / TPR;
    if (threadIdx.x >= RPB * TPR) return;
    const size_t i = (size_t)blockIdx.x * RPB + threadIdx.x / TPR;
    if (i >= rows) return;
    const int l4 = threadIdx.x % TPR;
    const size_t cloud = i / N;
    *(f32x4*)(Xs + i * D + 4 * l4) = *(const f32x4*)(X + (cloud * N + order[i]) * D + 4 * l4);
}

}  // namespace

// workspace of ms_tree_order: two orders, the keys, the per-node slots
size_t ms_tree_workspace_bytes(int B, int N) {
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    const size_t nt = (size_t)(N + 31) / 32;
    return 2 * up((size_t)B * N * sizeof(int)) + up((size_t)B * N * sizeof(float)) + up((size_t)B * nt * sizeof(unsigned long long)) +
           up((size_t)B * nt * 2 * sizeof(int));
}

// X [B,N,d] (d = 128 / 160; unit rows) -> order [B,N] (sorted position -> row), Xs [B,N,d] the rows in that order, scomp [B,N] = which of
// its tile's two groups a row belongs to
int ms_tree_order(int B, int N, int d, const float* X, int* order, float* Xs, int* scomp, void* ws, hipStream_t stream) {
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    if (N > 16384) return SED_EUNSUPPORTED;
    const int ntiles = (N + 31) / 32;
    const size_t rows = (size_t)B * N;
    uint8_t* b = (uint8_t*)ws;
    int* pa = (int*)b; b += up(rows * sizeof(int));
    int* pb = (int*)b; b += up(rows * sizeof(int));
    float* keys = (float*)b; b += up(rows * sizeof(float));
    unsigned long long* far = (unsigned long long*)b; b += up((size_t)B * ntiles * sizeof(unsigned long long));
    int* nodes = (int*)b;
    int levels = 0;
    for (int t = ntiles; t > 1; t = (t + 1) / 2) ++levels;
    if (levels) levels += 5;                                  // cuts between 1/8 and 7/8 of a node: deeper than the balanced tree
    int M = 32;
    while (M < N) M <<= 1;
    const int sm = M * (int)sizeof(unsigned long long);
    static std::atomic<unsigned long long> attr{0};          // devices whose dynamic-LDS limit has been raised (common.h)
    int attr_err = 0;
    if (sed_first_on_device(attr, &attr_err)) {
        const hipError_t e = hipFuncSetAttribute((const void*)tree_sort_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 8);
        if (e != hipSuccess) return (int)e;
        sed_mark_device(attr);
    } else if (attr_err) return attr_err;
    // (the last level's result lands in `order` itself; levels alternate between the two workspace orders before it)
    tree_iota_kernel<<<(unsigned)((rows + 255) / 256), 256, 0, stream>>>(levels ? pa : order, rows, N);
    tree_nodes_init_kernel<<<(unsigned)(((size_t)B * ntiles + 255) / 256), 256, 0, stream>>>(nodes, (size_t)B * ntiles, N);
    const dim3 gt(ntiles, B);
    int* cur = pa;
    for (int l = 0; l < levels; ++l) {
        int* nxt = l + 1 == levels ? order : (cur == pa ? pb : pa);
        const hipError_t e = hipMemsetAsync(far, 0xff, (size_t)B * ntiles * sizeof(unsigned long long), stream);
        if (e != hipSuccess) return (int)e;
        if (d == 160) {
            tree_far_kernel<160><<<gt, 256, 0, stream>>>(X, cur, N, nodes, far, ntiles);
            tree_key_kernel<160><<<gt, 256, 0, stream>>>(X, cur, N, nodes, far, ntiles, keys);
        } else {
            tree_far_kernel<128><<<gt, 256, 0, stream>>>(X, cur, N, nodes, far, ntiles);
            tree_key_kernel<128><<<gt, 256, 0, stream>>>(X, cur, N, nodes, far, ntiles, keys);
        }
        tree_sort_kernel<<<B, 1024, sm, stream>>>(keys, cur, N, ntiles, M, nodes, nxt);
        cur = nxt;
    }
    if (d == 160) {
        tree_gather_kernel<160><<<(unsigned)((rows + 5) / 6), 256, 0, stream>>>(X, order, rows, N, Xs);
        tree_tile_groups_kernel<160><<<gt, 256, 0, stream>>>(Xs, N, scomp);
    } else {
        tree_gather_kernel<128><<<(unsigned)((rows + 7) / 8), 256, 0, stream>>>(X, order, rows, N, Xs);
        tree_tile_groups_kernel<128><<<gt, 256, 0, stream>>>(Xs, N, scomp);
    }
    SED_LAUNCH_CHECK();
    return SED_OK;
}
