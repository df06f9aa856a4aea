// A tile-coherent row order for the feature-space kNN sweeps (knn_fused.hip): Morton order of the INPUT cloud.
//
// The reference has no counterpart (src/PointNet.py:62-87 materialises all N x N distances); this is scheduling only. The sweeps
// of knn_fused.hip work on 32-row key tiles against 32-query waves. In the cloud's input order the ~50 candidates of a query (keys
// within its threshold T) are spread over all 313 tiles, so every (wave, tile) pair holds a few and every element goes through the
// append test. Rows that are close in xyz + normals have similar EdgeConv features (layer l + 1's features are functions of a
// layer-l neighbourhood), so in a space-filling order of the INPUT the candidates of a wave's 32 queries sit in ~8 % of the key
// tiles (CPU emulation on the trained network's layer-2 / layer-3 features) and the other tiles are
// dismissed by one comparison per lane. The order is a function of the cloud alone, is shared by both models and both feature
// layers of a step, and NEVER enters a result: scores are per (query, key) pair, candidates carry their original index, ties go
// by original index -- any permutation gives the same neighbours bit for bit (tests/test_gpu_knn.py).
//
// One workgroup per cloud: bounding box of the 6 channels, 5 bits per channel interleaved into a 30-bit code, (code, index)
// sorted as one 64-bit word by a bitonic network in LDS (N <= 16384: 128 KiB).
#include "common.h"

namespace {

__global__ __launch_bounds__(1024) void spatial_order_kernel(const float* __restrict__ x6, int N, int M /* power of two >= N */,
                                                             int* __restrict__ perm) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long el[];      // [M]
    __shared__ float red[2][6][16];
    __shared__ float lo_s[6], sc_s[6];
    const int cloud = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* xc = x6 + (size_t)cloud * 6 * N;
    float lo[6], hi[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) { lo[c] = 3.0e38f; hi[c] = -3.0e38f; }
    for (int i = tid; i < N; i += 1024)
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const float v = xc[(size_t)c * N + i];
            if (v > -3.0e38f && v < 3.0e38f) { lo[c] = fminf(lo[c], v); hi[c] = fmaxf(hi[c], v); }     // finite values only
        }
#pragma unroll
    for (int c = 0; c < 6; ++c) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            lo[c] = fminf(lo[c], __shfl_xor(lo[c], off, 64));
            hi[c] = fmaxf(hi[c], __shfl_xor(hi[c], off, 64));
        }
        if (lane == 0) { red[0][c][wave] = lo[c]; red[1][c][wave] = hi[c]; }
    }
    __syncthreads();
    if (tid < 6) {
        float a = 3.0e38f, b = -3.0e38f;
        for (int w = 0; w < 16; ++w) { a = fminf(a, red[0][tid][w]); b = fmaxf(b, red[1][tid][w]); }
        const float ext = b - a;
        lo_s[tid] = a;
        sc_s[tid] = (ext > 0.f && ext < 3.0e38f) ? 32.0f / ext : 0.f;
    }
    __syncthreads();
    for (int i = tid; i < M; i += 1024) {
        unsigned long long v = ~0ull;
        if (i < N) {
            uint32_t code = 0;
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                float t = (xc[(size_t)c * N + i] - lo_s[c]) * sc_s[c];
                t = fminf(fmaxf(t, 0.f), 31.f);                         // NaN -> 0
                const uint32_t q = (uint32_t)(int)t;
#pragma unroll
                for (int b = 0; b < 5; ++b) code |= ((q >> b) & 1u) << (6 * b + c);
            }
            v = ((unsigned long long)code << 32) | (unsigned)i;
        }
        el[i] = v;
    }
    __syncthreads();
    for (int k = 2; k <= M; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int idx = tid; idx < (M >> 1); idx += 1024) {
                const int l = idx & (j - 1);
                const int i = ((idx - l) << 1) | l, p = i | j;
                const unsigned long long a = el[i], b = el[p];
                const bool up = (i & k) == 0;
                if ((a > b) == up) { el[i] = b; el[p] = a; }
            }
            __syncthreads();
        }
    int* pc = perm + (size_t)cloud * N;
    for (int i = tid; i < N; i += 1024) pc[i] = (int)(uint32_t)el[i];
}

}  // namespace

extern "C" int sed_spatial_order_max_points(void) { return 16384; }

// x6 [B,6,N] channel-major (the network's input) -> perm [B,N] int32: perm[b][j] = the index of the point that comes j-th in the
// Morton order of cloud b's (xyz, normal) bounding box; a permutation of 0 .. N-1 for every input (non-finite coordinates sort
// into cell 0). Scheduling aid of sed_knn_fused_order_f32; no reference counterpart.
extern "C" int sed_spatial_order_f32(int B, int N, const float* x6, int* perm, hipStream_t stream) {
    if (B <= 0 || N <= 0 || !x6 || !perm) return SED_EINVAL;
    if (N > 16384) return SED_EUNSUPPORTED;
    int M = 64;
    while (M < N) M <<= 1;
    const int sm = M * (int)sizeof(unsigned long long);
    static std::atomic<unsigned long long> attr{0};          // devices whose dynamic-LDS limit has been raised (common.h)
    int attr_err = 0;
    if (sed_first_on_device(attr, &attr_err)) {
        const hipError_t e = hipFuncSetAttribute((const void*)spatial_order_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 8);
        if (e != hipSuccess) return (int)e;
        sed_mark_device(attr);
    } else if (attr_err) return attr_err;
    spatial_order_kernel<<<B, 1024, sm, stream>>>(x6, N, M, perm);
    SED_LAUNCH_CHECK();
    return SED_OK;
}
