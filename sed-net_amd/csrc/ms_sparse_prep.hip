// Preparation of the block-sparse mean-shift schedule (ms_iterate_d128_f16s_kernel, ms_iterate_f16.hip): everything that
// turns a cloud's unit rows X [N,128] into cluster-pure 32-row tiles with two reference vectors each. The arithmetic being
// scheduled is /root/reference/src/mean_shift.py:45-79; nothing here changes it -- any row order and any unit references are
// CORRECT, they only decide how many 32 x 32 blocks the iteration kernel can prove negligible and skip.
#include "common.h"


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
