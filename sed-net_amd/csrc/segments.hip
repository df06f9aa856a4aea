// Small per-point / per-segment kernels that glue the stages together on the device, so that the batched
// driver never leaves the GPU between the network, the clustering and the fits:
//   * row_normalize : embedding rows -> unit rows, x / max(||x||, 1e-12)
//                     (generate_predictions_aug.py:377/:380  F.normalize(embedding, p=2, dim=1))
//   * row_argmax    : predicted primitive type per point (generate_predictions_aug.py:365)
//   * type_vote     : most frequent predicted type per segment, ties -> smallest id
//                     (Fitting_patches_and_edges/residual_utils.py:259  stats.mode(pred_primitives[segment]))
#include "common.h"

namespace {

// one wave per row; d <= 64 * 4
__global__ __launch_bounds__(256) void row_normalize_kernel(const float* __restrict__ in, int ldi, int d,
                                                            float* __restrict__ out, int ldo, int dpad,
                                                            size_t rows) {
    const size_t row = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* x = in + row * ldi;
    float v[4];
    float ss = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int c = lane + 64 * u;
        v[u] = c < d ? x[c] : 0.f;
        ss = fmaf(v[u], v[u], ss);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off, 64);
    const float den = fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int c = lane + 64 * u;
        if (c < dpad) out[row * ldo + c] = c < d ? v[u] / den : 0.f;
    }
}

__global__ void row_argmax_kernel(const float* __restrict__ in, int ld, int C, int* __restrict__ out, size_t rows) {
    const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const float* x = in + r * ld;
    float best = x[0];
    int bi = 0;
    for (int c = 1; c < C; ++c)
        if (x[c] > best) { best = x[c]; bi = c; }
    out[r] = bi;
}

__global__ __launch_bounds__(256) void type_vote_kernel(const int* __restrict__ labels, const int* __restrict__ types,
                                                        int N, int S, int C, int* __restrict__ seg_type,
                                                        int* __restrict__ seg_count) {
    extern __shared__ int hist[];            // [S][C]
    const int cloud = blockIdx.x;
    for (int i = threadIdx.x; i < S * C; i += 256) hist[i] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < N; i += 256) {
        const int l = labels[(size_t)cloud * N + i], t = types[(size_t)cloud * N + i];
        if (l >= 0 && l < S && t >= 0 && t < C) atomicAdd(&hist[l * C + t], 1);
    }
    __syncthreads();
    for (int s = threadIdx.x; s < S; s += 256) {
        int best = -1, bt = 0, tot = 0;
        for (int c = 0; c < C; ++c) {
            const int h = hist[s * C + c];
            tot += h;
            if (h > best) { best = h; bt = c; }
        }
        seg_type[(size_t)cloud * S + s] = bt;
        if (seg_count) seg_count[(size_t)cloud * S + s] = tot;
    }
}

}  // namespace

// out[r, :dpad] = in[r, :d] / max(||in[r, :d]||, 1e-12), zero padded to dpad columns (d <= 256)
extern "C" int sed_row_normalize_f32(size_t rows, int d, int dpad, const float* in, int ldi, float* out, int ldo,
                                     hipStream_t stream) {
    if (rows == 0 || d <= 0 || d > 256 || dpad < d || dpad > 256 || !in || !out || ldi < d || ldo < dpad) return SED_EINVAL;
    row_normalize_kernel<<<(unsigned)((rows + 3) / 4), 256, 0, stream>>>(in, ldi, d, out, ldo, dpad, rows);
    SED_LAUNCH_CHECK();
    return SED_OK;
}

// out[r] = argmax_c in[r, c] (first maximum)
extern "C" int sed_row_argmax_f32(size_t rows, int C, const float* in, int ld, int* out, hipStream_t stream) {
    if (rows == 0 || C <= 0 || !in || !out || ld < C) return SED_EINVAL;
    row_argmax_kernel<<<(unsigned)((rows + 255) / 256), 256, 0, stream>>>(in, ld, C, out, rows);
    SED_LAUNCH_CHECK();
    return SED_OK;
}

// seg_type[b,s] = mode of types[b, labels[b,:] == s] over C classes (ties -> smallest), seg_count optional
extern "C" int sed_segment_type_vote(int B, int N, int S, int C, const int* labels, const int* types, int* seg_type,
                                     int* seg_count, hipStream_t stream) {
    if (B <= 0 || N <= 0 || S <= 0 || C <= 0 || S * C > 8192 || !labels || !types || !seg_type) return SED_EINVAL;
    type_vote_kernel<<<B, 256, (size_t)S * C * sizeof(int), stream>>>(labels, types, N, S, C, seg_type, seg_count);
    SED_LAUNCH_CHECK();
    return SED_OK;
}
