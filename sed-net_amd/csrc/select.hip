// Exact per-row k-smallest selection over materialised distance rows.
//
// One 256-thread workgroup per row; the row lives in registers (VPT values per thread, coalesced
// loads). The k-th smallest key is found by bisection on the order-preserving uint32 image of the
// floats: each step counts `key <= mid` with wave ballots + scalar popcounts (no atomics, no LDS
// traffic besides 4 partial counts), so the result is deterministic. Ties at the k-th value resolve
// to the lowest column index (torch.topk leaves this unspecified; PointNet.py:83, mean_shift.py:133).
//   mode KTH : write the k-th smallest value                (mean_shift.py:133-135 top_k[:, -1])
//   mode IDX : write the k column indices, ascending by (value, index)   (PointNet.py:83 topk indices)
#include "common.h"

namespace {

constexpr int KMAX = 512;

__device__ __forceinline__ int block_sum_int(int v_wave_total, int* slots, int lane, int wave) {
    // v_wave_total is wave-uniform; 4 waves
    if (lane == 0) slots[wave] = v_wave_total;
    __syncthreads();
    int s = slots[0] + slots[1] + slots[2] + slots[3];
    return s;
}

template <int VPT, bool WANT_IDX>
__global__ __launch_bounds__(256) void row_select_kernel(const float* __restrict__ Dm, int N, int ldD, int k,
                                                         float* __restrict__ kth_out, int* __restrict__ idx_out) {
    __shared__ int slots[2][4];
    __shared__ uint32_t red[8];
    __shared__ uint32_t sel_key[WANT_IDX ? KMAX : 1];
    __shared__ int sel_idx[WANT_IDX ? KMAX : 1];
    __shared__ int sel_cnt;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t row = (size_t)blockIdx.y * N + blockIdx.x;
    const float* src = Dm + row * ldD;

    uint32_t key[VPT];
    uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
#pragma unroll
    for (int v = 0; v < VPT; ++v) {
        const int j = v * 256 + tid;
        uint32_t u = 0xFFFFFFFFu;
        if (j < N) {
            u = f32_sortable(src[j]);
            kmin = min(kmin, u);
            kmax = max(kmax, u);
        }
        key[v] = u;
    }
    // block min / max to narrow the bisection range
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, off, 64));
        kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, off, 64));
    }
    if (lane == 0) { red[wave] = kmin; red[4 + wave] = kmax; }
    if (tid == 0) sel_cnt = 0;
    __syncthreads();
    uint32_t lo = min(min(red[0], red[1]), min(red[2], red[3]));
    uint32_t hi = max(max(red[4], red[5]), max(red[6], red[7]));

    int it = 0;
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        int c = 0;
#pragma unroll
        for (int v = 0; v < VPT; ++v) c += __popcll(__ballot(key[v] <= mid));
        const int total = block_sum_int(c, slots[it & 1], lane, wave);
        if (total >= k) hi = mid; else lo = mid + 1;
        ++it;
    }
    const uint32_t kth = lo;

    if (!WANT_IDX) {
        if (tid == 0) kth_out[row] = sortable_f32(kth);
        return;
    } else {
        int cl = 0, ce = 0;
#pragma unroll
        for (int v = 0; v < VPT; ++v) {
            cl += __popcll(__ballot(key[v] < kth));
            ce += __popcll(__ballot(key[v] == kth));
        }
        const int n_less = block_sum_int(cl, slots[it & 1], lane, wave); ++it;
        const int n_eq = block_sum_int(ce, slots[it & 1], lane, wave); ++it;
        const int need = k - n_less;              // ties to take, lowest index first
        const bool take_all_eq = (n_eq == need);
#pragma unroll
        for (int v = 0; v < VPT; ++v) {
            const bool take = key[v] < kth || (take_all_eq && key[v] == kth);
            if (take) {
                const int pos = atomicAdd(&sel_cnt, 1);
                sel_key[pos] = key[v];
                sel_idx[pos] = v * 256 + tid;
            }
        }
        __syncthreads();
        if (!take_all_eq) {
            // rare: more equal values than slots -> pick the `need` lowest indices one by one
            int last = -1;
            for (int n = 0; n < need; ++n) {
                int best = 0x7fffffff;
#pragma unroll
                for (int v = 0; v < VPT; ++v) {
                    const int j = v * 256 + tid;
                    if (key[v] == kth && j > last && j < best) best = j;
                }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) best = min(best, __shfl_xor(best, off, 64));
                if (lane == 0) slots[it & 1][wave] = best;
                __syncthreads();
                last = min(min(slots[it & 1][0], slots[it & 1][1]), min(slots[it & 1][2], slots[it & 1][3]));
                ++it;
                if (tid == 0) { sel_key[n_less + n] = kth; sel_idx[n_less + n] = last; }
            }
            __syncthreads();
        }
        // rank sort the k winners by (key, index)
        int* out = idx_out + row * k;
        for (int e = tid; e < k; e += 256) {
            const uint32_t ke = sel_key[e];
            const int ie = sel_idx[e];
            int rank = 0;
            for (int j = 0; j < k; ++j) {
                const uint32_t kj = sel_key[j];
                rank += (kj < ke) || (kj == ke && sel_idx[j] < ie);
            }
            out[rank] = ie;
        }
    }
}

template <bool WANT_IDX>
int launch_select(int B, int N, int ldD, int k, const float* D, float* kth, int* idx, hipStream_t stream) {
    dim3 grid(N, B), block(256);
    if (N <= 8 * 256) row_select_kernel<8, WANT_IDX><<<grid, block, 0, stream>>>(D, N, ldD, k, kth, idx);
    else if (N <= 16 * 256) row_select_kernel<16, WANT_IDX><<<grid, block, 0, stream>>>(D, N, ldD, k, kth, idx);
    else if (N <= 40 * 256) row_select_kernel<40, WANT_IDX><<<grid, block, 0, stream>>>(D, N, ldD, k, kth, idx);
    else if (N <= 64 * 256) row_select_kernel<64, WANT_IDX><<<grid, block, 0, stream>>>(D, N, ldD, k, kth, idx);
    else return SED_EUNSUPPORTED;
    SED_LAUNCH_CHECK();
    return SED_OK;
}

// bw[b] = max(mean_i sqrt(max(kth[b][i], 1e-6)), min_bw)   (mean_shift.py:135-137 and :34)
__global__ __launch_bounds__(256) void bandwidth_finalize_kernel(const float* __restrict__ kth, int N, float min_bw,
                                                                 float* __restrict__ bw) {
    __shared__ double part[256];
    const float* k = kth + (size_t)blockIdx.x * N;
    double acc = 0.0;
    for (int i = threadIdx.x; i < N; i += 256) acc += (double)sqrtf(fmaxf(k[i], 1e-6f));
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) bw[blockIdx.x] = fmaxf((float)(part[0] / (double)N), min_bw);
}

}  // namespace

// k-th smallest value per row of D [B,N,ldD] -> kth [B,N]
extern "C" int sed_row_kth_f32(int B, int N, int ldD, int k, const float* D, float* kth, hipStream_t stream) {
    if (B <= 0 || N <= 0 || k <= 0 || k > N || !D || !kth || ldD < N) return SED_EINVAL;
    return launch_select<false>(B, N, ldD, k, D, kth, nullptr, stream);
}

// indices of the k smallest values per row, ascending by (value, index) -> idx [B,N,k] int32
extern "C" int sed_row_topk_idx_f32(int B, int N, int ldD, int k, const float* D, int* idx, hipStream_t stream) {
    if (B <= 0 || N <= 0 || k <= 0 || k > N || k > KMAX || !D || !idx || ldD < N) return SED_EINVAL;
    return launch_select<true>(B, N, ldD, k, D, nullptr, idx, stream);
}

extern "C" int sed_ms_bandwidth_finalize_f32(int B, int N, float min_bw, const float* kth, float* bw,
                                             hipStream_t stream) {
    if (B <= 0 || N <= 0 || !kth || !bw) return SED_EINVAL;
    bandwidth_finalize_kernel<<<B, 256, 0, stream>>>(kth, N, min_bw, bw);
    SED_LAUNCH_CHECK();
    return SED_OK;
}
