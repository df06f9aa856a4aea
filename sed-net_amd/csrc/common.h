// Shared helpers for the SED-Net gfx950 kernels (wave64, CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SED_OK 0
#define SED_EINVAL (-1)      // bad argument (null pointer, unsupported size, workspace too small)
#define SED_EUNSUPPORTED (-2)

#define SED_LAUNCH_CHECK()                                   \
    do {                                                     \
        hipError_t e__ = hipGetLastError();                  \
        if (e__ != hipSuccess) return (int)e__;              \
    } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// C/D row of accumulator register r for v_mfma_f32_32x32x2_f32 (lane>>5 = hi):
//   row = (r & 3) + 8 * (r >> 2) + 4 * hi ; col = lane & 31
__device__ __forceinline__ int mfma_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float xor32(float v) { return __shfl_xor(v, 32, 64); }

// float -> uint32 whose unsigned order equals the float order (-0 < +0, NaNs at the ends)
__device__ __forceinline__ uint32_t f32_sortable(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float sortable_f32(uint32_t u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

// XCD-aware (x, y) block mapping for grids of shape (blocks per cloud, clouds): the dispatcher deals consecutive
// workgroup ids round-robin to the 8 XCDs, each with a private L2. Remapping gives every XCD a contiguous range of ids =
// whole clouds, so the workgroups resident on an XCD stream the SAME cloud through its L2 instead of 8 different ones.
// Speed only: any placement is correct. Returns the cloud, writes the block index within the cloud.
__device__ __forceinline__ int sed_xcd_cloud_block(int* bx) {
    const int nbx = gridDim.x, nwg = gridDim.x * gridDim.y;
    const int orig = blockIdx.y * nbx + blockIdx.x;
    const int xcd = orig & 7, qd = nwg >> 3, rm = nwg & 7;
    const int wgid = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (orig >> 3);
    const int cloud = wgid / nbx;
    *bx = wgid - cloud * nbx;
    return cloud;
}

// Per-row squared norm, sequential over channels with the products rounded before the adds
// (PointNet.py:77  xx = torch.sum(x ** 2, dim=1)). One thread = one row (256 rows per block); the rows are staged
// through LDS in 32-channel slices so that global reads are coalesced (a thread walking its own 256-byte row re-fetches
// every line ~20 times through L1). Same additions in the same order as the plain per-thread loop.
__device__ __forceinline__ void sed_row_sqnorm_block(const float* __restrict__ X, float* __restrict__ xx, int rows,
                                                     int D, int C, float* tile /* [256 * 33] */) {
    const int r0 = blockIdx.x * 256, tid = threadIdx.x;
    float acc = 0.f;
    for (int c0 = 0; c0 < C; c0 += 32) {
        for (int i = tid; i < 256 * 32; i += 256) {
            const int row = i >> 5, c = i & 31;
            tile[row * 33 + c] = (r0 + row < rows && c0 + c < C) ? X[(size_t)(r0 + row) * D + c0 + c] : 0.f;
        }
        __syncthreads();
        const int cn = C - c0 < 32 ? C - c0 : 32;
        for (int c = 0; c < cn; ++c) {
            const float v = tile[tid * 33 + c];
            acc = __fadd_rn(acc, __fmul_rn(v, v));
        }
        __syncthreads();
    }
    if (r0 + tid < rows) xx[r0 + tid] = acc;
}

// v_min_f32 / v_max_f32 as single instructions. fminf / fmaxf are lowered with a canonicalising v_max_f32 x, x, x in front
// of (almost) every operand the compiler cannot prove canonical -- in the bucket-minima networks of the selection sweeps
// that was 112 of 368 instructions per 16 values. NaN operands: the other operand is returned, like fminf / fmaxf.
__device__ __forceinline__ float sed_vmin(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float sed_vmax(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// The same for an operand that comes STRAIGHT out of an MFMA accumulator: the hazard recogniser does not count an inline-asm
// read as a VALU read of the matrix pipe's result, so sed_vmax on a fresh accumulator register can issue before the MFMA has
// written it (seen in edgeconv_x3_kernel: the first rows of a tile kept stale values). v_med3_f32 with +inf as third operand is
// max(a, b) as one instruction the compiler knows.
__device__ __forceinline__ float sed_vmax_acc(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, __builtin_inff()); }


// A kernel's dynamic-LDS limit must be raised once PER DEVICE before its first launch there (hipFuncSetAttribute). The launchers
// remember which devices have seen the call in a function-local bit set (written atomically; the call is idempotent, so a race
// between host threads only repeats it). Not behaviour: nothing a caller can observe depends on it.
#include <atomic>
static inline bool sed_first_on_device(std::atomic<unsigned long long>& seen, int* err) {
    int dev = 0;
    const hipError_t e = hipGetDevice(&dev);
    *err = e == hipSuccess ? 0 : (int)e;
    if (e != hipSuccess) return false;
    return dev >= 64 || !((seen.load(std::memory_order_relaxed) >> dev) & 1ull);
}
static inline void sed_mark_device(std::atomic<unsigned long long>& seen) {
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && dev < 64) seen.fetch_or(1ull << dev, std::memory_order_relaxed);
}

static inline int sed_pad_dim(int d) {      // feature width the MFMA kernels are instantiated for
    if (d <= 32) return 32;
    if (d <= 64) return 64;
    if (d <= 96) return 96;
    if (d <= 128) return 128;
    if (d <= 160) return 160;
    return -1;
}
