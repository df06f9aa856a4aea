// Microbenchmark: issue rate of v_mfma_f32_32x32x16_f16 for ONE wave per SIMD (4-wave workgroup, one workgroup per CU) when
//   mode 0: every MFMA accumulates into the SAME registers (a k-loop: each MFMA depends on the previous one),
//   mode 1: two alternating accumulators, mode 2: four, mode 3: six consecutive MFMAs per accumulator, four accumulators in turn
//           (the second product of the mean-shift kernels: 3 terms x 2 key halves per feature tile),
//   mode 4: as mode 3 but the accumulators interleaved (tile 0, 1, 2, 3, 0, 1, ...), same order per accumulator.
// Also with TWO waves per SIMD (8-wave workgroups) for comparison.
// hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_chain.hip -o /tmp/chain && /tmp/chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE, int THREADS>
__global__ __launch_bounds__(THREADS, 1) void k(float* out, int iters) {
    h16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 24; ++t) {
            const int ai = MODE == 0 ? 0 : MODE == 1 ? (t & 1) : MODE == 2 ? (t & 3) : MODE == 3 ? (t / 6) : (t & 3);
            acc[ai] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[ai], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * THREADS + threadIdx.x] = s;
}

template <int MODE, int THREADS>
float run(float* d) {
    const int iters = 100000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE, THREADS><<<256, THREADS>>>(d, 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<MODE, THREADS><<<256, THREADS>>>(d, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    int clk = 0; (void)hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    printf("mode %d, %d wave(s) per SIMD: %8.2f ms = %5.1f cycles per MFMA and wave at the nominal clock\n", MODE, THREADS / 256, ms,
           ms * 1e-3 * clk * 1e3 / (iters * 24.0));
    return ms;
}
int main() {
    float* d; (void)hipMalloc(&d, 256 * 512 * 4);
    run<0, 256>(d); run<1, 256>(d); run<2, 256>(d); run<3, 256>(d); run<4, 256>(d);
    run<0, 512>(d); run<2, 512>(d); run<3, 512>(d);
    return 0;
}
