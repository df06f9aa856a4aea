// Microbenchmark: do v_mfma_f32_32x32x16_f16 and VALU work (v_exp_f32, conversions) overlap on one SIMD of gfx950
//   (a) across the two waves that share a SIMD (wave w: MFMAs only, wave w + 4: VALU only),
//   (b) inside one wave (independent VALU instructions interleaved between MFMAs)?
// hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_valu_overlap.hip -o /tmp/overlap && /tmp/overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// 0: all waves MFMA only; 1: all waves VALU only; 2: waves 0-3 MFMA, 4-7 VALU; 3: every wave interleaves both (4 independent
// accumulators); 4: as 3 with ONE accumulator (every MFMA depends on the previous one, like a k-loop); 5: as 4, MFMA only
template <int MODE>
__global__ __launch_bounds__(512, 1) void k(float* out, int iters) {
    const int wave = threadIdx.x >> 6;
    h16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float v[16];
    for (int r = 0; r < 16; ++r) v[r] = threadIdx.x * 1e-3f + r;
    const bool do_mfma = MODE == 0 || MODE == 3 || (MODE == 2 && wave < 4);
    const bool do_valu = MODE == 1 || MODE == 3 || (MODE == 2 && wave >= 4);
    for (int it = 0; it < iters; ++it) {
        if (MODE == 5) {
#pragma unroll
            for (int t = 0; t < 24; ++t) acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[0], 0, 0, 0);
        } else if (MODE == 3 || MODE == 4) {
#pragma unroll
            for (int t = 0; t < 24; ++t) {
                const int ai = MODE == 3 ? (t & 3) : 0;
                acc[ai] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[ai], 0, 0, 0);
                // ~8 VALU per MFMA slot (32 cycles): two thirds of an exp + conversion chain per MFMA
                if (t < 16) {
                    float p = __builtin_amdgcn_exp2f(fmaxf(fmaf(v[t], 1.0001f, 0.5f), -100.f));
                    _Float16 h = (_Float16)p;
                    v[t] = p - (float)h + v[t] * 0.5f;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            if (do_mfma) {
#pragma unroll
                for (int t = 0; t < 24; ++t) acc[t & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[t & 3], 0, 0, 0);
            }
            if (do_valu) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float p = __builtin_amdgcn_exp2f(fmaxf(fmaf(v[r], 1.0001f, 0.5f), -100.f));
                    _Float16 h = (_Float16)p;
                    v[r] = p - (float)h + v[r] * 0.5f;
                }
            }
        }
    }
    float s = 0.f;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    for (int r = 0; r < 16; ++r) s += v[r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE> float run(float* d, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<256, 512>>>(d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<256, 512>>>(d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    float* d; hipMalloc(&d, 256 * 512 * 4);
    const int iters = 200000;
    const float m0 = run<0>(d, iters), m1 = run<1>(d, iters), m2 = run<2>(d, iters), m3 = run<3>(d, iters);
    const float m4 = run<4>(d, iters), m5 = run<5>(d, iters);
    printf("per iteration and SIMD (2 waves): 24 MFMAs each -> ideal 2 x 24 x 32 = 1536 cycles\n");
    printf("mode 0 (both waves MFMA only)            %.2f ms\n", m0);
    printf("mode 1 (both waves VALU only, 16 exp chains each) %.2f ms\n", m1);
    printf("mode 2 (one wave MFMA, the other VALU)   %.2f ms   [perfect overlap: max(%.2f, %.2f); none: %.2f]\n", m2, m0 / 2, m1 / 2, m0 / 2 + m1 / 2);
    printf("mode 3 (every wave interleaves both)     %.2f ms   [perfect overlap: max(%.2f, %.2f); none: %.2f]\n", m3, m0, m1, m0 + m1);
    printf("mode 5 (MFMA only, ONE accumulator chain) %.2f ms\n", m5);
    printf("mode 4 (interleaved, ONE accumulator chain) %.2f ms   [perfect overlap: max(%.2f, %.2f); none: %.2f]\n", m4, m5, m1, m5 + m1);
    return 0;
}
