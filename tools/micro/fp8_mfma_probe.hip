// Microbenchmark for the "4.5-MFMA" idea: v_mfma_f32_32x32x64_f8f6f4 with fp8 (e4m3) operands on gfx950
//   (1) operand layout: which (row, k) does byte j of lane l of A / B hold?   (one-hot probes)
//   (2) throughput beside v_mfma_f32_32x32x16_f16: a loop of 8 fp16 MFMAs vs 4 fp16 + 1 fp8 (same K covered) per step
// hipcc --offload-arch=gfx950 -O3 tools/micro/fp8_mfma_probe.hip -o /tmp/fp8probe && /tmp/fp8probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x16 mfma8(v8i a, v8i b, f32x16 c) {
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
}

// A [32][64], B [64][32] as bytes (fp8 e4m3: 0x38 = 1.0, 0x40 = 2.0); lane l loads a[(l % 32) * 64 + (l / 32) * 32 + j] -- the
// HYPOTHESIS to test: lane l holds row l % 32, k = 32 (l / 32) + j, j = byte index 0 .. 31; same for B with its column.
__global__ void layout(const uint8_t* A, const uint8_t* B, float* C) {
    const int l = threadIdx.x;
    v8i a, b;
    for (int i = 0; i < 8; ++i) {
        uint32_t wa = 0, wb = 0;
        for (int j = 0; j < 4; ++j) {
            const int k = 32 * (l / 32) + 4 * i + j;
            wa |= (uint32_t)A[(l % 32) * 64 + k] << (8 * j);
            wb |= (uint32_t)B[k * 32 + (l % 32)] << (8 * j);
        }
        a[i] = (int)wa; b[i] = (int)wb;
    }
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    c = mfma8(a, b, c);
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * (l / 32)) * 32 + (l % 32)] = c[r];
}

template <int MODE>      // 0: 8 fp16 MFMAs per step; 1: 4 fp16 + 1 fp8; 2: 2 fp8
__global__ __launch_bounds__(512, 1) void rate(float* out, int iters) {
    h16x8 a, b;
    v8i a8, b8;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); a8[i] = 0x38383838 + threadIdx.x; b8[i] = 0x30303030 + i; }
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
            if (MODE == 0) {
#pragma unroll
                for (int t = 0; t < 8; ++t) acc[t & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[t & 3], 0, 0, 0);
            } else if (MODE == 1) {
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[t & 3], 0, 0, 0);
                acc[rep & 3] = mfma8(a8, b8, acc[rep & 3]);
            } else {
                acc[0] = mfma8(a8, b8, acc[0]);
                acc[1] = mfma8(a8, b8, acc[1]);
            }
        }
    }
    float s = 0.f;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

int main() {
    // ---- layout
    std::vector<uint8_t> A(32 * 64, 0), B(64 * 32, 0);
    std::vector<float> C(32 * 32), R(32 * 32, 0.f);
    uint32_t s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s >> 24; };
    const uint8_t vals[4] = {0x00, 0x38, 0x40, 0xB8};                 // 0, 1, 2, -1 in e4m3
    const float fv[4] = {0.f, 1.f, 2.f, -1.f};
    std::vector<int> ia(32 * 64), ib(64 * 32);
    for (auto& x : ia) x = rnd() & 3;
    for (auto& x : ib) x = rnd() & 3;
    for (int i = 0; i < 32 * 64; ++i) { A[i] = vals[ia[i]]; B[i] = vals[ib[i]]; }
    for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) for (int k = 0; k < 64; ++k) R[m * 32 + n] += fv[ia[m * 64 + k]] * fv[ib[k * 32 + n]];
    uint8_t *dA, *dB; float* dC;
    hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dC, C.size() * 4);
    hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
    layout<<<1, 64>>>(dA, dB, dC);
    hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
    double err = 0; for (int i = 0; i < 32 * 32; ++i) err = fmax(err, fabs(C[i] - R[i]));
    printf("layout hypothesis (lane l: row/col l %% 32, k = 32 (l / 32) + byte j; C like the fp16 32x32 MFMA): max |C - ref| = %g  (C[0] %g ref %g)\n", err, C[0], R[0]);
    // ---- rate
    float* out; hipMalloc(&out, 1024 * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    auto run = [&](auto kern, const char* name, double mfma_cycles_per_iter) {
        kern<<<1024, 512>>>(out, 100); hipDeviceSynchronize();
        hipEventRecord(e0); kern<<<1024, 512>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // 1024 workgroups of 8 waves on 256 CUs: 4 rounds, 2 waves per SIMD
        printf("%-28s %8.2f ms  -> %.1f ns per step and SIMD (2 waves); ideal at 2.4 GHz: %.1f ns\n", name, ms, ms * 1e6 / iters / 4 / 4,
               2 * mfma_cycles_per_iter / 2.4);
    };
    run(rate<0>, "8 x fp16 32x32x16", 8 * 32.0);
    run(rate<1>, "4 x fp16 + 1 x fp8 32x32x64", 4 * 32.0 + 64.0);
    run(rate<2>, "2 x fp8 32x32x64", 128.0);
    return 0;
}
