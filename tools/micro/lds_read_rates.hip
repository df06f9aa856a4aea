// Microbenchmark: LDS read throughput per CU of the two operand reads of the split-fp16 mean-shift kernels, with the block-sparse
// kernel's own address patterns and occupancy (two 4-wave workgroups per CU, 17 KiB row-major stage images, rows 272 B apart):
//   mode 0: the first product's reads   -- 16 x ds_read_b128 per 32-key block (8 features of one key per lane)
//   mode 1: the second product's reads  -- 32 x ds_read_b64_tr_b16 per block (transpose read: 4 keys of one feature per lane)
//   mode 2: 32 x plain ds_read_b64 at the same addresses (is the transpose itself slow?)
//   mode 3: what a four-plane image would cost for the second product: 16 x ds_read_b128 from a [feature][key] plane (rows 80 B apart)
// Prints cycles per block and wave, LDS bytes per clock and CU.
// hipcc --offload-arch=gfx950 -O3 tools/micro/lds_read_rates.hip -o /tmp/lds_rates && /tmp/lds_rates
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(int* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];     // 4 stage buffers of 17408 B
    constexpr int XROW = 272, OFF_XL = 32 * XROW, STAGE = 2 * OFF_XL, TROW = 80;
    for (int i = threadIdx.x; i < 4 * STAGE / 4; i += 256) ((int*)lds)[i] = i * 2654435761u;
    __syncthreads();
    const int lane = threadIdx.x & 63, li = lane & 31, hi = lane >> 5;
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const unsigned xoff = base + (16 * (li >> 4) + 4 * (li & 3) + ((li >> 2) & 3)) * XROW + hi * 16;
    const unsigned toff = base + (4 * ((lane & 15) >> 2) + hi) * XROW + 32 * ((lane >> 4) & 1) + 8 * (lane & 3);
    const unsigned t4off = base + li * TROW + hi * 16;
    i32x4 acc4 = {0, 0, 0, 0};
    i32x2 acc2 = {0, 0};
    for (int it = 0; it < iters; ++it) {
        const unsigned bo = (it & 3) * (MODE == 3 ? 12288 : STAGE);
        if (MODE == 0) {
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                i32x4 a, b;
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a) : "v"(xoff + bo + t * 32), "n"(0));
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b) : "v"(xoff + bo + t * 32), "n"(OFF_XL));
                asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(a), "+v"(b));
                acc4 ^= a ^ b;
            }
        } else if (MODE == 3) {
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                i32x4 a, b;
                asm volatile("ds_read_b128 %0, %1" : "=v"(a) : "v"(t4off + bo + (t >> 1) * 32 * TROW + (t & 1) * 32));
                asm volatile("ds_read_b128 %0, %1" : "=v"(b) : "v"(t4off + bo + 128 * TROW / 2 + (t >> 1) * 32 * TROW + (t & 1) * 32));
                asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(a), "+v"(b));
                acc4 ^= a ^ b;
            }
        } else {
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int c = t >> 1, j = t & 1;
                i32x2 a, b, c2, d;
                const unsigned ad = toff + bo + (16 * j) * XROW + 64 * c;
                if (MODE == 1) {
                    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(a) : "v"(ad));
                    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:544" : "=v"(b) : "v"(ad));
                    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:8704" : "=v"(c2) : "v"(ad));
                    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:9248" : "=v"(d) : "v"(ad));
                } else {
                    asm volatile("ds_read_b64 %0, %1" : "=v"(a) : "v"(ad));
                    asm volatile("ds_read_b64 %0, %1 offset:544" : "=v"(b) : "v"(ad));
                    asm volatile("ds_read_b64 %0, %1 offset:8704" : "=v"(c2) : "v"(ad));
                    asm volatile("ds_read_b64 %0, %1 offset:9248" : "=v"(d) : "v"(ad));
                }
                asm volatile("s_waitcnt lgkmcnt(12)" : "+v"(a), "+v"(b), "+v"(c2), "+v"(d));
                acc2 ^= a ^ b ^ c2 ^ d;
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    out[blockIdx.x * 256 + threadIdx.x] = acc4[0] ^ acc4[1] ^ acc4[2] ^ acc4[3] ^ acc2[0] ^ acc2[1];
}

template <int MODE>
void run(int* d, const char* what, double bytes_per_iter_wave) {
    const int iters = 100000, sm = 4 * 17408;
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, sm);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<MODE><<<512, 256, sm>>>(d, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<512, 256, sm>>>(d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    int clk = 0;
    hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);      // kHz
    const double cyc = ms * 1e-3 * clk * 1e3 / iters;                  // cycles per iteration at the nominal clock
    printf("%-58s %8.2f ms: %6.0f cycles per block and wave (nominal clock), %6.1f LDS bytes per clock and CU (8 waves)\n", what, ms, cyc,
           8.0 * bytes_per_iter_wave / cyc);
}

int main() {
    int* d;
    hipMalloc(&d, 512 * 256 * 4);
    run<0>(d, "first product: 16 x ds_read_b128 (16 KiB per wave)", 16384.0);
    run<1>(d, "second product: 32 x ds_read_b64_tr_b16 (16 KiB per wave)", 16384.0);
    run<2>(d, "same addresses, plain ds_read_b64", 16384.0);
    run<3>(d, "four-plane layout: 16 x ds_read_b128 from [feature][key]", 16384.0);
    return 0;
}
