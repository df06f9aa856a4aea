// What does ds_read_b64_tr_b16 return? Every lane passes its own 8-byte-aligned LDS address; LDS holds element index as value.
// hipcc --offload-arch=gfx950 -O3 tools/micro/tr_read_probe.hip -o tools/micro/tr_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(int* out, int mode) {
    __shared__ __attribute__((aligned(16))) short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
    __syncthreads();
    const int l = threadIdx.x;
    // mode 0: lane l points at elements 4 l .. 4 l + 3 (contiguous chunks)
    // mode 1: [4 keys][16 cols] block per 16-lane group, row stride 136 elements: lane i -> key i / 4, chunk i % 4
    int e;
    if (mode == 0) e = 4 * l;
    else e = (l >> 4) * 1024 + ((l & 15) >> 2) * 136 + (l & 3) * 4;
    unsigned addr = (unsigned)(size_t)(&lds[e]);
    s16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
    int* d; hipMalloc(&d, 64 * 4 * 4);
    int h[256];
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) { if (l < 20 || l == 32 || l == 48) printf("lane %2d: %5d %5d %5d %5d\n", l, h[4*l], h[4*l+1], h[4*l+2], h[4*l+3]); }
    }
    return 0;
}
