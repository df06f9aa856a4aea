// Mean-shift iterations on the unit hypersphere, all `iters` iterations in ONE launch.
//
// Replaces the loop body of /root/reference/src/mean_shift.py:56-77
//     dist = 2 - 2 new_X X^T ; K = exp(clamp(-dist / b^2 / 2, -75, 75)) ; D = 1 / sum_j K
//     new_X = new_X + ((K X) * D - new_X) ; new_X /= ||new_X||
// without ever materialising the N x N matrices: flash-attention shaped, keys == values == X.
//
// Structure (gfx950, wave64, fp32-input MFMA v_mfma_f32_32x32x2_f32 = exact fp32 fma chains):
//   * rows never interact across iterations (new_X[i] at t+1 depends on new_X[i] at t and the
//     FIXED X), so one workgroup owns 128 query rows (4 waves x 32) for all iterations; no grid sync.
//   * per 32-key tile a wave computes  S^T = X_tile . Q^T  (keys on accumulator rows, queries on
//     lanes), P = exp(...) in registers, then  O^T += X_tile^T . P^T.  With that operand order the
//     accumulator layout of O^T *is* the B-operand layout of Q for the next iteration's S^T, and the
//     P registers *are* the B operand of the second product: no transposes, no LDS round trips.
//       lane = (query = lane & 31, hi = lane >> 5); register (t, r) of Q / O holds feature
//       d = 32 t + (r & 3) + 8 (r >> 2) + 4 hi.
//   * X tiles (32 keys x D floats, row stride D + 4 floats -> conflict-free ds_read_b128) are staged
//     through a double-buffered LDS ring shared by the 4 waves, one barrier per tile; X (N*D*4 B =
//     5 MB at N = 10k, D = 128) is re-streamed from L2 / Infinity Cache every iteration.
// Algorithmic work: 4 N^2 D flops per iteration per cloud; bound: fp32 MFMA.
#include "common.h"

namespace {

// exp(a) for a in [-75, 75] on the hardware exp2: t = RN(a * log2e), e = the rounding error of that product
// (recovered exactly with one fma) plus a * lo(log2e); exp(a) = 2^t * 2^e ~= 2^t * (1 + e ln2).
// ~1 ulp of v_exp_f32 instead of the |a| * 2^-24 relative error of a bare exp2f(a * log2e); 6 instructions.
__device__ __forceinline__ float exp_compensated(float a) {
    const float L2E_HI = 1.44269502162933349609375f;          // float(log2(e))
    const float L2E_LO = 1.925963033500011e-08f;              // log2(e) - L2E_HI
    const float LN2 = 0.693147182464599609375f;
    const float t = a * L2E_HI;
    const float e = fmaf(a, L2E_LO, fmaf(a, L2E_HI, -t));
    const float r = __builtin_amdgcn_exp2f(t);
    return fmaf(r, e * LN2, r);
}

template <int NT, int ABL = 0>
__global__ __launch_bounds__(256, 2) void ms_iterate_kernel(const float* __restrict__ X,
                                                            float* __restrict__ newX,
                                                            const float* __restrict__ bw, int N, int iters) {
    constexpr int D = 32 * NT;
    constexpr int LDX = D + 4;
    constexpr int C4 = D / 4;   // float4 per row
    __shared__ __attribute__((aligned(16))) float lds[2][32 * LDX];

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int li = lane & 31, hi = lane >> 5;
    const int cloud = blockIdx.y;
    const float* Xc = X + (size_t)cloud * N * D;
    const int qrow = blockIdx.x * 128 + wave * 32 + li;
    const int qrow_c = qrow < N ? qrow : N - 1;

    const float b = bw[cloud];
    const float b2 = b * b;
    // -dist / b^2 / 2 as one multiply by a per-cloud constant: differs from the two IEEE divisions by <= 1 ulp of
    // the exponent argument, far below what the rounding of the dot products already contributes
    const float neg_half_inv_b2 = -0.5f / b2;
    const int ntiles = (N + 31) >> 5;

    // Q fragment: q[t][4g + c] = X[qrow][32 t + 8 g + 4 hi + c]
    float q[NT][16];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 v = *(const f32x4*)(Xc + (size_t)qrow_c * D + 32 * t + 8 * g + 4 * hi);
#pragma unroll
            for (int c = 0; c < 4; ++c) q[t][4 * g + c] = v[c];
        }

    // staging assignment: NT float4 per thread per tile
    f32x4 stage[NT];
    auto stage_load = [&](int tile) {
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            const int i = tid + 256 * u;
            const int row = i / C4, c4 = i % C4;
            const int key = tile * 32 + row;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (key < N) v = *(const f32x4*)(Xc + (size_t)key * D + 4 * c4);
            stage[u] = v;
        }
    };
    auto stage_store = [&](int buf) {
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            const int i = tid + 256 * u;
            const int row = i / C4, c4 = i % C4;
            *(f32x4*)(&lds[buf][row * LDX + 4 * c4]) = stage[u];
        }
    };

    stage_load(0);
    stage_store(0);
    __syncthreads();
    int cur = 0;

    for (int it = 0; it < iters; ++it) {
        f32x16 o[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
        float rsum = 0.f;

        for (int tile = 0; tile < ntiles; ++tile) {
            const bool last = (it == iters - 1) && (tile == ntiles - 1);
            if (!last && !(ABL & 1)) stage_load(tile + 1 == ntiles ? 0 : tile + 1);

            const float* xt = lds[cur];
            // ---- S^T = X_tile . Q^T  (keys on rows, queries on lanes)
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 xa = *(const f32x4*)(xt + li * LDX + 32 * t + 8 * g + 4 * hi);
#pragma unroll
                    for (int c = 0; c < 4; ++c) s = mfma32(xa[c], q[t][4 * g + c], s);
                }
            // ---- P = exp(clamp(-(2 - 2 s) / b^2 / 2))   (mean_shift.py:60-63, guard.py:7-9)
            float p[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float dist = 2.0f - 2.0f * s[r];
                float a = dist * neg_half_inv_b2;
                a = fminf(fmaxf(a, -75.0f), 75.0f);
                p[r] = (ABL & 2) ? a : exp_compensated(a);
            }
            if (tile == ntiles - 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (tile * 32 + mfma_row(r, hi) >= N) p[r] = 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) rsum += p[r];
            // ---- O^T += X_tile^T . P^T
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float* xr = xt + mfma_row(r, hi) * LDX + li;
#pragma unroll
                for (int t = 0; t < NT; ++t) o[t] = mfma32(xr[32 * t], p[r], o[t]);
            }

            if (!last && !(ABL & 1)) stage_store(cur ^ 1);
            if (!(ABL & 4)) __syncthreads();
            cur ^= 1;
        }

        // ---- row update (mean_shift.py:70-77)
        const float rs = rsum + xor32(rsum);
        const float Dinv = 1.0f / rs;
        float n2 = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float m = o[t][r] * Dinv - q[t][r];
                const float nq = q[t][r] + m;
                q[t][r] = nq;
                n2 += nq * nq;
            }
        n2 += xor32(n2);
        const float nrm = sqrtf(n2);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) q[t][r] = q[t][r] / nrm;
    }

    if (qrow < N) {
        float* out = newX + ((size_t)cloud * N + qrow) * D;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v = {q[t][4 * g], q[t][4 * g + 1], q[t][4 * g + 2], q[t][4 * g + 3]};
                *(f32x4*)(out + 32 * t + 8 * g + 4 * hi) = v;
            }
    }
}


// ------------------------------------------------------------------------------------------------------------
// D = 128 production kernel: same mathematics and MFMA chain order per product as above, restructured as a
// 2-deep software pipeline so that the matrix pipe never waits for the exponentials:
//     step j :  S(j+1) = X_{j+1} . Q^T     (64 MFMAs, one dependent chain)
//               P(j)   = exp(...S(j)...)   (VALU, ~11 instructions per element, placed between the MFMAs)
//               O     += X_{j-1}^T . P(j-1) (64 MFMAs, 4 chains)
// The three are mutually independent inside a step; S and O MFMAs alternate (different accumulators), so the
// VALU work sits in MFMA shadows. Feature <-> register map for this kernel: register (c, r) of Q / O holds
// d = 4 * ((r&3) + 8(r>>2) + 4 hi) + c, which makes BOTH LDS operand reads ds_read_b128 (16 + 16 per tile
// instead of 16 + 64).
// X tiles arrive by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction, no staging VGPRs, no
// ds_write): the LDS image is lane-linear (rows of 512 B, unpadded), so bank conflicts are avoided with an XOR
// swizzle of the 16-byte slot index by (row & 15), applied on the per-lane SOURCE address and on both reads
// (cdna_hip_programming.md rule 21). Ring of 4 tiles (j-1, j, j+1 resident, j+2 in flight) = 64 KiB per
// workgroup, one barrier per step; the barrier's vmcnt(0) is where the DMA issued a whole step earlier lands.
template <bool DO_S, bool DO_E_, bool DO_V, int ABL>
__device__ __forceinline__ void p128_step(const float* __restrict__ xs, const float* __restrict__ xv, int li, int hi,
                                          const float (&q)[4][16], f32x16 (&o)[4], f32x16& s_next, f32x16& sp,
                                          float (&p_prev)[16], float& rsum, float coef, unsigned keymask) {
    // sp holds S(j) on entry and P(j) on exit (element r is overwritten once its chain completes);
    // p_prev holds P(j-1) and is refreshed from sp by the caller.
    constexpr bool DO_E = DO_E_ && !(ABL & 2);
    const float L2E_HI = 1.44269502162933349609375f, L2E_LO = 1.925963033500011e-08f;
    const float LN2 = 0.693147182464599609375f;
    // Swizzled LDS addresses as  slot_base + (lane_constant ^ C(r))  with C(r) a compile-time constant: row(r,hi)
    // = R(r) ^ (hi << 2) has no carries, so the XOR swizzle folds into one v_xor per read. (Written on the byte
    // address, and tied to the per-step slot base so that the compiler does not hoist 32 address registers.)
    typedef const __attribute__((address_space(3))) f32x4* lds_v4;
    const unsigned sbase = (unsigned)(size_t)(const __attribute__((address_space(3))) float*)xs;
    const unsigned vbase = (unsigned)(size_t)(const __attribute__((address_space(3))) float*)xv;
    const unsigned lane_s = (unsigned)li * 512u + 16u * ((unsigned)(hi << 2) ^ (unsigned)(li & 15));
    const unsigned lane_v = (16u * (unsigned)li) ^ (64u * (unsigned)hi) ^ (2048u * (unsigned)hi);
    f32x4 xa, xb, xa_n, xb_n;
    // the ring is 16 KiB-slot aligned (LDS base aligned to 1 KiB), so the XOR commutes with adding the slot base
    const unsigned sl = sbase + lane_s, vl = vbase + lane_v;
    auto lds_a = [&](int r) {
        const unsigned R = (r & 3) + 8 * (r >> 2);
        return *(lds_v4)(size_t)(sl ^ (16u * R));
    };
    auto lds_b = [&](int r) {
        const unsigned R = (r & 3) + 8 * (r >> 2);
        return *(lds_v4)(size_t)(vl ^ ((16u * (R & 15u)) ^ (512u * R)));
    };
    if (DO_S) xa_n = lds_a(0);
    if (DO_V) xb_n = lds_b(0);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        xa = xa_n; xb = xb_n;
        if (r + 1 < 16) {
            if (DO_S) xa_n = lds_a(r + 1);
            if (DO_V) xb_n = lds_b(r + 1);
        }
        float a = 0.f, t = 0.f, e = 0.f, ex = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (DO_S) s_next = mfma32(xa[c], q[c][r], s_next);
            if (DO_V) o[c] = mfma32(xb[c], p_prev[r], o[c]);
            if (DO_E) {     // element r of P(j), one quarter of its chain per MFMA pair
                if (c == 0) {
                    const float dist = 2.0f - 2.0f * sp[r];
                    a = fminf(fmaxf(dist * coef, -75.0f), 75.0f);
                } else if (c == 1) {
                    t = a * L2E_HI;
                    e = fmaf(a, L2E_LO, fmaf(a, L2E_HI, -t));
                } else if (c == 2) {
                    ex = __builtin_amdgcn_exp2f(t);
                    e = e * LN2;
                } else {
                    float pv = fmaf(ex, e, ex);
                    pv = (keymask >> r) & 1u ? pv : 0.f;
                    sp[r] = pv;
                    rsum += pv;
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int ABL>
__global__ __launch_bounds__(256, 1) void ms_iterate_p128_kernel(const float* __restrict__ X,
                                                                 float* __restrict__ newX,
                                                                 const float* __restrict__ bw, int N, int iters) {
    constexpr int D = 128, NBUF = 4, TILE = 32 * 128;
    extern __shared__ __attribute__((aligned(1024))) float lds[];    // [NBUF][32][128], no static LDS before it

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int li = lane & 31, hi = lane >> 5;
    const int cloud = blockIdx.y;
    const float* Xc = X + (size_t)cloud * N * D;
    const int qrow = blockIdx.x * 128 + wave * 32 + li;
    const int qrow_c = qrow < N ? qrow : N - 1;
    const float b = bw[cloud];
    const float coef = -0.5f / (b * b);
    const int ntiles = (N + 31) >> 5;

    float q[4][16];                      // q[c][r] = X[qrow][4 * row(r,hi) + c]
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const f32x4 v = *(const f32x4*)(Xc + (size_t)qrow_c * D + 4 * mfma_row(r, hi));
#pragma unroll
        for (int c = 0; c < 4; ++c) q[c][r] = v[c];
    }
    unsigned lastmask = 0;               // valid keys of the ragged last tile
#pragma unroll
    for (int r = 0; r < 16; ++r) lastmask |= ((ntiles - 1) * 32 + mfma_row(r, hi) < N ? 1u : 0u) << r;

    // LDS-DMA of one tile: wave w moves rows 8w .. 8w+7 (4 instructions of 2 rows)
    auto stage_dma = [&](int tile, int slot) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int row0 = 2 * (4 * wave + u);
            const int row = row0 + hi;                    // this lane's row within the tile
            int key = tile * 32 + row;
            key = key < N ? key : N - 1;                  // ragged tail: finite duplicate rows, masked in P
            const float* src = Xc + (size_t)key * D + 4 * (li ^ (row & 15));
            float* dst = lds + slot * TILE + row0 * 128;  // wave-uniform; lane l lands at + 16 l bytes
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };

    int n_staged = 0, n_s = 0, n_v = 0;      // running tile sequence numbers (slot = n % NBUF)
    stage_dma(0, n_staged++ % NBUF);
    __syncthreads();

    for (int it = 0; it < iters; ++it) {
        f32x16 o[4];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[c][r] = 0.f;
        float rsum = 0.f;
        f32x16 sp, s_next;
        float p_prev[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) { sp[r] = 0.f; p_prev[r] = 0.f; }

        for (int j = -1; j <= ntiles; ++j) {
            const bool doS = j + 1 < ntiles, doE = j >= 0 && j < ntiles, doV = j >= 1;
            // tile that the NEXT step's S needs
            int nxt = -1;
            if (j + 2 < ntiles) nxt = j + 2;
            else if (j == ntiles && it + 1 < iters) nxt = 0;
            if (nxt >= 0 && !(ABL & 1)) stage_dma(nxt, n_staged++ % NBUF);

#pragma unroll
            for (int r = 0; r < 16; ++r) s_next[r] = 0.f;
            const float* xs = lds + (n_s % NBUF) * TILE;
            const float* xv = lds + (n_v % NBUF) * TILE;
            const unsigned km = (j == ntiles - 1) ? lastmask : 0xffffu;
            if (doS && doE && doV) p128_step<true, true, true, ABL>(xs, xv, li, hi, q, o, s_next, sp, p_prev, rsum, coef, km);
            else if (doS && doE) p128_step<true, true, false, ABL>(xs, xv, li, hi, q, o, s_next, sp, p_prev, rsum, coef, km);
            else if (doS) p128_step<true, false, false, ABL>(xs, xv, li, hi, q, o, s_next, sp, p_prev, rsum, coef, km);
            else if (doE && doV) p128_step<false, true, true, ABL>(xs, xv, li, hi, q, o, s_next, sp, p_prev, rsum, coef, km);
            else if (doE) p128_step<false, true, false, ABL>(xs, xv, li, hi, q, o, s_next, sp, p_prev, rsum, coef, km);
            else if (doV) p128_step<false, false, true, ABL>(xs, xv, li, hi, q, o, s_next, sp, p_prev, rsum, coef, km);
            n_s += doS;
            n_v += doV;
#pragma unroll
            for (int r = 0; r < 16; ++r) p_prev[r] = sp[r];        // P(j) feeds the next step's O product
            sp = s_next;                                             // S(j+1) is exponentiated next step
            if (!(ABL & 4)) __syncthreads();
        }

        // ---- row update (mean_shift.py:70-77)
        const float rs = rsum + xor32(rsum);
        const float Dinv = 1.0f / rs;
        float n2 = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float m = o[c][r] * Dinv - q[c][r];
                const float nq = q[c][r] + m;
                q[c][r] = nq;
                n2 += nq * nq;
            }
        n2 += xor32(n2);
        const float nrm = sqrtf(n2);
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) q[c][r] = q[c][r] / nrm;
    }

    if (qrow < N) {
        float* out = newX + ((size_t)cloud * N + qrow) * D;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            f32x4 v = {q[0][r], q[1][r], q[2][r], q[3][r]};
            *(f32x4*)(out + 4 * mfma_row(r, hi)) = v;
        }
    }
}

}  // namespace

extern "C" int sed_ms_iterate_f32(int B, int N, int d, int iters, const float* bw, const float* X, float* newX,
                                  hipStream_t stream) {
    if (B <= 0 || N <= 0 || iters < 0 || !bw || !X || !newX) return SED_EINVAL;
    if (d % 32 != 0 || d < 32 || d > 160) return SED_EUNSUPPORTED;
    dim3 grid((N + 127) / 128, B), block(256);
    switch (d / 32) {
        case 1: ms_iterate_kernel<1><<<grid, block, 0, stream>>>(X, newX, bw, N, iters); break;
        case 2: ms_iterate_kernel<2><<<grid, block, 0, stream>>>(X, newX, bw, N, iters); break;
        case 3: ms_iterate_kernel<3><<<grid, block, 0, stream>>>(X, newX, bw, N, iters); break;
        case 4: {
            static bool attr_set = false;          // 66 KiB of dynamic LDS needs the opt-in once per process
            if (!attr_set) {
                hipError_t e = hipFuncSetAttribute((const void*)ms_iterate_p128_kernel<0>,
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32 * 128 * 4);
                if (e != hipSuccess) return (int)e;
                attr_set = true;
            }
            ms_iterate_p128_kernel<0><<<grid, block, 4 * 32 * 128 * sizeof(float), stream>>>(X, newX, bw, N, iters);
            break;
        }
        case 5: ms_iterate_kernel<5><<<grid, block, 0, stream>>>(X, newX, bw, N, iters); break;
    }
    SED_LAUNCH_CHECK();
    return SED_OK;
}

// ablation hook for kernel tuning (not part of the public header): variant bit0 no DMA, bit1 no exp, bit2 no barrier,
// variant 8 = the non-pipelined generic kernel
extern "C" int sed_ms_iterate_variant(int variant, int B, int N, int iters, const float* bw, const float* X, float* newX,
                                      hipStream_t stream) {
    dim3 grid((N + 127) / 128, B), block(256);
    const size_t sm = 4 * 32 * 128 * sizeof(float);
#define V(n) case n: hipFuncSetAttribute((const void*)ms_iterate_p128_kernel<n>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm); \
                     ms_iterate_p128_kernel<n><<<grid, block, sm, stream>>>(X, newX, bw, N, iters); break;
    switch (variant) {
        V(0) V(1) V(2) V(3) V(4) V(5) V(6) V(7)
        case 8: ms_iterate_kernel<4><<<grid, block, 0, stream>>>(X, newX, bw, N, iters); break;
        case 9: ms_iterate_kernel<4, 1><<<grid, block, 0, stream>>>(X, newX, bw, N, iters); break;
        case 10: ms_iterate_kernel<4, 2><<<grid, block, 0, stream>>>(X, newX, bw, N, iters); break;
        case 12: ms_iterate_kernel<4, 4><<<grid, block, 0, stream>>>(X, newX, bw, N, iters); break;
        case 15: ms_iterate_kernel<4, 7><<<grid, block, 0, stream>>>(X, newX, bw, N, iters); break;
        default: return SED_EINVAL;
    }
#undef V
    SED_LAUNCH_CHECK();
    return SED_OK;
}
