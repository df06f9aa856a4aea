// Mean-shift iterations (d = 128) on the fp16 matrix pipe with fp32-equivalent arithmetic: split-fp16 emulation.
//
// Same mathematics as ms_iterate.hip (/root/reference/src/mean_shift.py:56-77, guard.py:7-9); what changes is how the
// two fp32 products  S = Q X^T  and  O = P X  are evaluated.  v_mfma_f32_32x32x2_f32 runs at the fp32 vector rate
// (157 TFLOP/s); v_mfma_f32_32x32x16_f16 is 16 x faster.  Every fp32 operand v is split (round to nearest) into
//     v * 2^s = h + l + e,   h = fp16(v 2^s),  l = fp16(v 2^s - h),  |e| <= 2^-24 |v 2^s|      (two 11-bit signed digits)
// and a product a.b is evaluated as  l_a h_b + h_a l_b + h_a h_b : three fp16 MFMAs (products of two fp16 values are
// exact in fp32, accumulation is the MFMA's fp32 accumulator).  What is dropped -- l_a l_b and the e terms -- is
// <= 3 * 2^-24 relative to |a||b| per product, i.e. the size of ONE fp32 rounding, whereas the fp32 fma chain it replaces
// rounds 128 (S) / 10 000 (O) times.  3 fp16 MFMAs instead of 16 fp32-rate units: 5.3 x less matrix time.
// Scales: X and Q by 2^11 (unit rows: |x| <= 1 -> |h| <= 2048, l stays in fp16's normal range for |x| >= 2^-14),
// P by 2^14 (weights <= 1; anything below 2^-38 flushes to 0: a relative change of a row sum (>= ~1) of <= N 2^-39).
// The exponent argument needs p 2^14 <= 65504, i.e. rows of norm <= 1: ms_split_kernel measures the row norms and
// flags clouds that violate (|x|^2 - 1) / b^2 <= 1; flagged clouds are skipped here and done by the exact fp32 kernel.
//
// Data movement: X is fixed over the 50 iterations, so a split kernel lays it out ONCE per call as a sequence of 32-key
// stage images; a stage image is copied to LDS by LDS-DMA (global_load_lds_dwordx4: linear copy, no staging registers),
// three buffers, one barrier per 32 keys. Two image formats:
//   * row-major only (StageLayoutN, 17 KiB): h / l planes of X [key][feature]; the second product's operands come from
//     the same planes through gfx950's transpose read -- the dense kernel (ms_iterate_d128_f16r_kernel);
//   * four planes (StageLayout<32>, 37 KiB): adds the h / l planes of X^T [feature][key] in MFMA slot order -- the
//     block-sparse kernel (ms_iterate_d128_f16s_kernel).
// One workgroup = 256 query rows (8 waves x 32) x all keys x all iterations; Q lives in registers as MFMA B operands,
// the O^T accumulator becomes the Q operand of the next iteration without leaving the registers.
// The kernels that lost their A/B during round 2 (first unpipelined version, staggered wave groups, four-plane dense form,
// fp8 correction term, row-major sparse form) are archived, not built: tools/experiments/ms_iterate_f16_round2.hip.
#include "common.h"
#include <type_traits>

namespace {

typedef _Float16 h16;
typedef h16 h16x8 __attribute__((ext_vector_type(8)));
typedef h16 h16x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// Four-plane stage image for KT keys (KT = 32: 37 KiB)
template <int KT_>
struct StageLayout {
    static constexpr int KT = KT_;
    static constexpr int XROW = 272;                     // bytes per key row of an X plane: 128 halves + 16 pad
    static constexpr int TROW = 2 * KT + 16;             // bytes per feature row of an X^T plane: KT halves + 16 pad
    static constexpr int XPLANE = KT * XROW;
    static constexpr int TPLANE = 128 * TROW;
    static constexpr int OFF_XH = 0, OFF_XL = XPLANE, OFF_TH = 2 * XPLANE, OFF_TL = 2 * XPLANE + TPLANE;
    static constexpr int STAGE = 2 * XPLANE + 2 * TPLANE;        // 71680 / 37888 B: whole 1 KiB DMA pieces
    static_assert(STAGE % 1024 == 0, "stage image must be a whole number of wave-sized DMA pieces");
};
constexpr float SCALE_X = 2048.0f;               // 2^11
constexpr float LOG2_SCALE_P = 14.0f;            // P is produced as 2^14 p
constexpr float UNSCALE_Q = 1.0f / 2048.0f;
constexpr float UNSCALE_O = 1.0f / 2048.0f;      // O carries 2^11 (X) * 2^14 (P); the row sum carries 2^14

// position of element m (0..31) of a 32-group in MFMA operand slot order: the accumulator row of register r on lane
// half hi is (r & 3) + 8 (r >> 2) + 4 hi; slot (j = r >> 3, hi, i = r & 7) sits at j * 16 + hi * 8 + i
__host__ __device__ constexpr int slot_pos(int m) {
    return (m >> 4) * 16 + ((m >> 2) & 1) * 8 + ((m >> 3) & 1) * 4 + (m & 3);
}

__device__ __forceinline__ f32x16 mfma16(h16x8 a, h16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// ------------------------------------------------------------------------------------------------------------
// X [B, N, 128] fp32 -> stage images [B, nst, STAGE] + per-cloud fallback flag. One workgroup per (stage, cloud).
template <int KT_>
__global__ __launch_bounds__(256) void ms_split_kernel(const float* __restrict__ X, const float* __restrict__ bw,
                                                       uint8_t* __restrict__ blob, int* __restrict__ flags, int N,
                                                       int nst) {
    using L = StageLayout<KT_>;
    constexpr int KT = L::KT, XROW = L::XROW, TROW = L::TROW, STAGE = L::STAGE;
    constexpr int OFF_XH = L::OFF_XH, OFF_XL = L::OFF_XL, OFF_TH = L::OFF_TH, OFF_TL = L::OFF_TL;
    extern __shared__ __attribute__((aligned(16))) uint8_t img[];    // STAGE bytes
    const int stage = blockIdx.x, cloud = blockIdx.y, tid = threadIdx.x;
    const float* Xc = X + (size_t)cloud * N * 128;
    for (int i = tid; i < STAGE / 16; i += 256) ((uint4*)img)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    float n2max = 0.f;
    for (int e = tid; e < KT * 32; e += 256) {             // one float4 of one key row per step
        const int kk = e >> 5, d0 = (e & 31) * 4;
        const int key = stage * KT + kk;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (key < N) v = *(const f32x4*)(Xc + (size_t)key * 128 + d0);
        float n2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) n2 += __shfl_xor(n2, off, 64);      // 32 lanes = one key row
        n2max = fmaxf(n2max, n2);
        const int sub = kk >> 5, km = kk & 31;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int d = d0 + u, c = d >> 5, m = d & 31;
            const float s = v[u] * SCALE_X;
            const h16 h = (h16)s;
            const h16 l = (h16)(s - (float)h);
            const int xo = kk * XROW + 2 * (c * 32 + slot_pos(m));
            const int to = d * TROW + 2 * (sub * 32 + slot_pos(km));
            *(h16*)(img + OFF_XH + xo) = h;
            *(h16*)(img + OFF_XL + xo) = l;
            *(h16*)(img + OFF_TH + to) = h;
            *(h16*)(img + OFF_TL + to) = l;
        }
    }
    const float b = bw[cloud];
    if (!((n2max - 1.0f) / (b * b) <= 1.0f)) atomicOr(flags + cloud, 1);           // also catches NaN rows
    __syncthreads();
    uint4* dst = (uint4*)(blob + ((size_t)cloud * nst + stage) * STAGE);
    for (int i = tid; i < STAGE / 16; i += 256) dst[i] = ((const uint4*)img)[i];
}

// ------------------------------------------------------------------------------------------------------------
// Dense schedule: one 8-wave workgroup per CU, 32-key stages, THREE LDS buffers.
//   * software pipeline inside every wave: the exponentials and fp16 splits of block n are issued BETWEEN the MFMAs of
//     block n + 1's first product, then block n's second product runs (VALU work of the OTHER wave of a SIMD does not run
//     under a wave's MFMAs on gfx950, independent VALU instructions of the SAME wave do: tools/micro/mfma_valu_overlap.hip);
//   * the A operands of both products travel through a 4-slot register ring loaded RD MFMA steps ahead of their use,
//     across the phase boundary and across blocks, so no ds_read latency sits in front of an MFMA (ring distance 3 spills
//     inside the loop with the 5-MFMA form: 836 instead of 337 ms);
//   * block n lives in buffer n % 3; one barrier per block, placed after the last step that still loads from the block's
//     buffer, after which block n + 3 is copied into that buffer.
// CHUNKED (few clouds per call: 40 workgroups per 10 000-point cloud cannot fill 256 CUs): one workgroup = 256 queries x
// ONE CHUNK of the stages x ONE iteration. Q is the current iterate `Qin` (fp32), the un-normalised partial (sum p x,
// sum p) goes to a workspace and ms_combine_kernel (ms_iterate.hip) finishes the iteration, one launch pair per iteration.
#ifndef F16Q_RING_DISTANCE
#define F16Q_RING_DISTANCE 2
#endif
// PL = false (weight_digits = 1, the default; 5 MFMAs per block pair instead of 6): the weights enter the second product --
// and the row sum, consistently -- as their fp16 heads only, O = sum_j fp16(2^14 p_j) (xh_j + xl_j) / sum_j fp16(2^14 p_j):
// an exactly evaluated weighted mean under weights perturbed by <= 2^-12 relative, independently per key. The perturbation
// of a row is sum_j p_j e_j (x_j - o) / sum_j p_j ~ 2^-12 / sqrt(3) * (spread of the keys under the kernel) /
// sqrt(#effective keys): 1e-7 .. 9e-7 on the golden snapshots (tools/f16split_emulation.py) against their 2e-6 / 5e-6 / 1e-5
// tolerances. The first product keeps its three terms: an error there is amplified by 1 / b^2. A cloud in which a weighted
// mean nearly cancels (|o| < 1/2 in any iteration) is flagged in `lowq` and redone by a PL = true launch (weight_digits = 2:
// (h, l) weights, fp32-equivalent), whose workgroups return at once for every other cloud.
//
// Stage images are ROW-MAJOR ONLY: the second product's A operand -- 8 keys of one feature per lane -- comes from the same
// [key][feature] planes the first product reads, through gfx950's transpose read (ds_read_b64_tr_b16): 17 KiB per 32-key stage
// (17 LDS-DMA pieces per block). Features are in natural order; the O^T accumulator -> Q operand hand-off costs one half-wave
// exchange per row and iteration.
struct StageLayoutN {
    static constexpr int XROW = 272, XPLANE = 32 * XROW, OFF_XH = 0, OFF_XL = XPLANE, STAGE = 2 * XPLANE;   // 17408 B
    static_assert(STAGE % 1024 == 0, "whole DMA pieces");
};
// accumulator row m = 16 a + 4 b + c  <->  image row sigma(m) = 16 a + 4 c + b (a 4 x 4 transpose inside every group of 16 rows)
__host__ __device__ constexpr int sigma_row(int m) { return 16 * (m >> 4) + 4 * (m & 3) + ((m >> 2) & 3); }


// X [B, N, 128] fp32 -> row-major stage images [B, nst, 17408] (h plane | l plane, rows = keys in natural order, 272 B apart)
__global__ __launch_bounds__(256) void ms_split_n_kernel(const float* __restrict__ X, const float* __restrict__ bw,
                                                         uint8_t* __restrict__ blob, int* __restrict__ flags, int N,
                                                         int nst) {
    using L = StageLayoutN;
    const int stage = blockIdx.x, cloud = blockIdx.y, tid = threadIdx.x;
    const float* Xc = X + (size_t)cloud * N * 128;
    uint8_t* dst = blob + ((size_t)cloud * nst + stage) * L::STAGE;
    float n2max = 0.f;
    for (int e = tid; e < 32 * 32; e += 256) {              // one float4 of one key row per step
        const int kk = e >> 5, d0 = (e & 31) * 4;
        const int key = stage * 32 + kk;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (key < N) v = *(const f32x4*)(Xc + (size_t)key * 128 + d0);
        float n2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) n2 += __shfl_xor(n2, off, 64);
        n2max = fmaxf(n2max, n2);
        typedef h16 h16x4 __attribute__((ext_vector_type(4)));
        h16x4 hh, ll;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float sc = v[u] * SCALE_X;
            const h16 h = (h16)sc;
            hh[u] = h;
            ll[u] = (h16)(sc - (float)h);
        }
        *(h16x4*)(dst + L::OFF_XH + kk * L::XROW + 2 * d0) = hh;
        *(h16x4*)(dst + L::OFF_XL + kk * L::XROW + 2 * d0) = ll;
    }
    if (tid < 64) {                                          // the 16 pad bytes of every row (never read as data)
        const int kk = tid & 31, pl = tid >> 5;
        *(uint4*)(dst + pl * L::XPLANE + kk * L::XROW + 256) = make_uint4(0, 0, 0, 0);
    }
    const float b = bw[cloud];
    if (!((n2max - 1.0f) / (b * b) <= 1.0f)) atomicOr(flags + cloud, 1);
}

template <bool CHUNKED = false, bool PL = true>
__global__ __launch_bounds__(512, 1) void ms_iterate_d128_f16r_kernel(const float* __restrict__ X,
                                                                      const uint8_t* __restrict__ blob,
                                                                      float* __restrict__ newX,
                                                                      const float* __restrict__ bw,
                                                                      const int* __restrict__ flags, int N, int iters,
                                                                      const float* __restrict__ Qin = nullptr,
                                                                      float* __restrict__ partO = nullptr,
                                                                      float* __restrict__ partS = nullptr,
                                                                      int* __restrict__ lowq = nullptr) {
    using L = StageLayoutN;
    constexpr int XROW = L::XROW, STAGE = L::STAGE, NPIECE = L::STAGE / 1024;
    constexpr int NBUF = 3;
    constexpr int RD = F16Q_RING_DISTANCE;                              // the operand ring runs RD steps ahead
    constexpr int OFF_XH = L::OFF_XH, OFF_XL = L::OFF_XL;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];    // [3][STAGE]
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;      // wave id in an SGPR
    const int li = lane & 31, hi = lane >> 5;
    int bx;
    const int cloud = sed_xcd_cloud_block(&bx);
    if (flags[cloud]) return;
    if (PL && lowq != nullptr && !lowq[cloud]) return;
    const float* Xc = (CHUNKED ? Qin : X) + (size_t)cloud * N * 128;       // where the query rows come from
    const int nst = (N + 31) >> 5;
    const size_t bstride = (size_t)STAGE;
    const uint8_t* blob_c = blob + (size_t)cloud * nst * bstride;
    const int qrow = bx * 256 + wave * 32 + li;
    const int qrow_c = qrow < N ? qrow : N - 1;
    const int nchunk = CHUNKED ? gridDim.z : 1, chunk = CHUNKED ? blockIdx.z : 0;
    const int s0 = (int)((long)chunk * nst / nchunk), s1 = (int)((long)(chunk + 1) * nst / nchunk);

    const float b = bw[cloud];
    const float inv_b2_l2e = 1.44269504088896340736f / (b * b);
    const float K1 = inv_b2_l2e * (1.0f / 4194304.0f);
    const float K0 = LOG2_SCALE_P - inv_b2_l2e;
    const float TMIN = LOG2_SCALE_P - 75.0f * 1.44269504088896340736f;

    h16x8 qh[8], ql[8];
    auto split_q = [&](int ks, const float* v) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const h16 h = (h16)v[i];
            qh[ks][i] = h;
            ql[ks][i] = (h16)(v[i] - (float)h);
        }
    };
    // Q operand of k-step ks: features 16 ks + 8 hi + i in natural order (the stage images are row-major, unpermuted)
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        float v[8];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const f32x4 t = *(const f32x4*)(Xc + (size_t)qrow_c * 128 + 16 * ks + 8 * hi + 4 * g);
#pragma unroll
            for (int u = 0; u < 4; ++u) v[4 * g + u] = t[u] * SCALE_X;
        }
        split_q(ks, v);
    }

    // DMA pieces (1 KiB each) of a stage image: wave w moves pieces 2 w, 2 w + 1 (immediate offset), wave 0 also piece 16
    static_assert(NPIECE == 17, "piece distribution below is written for 17 pieces");
    const unsigned lane16 = lane * 16;
    auto stage_dma = [&](int st, int buf) {
        const uint8_t* src = blob_c + (size_t)st * bstride;
        uint8_t* dst = lds + buf * STAGE;
        const auto g = (const __attribute__((address_space(1))) void*)(src + wave * 2048 + lane16);
        const auto l = (__attribute__((address_space(3))) void*)(dst + wave * 2048);
        __builtin_amdgcn_global_load_lds(g, l, 16, 0, 0);
        __builtin_amdgcn_global_load_lds(g, l, 16, 1024, 0);
        if (wave == 0)                                 // piece 16
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 16 * 1024 + lane16),
                                             (__attribute__((address_space(3))) void*)(dst + 16 * 1024), 16, 0, 0);
    };
    // ping-pong stage sequence 0 .. nst-1, nst-1 .. 0, 0 .. : one step
    auto advance = [&](int& st, bool& fwd) {
        if (fwd) {
            if (st == nst - 1) fwd = false; else ++st;
        } else {
            if (st == 0) fwd = true; else --st;
        }
    };

    const int total = CHUNKED ? s1 - s0 : iters * nst;   // CHUNKED: stages s0 .. s1 - 1 in order, once
    int st_cur = s0, st_dma = s0;
    bool fwd_cur = true, fwd_dma = true;
    if (total > 0) stage_dma(s0, 0);
    advance(st_dma, fwd_dma);
    if (total > 1) stage_dma(st_dma, 1);
    advance(st_dma, fwd_dma);
    if (total > 2) stage_dma(st_dma, 2);
    advance(st_dma, fwd_dma);                            // st_dma = stage of block 3
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    h16x8 fa[4], fb[4];
    // Per-lane LDS offsets of the operand reads (recomputed per block from a lane id the compiler cannot hoist: invariants
    // that live across the register-hungry row update end up in scratch, and a reload's vmcnt wait in the hot loop also waits
    // for the stage copy in flight).
    //   first product: accumulator row m reads image row sigma(m), sigma(16 a + 4 b + c) = 16 a + 4 c + b -- a permutation
    //     inside each group of 16 rows, conflict-free for ds_read_b128 (its four 16-lane groups see 16 distinct rows mod 16,
    //     272-byte rows = 4 banks apart), chosen so that
    //   second product: the four keys of one transpose read (accumulator rows rho .. rho + 3) sit in image rows FOUR apart
    //     (16 banks): ds_read_b64_tr_b16 -- every lane passes the address of 4 consecutive features of one key, a 16-lane
    //     group gets back the 4 keys x 16 features block transposed: lane = feature, 4 keys. Its conflict groups are the two
    //     32-lane halves: lanes 0-15 (features 0-15 of the tile) and 16-31 (features 16-31, 8 banks further) read the same
    //     four rows, so rows 16 banks apart make the 64 dwords of a half distinct. (The first version put the rows two
    //     apart, 8 banks: lanes 16-31 then collided with the next row of lanes 0-15 -- SQ_LDS_BANK_CONFLICT 1.3 extra cycles
    //     per LDS instruction, a third of the LDS-array cycles.) No transposed planes in the image: 17 KiB per stage instead of 37.
    int xoff, toff;
    auto refresh_offsets = [&]() {
        int l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        const int m = l & 31, h = l >> 5;
        const int sig = 16 * (m >> 4) + 4 * (m & 3) + ((m >> 2) & 3);
        xoff = sig * XROW + h * 16;
        const int i16 = l & 15;
        toff = (4 * (i16 >> 2) + h) * XROW + 32 * ((l >> 4) & 1) + 8 * (i16 & 3);
    };
    refresh_offsets();
    typedef short v4s __attribute__((__vector_size__(4 * sizeof(short))));
    auto tr8 = [&](const uint8_t* plane, int c, int j) {       // keys 16 j + {4 hi .. + 3, 8 + 4 hi .. + 3} of feature 32 c + li
        const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) v4s*)(plane + toff + (16 * j) * XROW + 64 * c));
        const v4s hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) v4s*)(plane + toff + (16 * j + 2) * XROW + 64 * c));
        typedef short v8s __attribute__((__vector_size__(8 * sizeof(short))));
        const v8s both = __builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(h16x8, both);
    };
    auto ring_load = [&](int t, const uint8_t* base) {      // t in 0..15, compile-time after unrolling
        if (t < 8) {
            fa[t & 3] = *(const h16x8*)(base + OFF_XH + xoff + t * 32);
            fb[t & 3] = *(const h16x8*)(base + OFF_XL + xoff + t * 32);
        } else {
            const int c = (t - 8) >> 1, j = (t - 8) & 1;
            fa[t & 3] = tr8(base + OFF_XH, c, j);
            fb[t & 3] = tr8(base + OFF_XL, c, j);
        }
    };
    // first product of a block outside the pipeline (first block of the launch and of every sweep: Q has just changed),
    // operands read directly; same MFMA order as the pipelined form -> same bits
    auto plain_first_product = [&](const uint8_t* base) {
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const h16x8 a = *(const h16x8*)(base + OFF_XH + xoff + t * 32);
            const h16x8 l = *(const h16x8*)(base + OFF_XL + xoff + t * 32);
            s = mfma16(l, qh[t], s);
            s = mfma16(a, ql[t], s);
            s = mfma16(a, qh[t], s);
        }
        return s;
    };

    f32x16 o[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[c][r] = 0.f;
    float rsum = 0.f;
    int buf = 0;                                          // buffer of the current block = n % NBUF
    i32x4 phv[2], plv[2];                                 // weights of the current block: two accumulator rows (fp16 pair) per dword
    f32x16 s_cur;
#pragma unroll
    for (int r = 0; r < 16; ++r) s_cur[r] = 0.f;
    if (total > 0) s_cur = plain_first_product(lds);
    if (total > 1) {
#pragma unroll
        for (int t = 0; t < RD; ++t) ring_load(t, lds + STAGE);
    }

    for (int n = 0; n < total; ++n) {
        const uint8_t* base = lds + buf * STAGE;
        const int nbuf = buf == NBUF - 1 ? 0 : buf + 1;
        const uint8_t* nbase = lds + nbuf * STAGE;
        const uint8_t* n2base = lds + (nbuf == NBUF - 1 ? 0 : nbuf + 1) * STAGE;
        const int key0 = st_cur * 32;
        const bool sweep_end = !CHUNKED && (fwd_cur ? st_cur == nst - 1 : st_cur == 0);
        const bool has_next = n + 1 < total && !sweep_end;
        const bool tail = key0 + 32 > N;
        refresh_offsets();

        // weights of block n from s_cur, two accumulator rows at a time; TAIL: the cloud's last, partly filled stage
        auto weights2 = [&](int t, auto tail_c) {
            float p[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int r = 2 * t + u;
                p[u] = __builtin_amdgcn_exp2f(fmaxf(fmaf(s_cur[r], K1, K0), TMIN));
                if (decltype(tail_c)::value && key0 + sigma_row(mfma_row(r, hi)) >= N) p[u] = 0.f;
                if (PL) rsum += p[u];
            }
            const h16x2 h = {(h16)p[0], (h16)p[1]};
            phv[t >> 2][t & 3] = __builtin_bit_cast(int, h);
            if (PL) {
                const h16x2 l = {(h16)(p[0] - (float)h[0]), (h16)(p[1] - (float)h[1])};
                plv[t >> 2][t & 3] = __builtin_bit_cast(int, l);
            } else {
                // (one v_dot2c_f32_f16 against (1, 1) per pair instead of two conversions + two adds was tried in the f16r kernel: 361 vs
                // 287 ms -- the dot instruction is slow beside the MFMAs and the allocator moved a reload into the loop)
                rsum += (float)h[0] + (float)h[1];
            }
        };

        // ---- phase 1: first product of block n + 1 with the exponentials and splits of block n BETWEEN its MFMAs.
        // (Measured on gfx950, tools/micro/mfma_valu_overlap.hip: VALU work of ANOTHER wave of the SIMD does not run under
        // a wave's MFMAs -- 94 % of the serial time -- while independent VALU instructions interleaved into the SAME wave's
        // MFMA stream do; the earlier schedules, staggered or not, ran matrix and vector phases back to back.)
        f32x16 s_next;
        auto phase1 = [&](auto tail_c) {
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                if (t == 0) {
                    f32x16 z;
#pragma unroll
                    for (int r = 0; r < 16; ++r) z[r] = 0.f;
                    s_next = mfma16(fb[0], qh[0], z);
                } else {
                    s_next = mfma16(fb[t & 3], qh[t], s_next);
                }
                weights2(t, tail_c);
                s_next = mfma16(fa[t & 3], ql[t], s_next);
                s_next = mfma16(fa[t & 3], qh[t], s_next);
                if (t + RD < 8) ring_load(t + RD, nbase);
                else ring_load(t + RD, base);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if (has_next) {
            if (tail) phase1(std::true_type{});
            else phase1(std::false_type{});
        } else {                                          // last block of a sweep / of the launch: nothing to overlap with
#pragma unroll
            for (int r = 0; r < 16; ++r) s_next[r] = 0.f;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                if (tail) weights2(t, std::true_type{});
                else weights2(t, std::false_type{});
                if (t + RD >= 8) ring_load(t + RD, base);
            }
            __builtin_amdgcn_sched_barrier(0);
        }

        // ---- phase 2: second product of block n. After step 13 nobody reads buffer n any more and everybody's pieces of
        // block n + 2 (issued a block ago) have landed: barrier, then block n + 3 -> buffer n, and the ring moves on to
        // block n + 2's first-product operands.
#pragma unroll
        for (int t = 8; t < 16; ++t) {
            const int c = (t - 8) >> 1, j = (t - 8) & 1;
            const h16x8 phj = __builtin_bit_cast(h16x8, phv[j]);
            o[c] = mfma16(fb[t & 3], phj, o[c]);
            if (PL) {
                const h16x8 plj = __builtin_bit_cast(h16x8, plv[j]);
                o[c] = mfma16(fa[t & 3], plj, o[c]);
            }
            o[c] = mfma16(fa[t & 3], phj, o[c]);
            if (t + RD < 16) ring_load(t + RD, base);
            else if (n + 2 < total) ring_load(t + RD - 16, n2base);
            __builtin_amdgcn_sched_barrier(0);
            if (t == 15 - RD) {                            // the last step that loads from this block's buffer
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (n + 3 < total) stage_dma(st_dma, buf);          // three buffers: block n + 3 replaces block n
                advance(st_dma, fwd_dma);
            }
        }

        advance(st_cur, fwd_cur);
        buf = nbuf;
        if (!sweep_end) {
            s_cur = s_next;
            continue;
        }
        // ---- end of a sweep: row update (mean_shift.py:70-77)
        const float rs = rsum + xor32(rsum);
        const float Dinv = UNSCALE_O / rs;
        // current Q in the accumulator layout (inverse of the hand-off below)
        float qacc[4][16];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float e0 = ((float)qh[2 * c + j][u] + (float)ql[2 * c + j][u]) * UNSCALE_Q;           // g = 0
                    const float e1 = ((float)qh[2 * c + j][4 + u] + (float)ql[2 * c + j][4 + u]) * UNSCALE_Q;   // g = 1
                    const float keep = hi ? e1 : e0, send = hi ? e0 : e1;
                    const float recv = __shfl_xor(send, 32, 64);
                    // own half g = hi holds register 4 (2 j + hi) + u; the partner's element is register 4 (2 j + 1 - hi) + u
                    qacc[c][8 * j + u] = hi ? recv : keep;
                    qacc[c][8 * j + 4 + u] = hi ? keep : recv;
                }
        float n2 = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float q = qacc[c][r];
                const float m = o[c][r] * Dinv - q;
                const float nq = q + m;
                o[c][r] = nq;
                n2 += nq * nq;
            }
        n2 += xor32(n2);
        const float nrm = sqrtf(n2);
        if (!PL && lowq != nullptr && nrm < 0.5f) lowq[cloud] = 1;
        if (n == total - 1) {
            if (qrow < N) {
                float* out = newX + ((size_t)cloud * N + qrow) * 128;
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f32x4 v = {o[c][4 * g] / nrm, o[c][4 * g + 1] / nrm, o[c][4 * g + 2] / nrm, o[c][4 * g + 3] / nrm};
                        *(f32x4*)(out + 32 * c + 8 * g + 4 * hi) = v;
                    }
            }
        } else {
            // accumulator (feature 32 c + (r & 3) + 8 (r >> 2) + 4 hi) -> Q operand (feature 16 ks + 8 hi + i): element
            // i = 4 g + u of k-step 2 c + j is register 4 (2 j + hi) + u of lane half g -- own half for g = hi, the partner
            // lane's otherwise (one exchange per row and iteration)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float v[8];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float a = (o[c][8 * j + u] / nrm) * SCALE_X, bq = (o[c][8 * j + 4 + u] / nrm) * SCALE_X;
                        const float keep = hi ? bq : a, send = hi ? a : bq;
                        const float recv = __shfl_xor(send, 32, 64);
                        v[u] = hi ? recv : keep;
                        v[4 + u] = hi ? keep : recv;
                    }
                    split_q(2 * c + j, v);
                }
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[c][r] = 0.f;
            rsum = 0.f;
            s_cur = plain_first_product(lds + buf * STAGE);          // first block of the next sweep, new Q
        }
    }
    if (CHUNKED) {
        // partial of this chunk, unscaled: O carries 2^11 (X) * 2^14 (P), the row sum 2^14
        const float rs = rsum + xor32(rsum);
        if (qrow < N) {
            const size_t slot = ((size_t)cloud * N + qrow) * nchunk + chunk;
            float* out = partO + slot * 128;
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    constexpr float U = 1.0f / 33554432.0f;          // 2^-25
                    f32x4 v = {o[c][4 * g] * U, o[c][4 * g + 1] * U, o[c][4 * g + 2] * U, o[c][4 * g + 3] * U};
                    *(f32x4*)(out + 32 * c + 8 * g + 4 * hi) = v;
                }
            if (hi == 0) partS[slot] = rs * (1.0f / 16384.0f);
        }
        return;
    }
    if (iters == 0 && qrow < N) {
        float* out = newX + ((size_t)cloud * N + qrow) * 128;
        const float* in = Xc + (size_t)qrow * 128;
        for (int d = 4 * hi; d < 128; d += 8) *(f32x4*)(out + d) = *(const f32x4*)(in + d);
    }
}

// ------------------------------------------------------------------------------------------------------------
// The same schedule with 64 QUERIES PER WAVE (round 3): one 4-wave workgroup per CU, one wave per SIMD, the whole
// 512-register file per wave. Every A operand read from LDS (a key tile's h / l fragment) now feeds the MFMAs of TWO 32-query
// groups, so the LDS traffic per MFMA halves (the 8-wave kernel keeps the LDS pipe ~50 % busy), and Q (128 registers), the O^T
// accumulators (128) and the two S accumulator pairs (64) live in registers without scratch -- the 8-wave kernel at 256
// registers per wave spills 65 of them around the row update. Per accumulator the MFMAs are issued in
// exactly the order of ms_iterate_d128_f16r_kernel (per k-step: x_l q_h, x_h q_l, x_h q_h; per feature tile and key half:
// x_l p, [x_h p_l,] x_h p), the keys of a stage are visited in the same image order and the row update is the same code:
// the two kernels return the SAME BITS (tests/test_gpu_mean_shift.py::test_wide_wave_kernel_is_bit_identical).
// With one wave per SIMD nothing hides a wave's latencies but its own software pipeline: the exponentials of block n sit
// between the MFMAs of block n + 1's first product (<= 5 single-issue instructions per MFMA gap), operands travel through the
// same register ring RD steps ahead, one barrier per block.
#ifndef F16W_RING_DISTANCE
#define F16W_RING_DISTANCE 1
#endif
// NT = feature tiles of 32: 4 (d = 128, the SED-Net embedding) or 5 (d = 160: the 140 columns of the HPNet-widened embedding,
// generate_predictions_aug.py:371-377, zero padded). Row-major stage images of 32 keys x NT * 32 features, rows padded by 16 B.
template <int NT>
struct StageLayoutD {
    static constexpr int D = 32 * NT, XROW = 2 * D + 16, XPLANE = 32 * XROW, OFF_XH = 0, OFF_XL = XPLANE, STAGE = 2 * XPLANE;
    static_assert(STAGE % 1024 == 0, "whole DMA pieces");             // 17408 B (NT = 4), 21504 B (NT = 5)
};
static_assert(StageLayoutD<4>::STAGE == StageLayoutN::STAGE && StageLayoutD<4>::XROW == StageLayoutN::XROW, "same images at d = 128");

template <int NT>
__global__ __launch_bounds__(256) void ms_split_d_kernel(const float* __restrict__ X, const float* __restrict__ bw,
                                                         uint8_t* __restrict__ blob, int* __restrict__ flags, int N, int nst) {
    using L = StageLayoutD<NT>;
    constexpr int D = L::D, Q4 = D / 4;                       // float4s per row
    const int stage = blockIdx.x, cloud = blockIdx.y, tid = threadIdx.x;
    const float* Xc = X + (size_t)cloud * N * D;
    uint8_t* dst = blob + ((size_t)cloud * nst + stage) * L::STAGE;
    __shared__ float n2row[32];
    if (tid < 32) n2row[tid] = 0.f;
    __syncthreads();
    for (int e = tid; e < 32 * Q4; e += 256) {
        const int kk = e / Q4, d0 = (e - kk * Q4) * 4;
        const int key = stage * 32 + kk;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (key < N) v = *(const f32x4*)(Xc + (size_t)key * D + d0);
        atomicAdd(&n2row[kk], v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]);   // only compared with a threshold: order-free
        typedef h16 h16x4 __attribute__((ext_vector_type(4)));
        h16x4 hh, ll;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float sc = v[u] * SCALE_X;
            const h16 h = (h16)sc;
            hh[u] = h;
            ll[u] = (h16)(sc - (float)h);
        }
        *(h16x4*)(dst + L::OFF_XH + kk * L::XROW + 2 * d0) = hh;
        *(h16x4*)(dst + L::OFF_XL + kk * L::XROW + 2 * d0) = ll;
    }
    if (tid < 64) {
        const int kk = tid & 31, pl = tid >> 5;
        *(uint4*)(dst + pl * L::XPLANE + kk * L::XROW + 2 * D) = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
    if (tid < 32) {
        const float b = bw[cloud];
        if (!((n2row[tid] - 1.0f) / (b * b) <= 1.0f)) atomicOr(flags + cloud, 1);
    }
}

template <int NT = 4, bool CHUNKED = false, bool PL = true>
__global__ __launch_bounds__(256, 1) void ms_iterate_f16w_kernel(const float* __restrict__ X,
                                                                      const uint8_t* __restrict__ blob,
                                                                      float* __restrict__ newX,
                                                                      const float* __restrict__ bw,
                                                                      const int* __restrict__ flags, int N, int iters,
                                                                      const float* __restrict__ Qin = nullptr,
                                                                      float* __restrict__ partO = nullptr,
                                                                      float* __restrict__ partS = nullptr,
                                                                      int* __restrict__ lowq = nullptr) {
    using L = StageLayoutD<NT>;
    constexpr int D = L::D, KS = 2 * NT, NSTEP = 4 * NT;  // k-steps of the first product, operand steps of a block
    constexpr int XROW = L::XROW, STAGE = L::STAGE, NPIECE = L::STAGE / 1024;
    constexpr int NBUF = 3;
    constexpr int RD = F16W_RING_DISTANCE;             // one step ahead = 6 / 4 MFMAs (192 / 128 matrix cycles) per operand pair
    constexpr int OFF_XH = L::OFF_XH, OFF_XL = L::OFF_XL;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];    // [3][STAGE]
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int li = lane & 31, hi = lane >> 5;
    int bx;
    const int cloud = sed_xcd_cloud_block(&bx);
    if (flags[cloud]) return;
    if (PL && lowq != nullptr && !lowq[cloud]) return;
    const float* Xc = (CHUNKED ? Qin : X) + (size_t)cloud * N * D;
    const int nst = (N + 31) >> 5;
    const uint8_t* blob_c = blob + (size_t)cloud * nst * STAGE;
    int qrow[2], qrow_c[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        qrow[g] = bx * 256 + wave * 64 + g * 32 + li;
        qrow_c[g] = qrow[g] < N ? qrow[g] : N - 1;
    }
    const int nchunk = CHUNKED ? gridDim.z : 1, chunk = CHUNKED ? blockIdx.z : 0;
    const int s0 = (int)((long)chunk * nst / nchunk), s1 = (int)((long)(chunk + 1) * nst / nchunk);

    const float b = bw[cloud];
    const float inv_b2_l2e = 1.44269504088896340736f / (b * b);
    const float K1 = inv_b2_l2e * (1.0f / 4194304.0f);
    const float K0 = LOG2_SCALE_P - inv_b2_l2e;
    const float TMIN = LOG2_SCALE_P - 75.0f * 1.44269504088896340736f;

    h16x8 qh[2][KS], ql[2][KS];
    auto split_q = [&](int g, int ks, const float* v) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const h16 h = (h16)v[i];
            qh[g][ks][i] = h;
            ql[g][ks][i] = (h16)(v[i] - (float)h);
        }
    };
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            float v[8];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const f32x4 t = *(const f32x4*)(Xc + (size_t)qrow_c[g] * D + 16 * ks + 8 * hi + 4 * q);
#pragma unroll
                for (int u = 0; u < 4; ++u) v[4 * q + u] = t[u] * SCALE_X;
            }
            split_q(g, ks, v);
        }

    // DMA pieces (1 KiB each) of a stage image: wave w moves pieces PW w .. PW w + PW - 1 (immediate offsets), wave 0 also the last
    constexpr int PW = NPIECE / 4;
    static_assert(NPIECE == 4 * PW + 1 && (PW == 4 || PW == 5), "piece distribution below is written for 17 / 21 pieces");
    const unsigned lane16 = lane * 16;
    auto stage_dma = [&](int st, int buf) {
        const uint8_t* src = blob_c + (size_t)st * STAGE;
        uint8_t* dst = lds + buf * STAGE;
        const auto g = (const __attribute__((address_space(1))) void*)(src + wave * (PW * 1024) + lane16);
        const auto l = (__attribute__((address_space(3))) void*)(dst + wave * (PW * 1024));
        __builtin_amdgcn_global_load_lds(g, l, 16, 0, 0);
        __builtin_amdgcn_global_load_lds(g, l, 16, 1024, 0);
        __builtin_amdgcn_global_load_lds(g, l, 16, 2048, 0);
        __builtin_amdgcn_global_load_lds(g, l, 16, 3072, 0);
        if constexpr (PW == 5)        // (own base pointers: the instruction's immediate offset is 13-bit signed, 4096 does not fit)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + wave * (PW * 1024) + 4096 + lane16),
                                             (__attribute__((address_space(3))) void*)(dst + wave * (PW * 1024) + 4096), 16, 0, 0);
        if (wave == 0)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 4 * PW * 1024 + lane16),
                                             (__attribute__((address_space(3))) void*)(dst + 4 * PW * 1024), 16, 0, 0);
    };
    auto advance = [&](int& st, bool& fwd) {
        if (fwd) {
            if (st == nst - 1) fwd = false; else ++st;
        } else {
            if (st == 0) fwd = true; else --st;
        }
    };

    const int total = CHUNKED ? s1 - s0 : iters * nst;
    int st_cur = s0, st_dma = s0;
    bool fwd_cur = true, fwd_dma = true;
    if (total > 0) stage_dma(s0, 0);
    advance(st_dma, fwd_dma);
    if (total > 1) stage_dma(st_dma, 1);
    advance(st_dma, fwd_dma);
    if (total > 2) stage_dma(st_dma, 2);
    advance(st_dma, fwd_dma);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    h16x8 fa[4], fb[4];
    int xoff, toff;                                       // see ms_iterate_d128_f16r_kernel: same image rows, same permutation
    auto refresh_offsets = [&]() {
        int l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        const int m = l & 31, h = l >> 5;
        const int sig = 16 * (m >> 4) + 4 * (m & 3) + ((m >> 2) & 3);
        xoff = sig * XROW + h * 16;
        const int i16 = l & 15;
        toff = (4 * (i16 >> 2) + h) * XROW + 32 * ((l >> 4) & 1) + 8 * (i16 & 3);
    };
    refresh_offsets();
    typedef short v4s __attribute__((__vector_size__(4 * sizeof(short))));
    auto tr8 = [&](const uint8_t* plane, int c, int j) {
        const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) v4s*)(plane + toff + (16 * j) * XROW + 64 * c));
        const v4s hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) v4s*)(plane + toff + (16 * j + 2) * XROW + 64 * c));
        typedef short v8s __attribute__((__vector_size__(8 * sizeof(short))));
        const v8s both = __builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(h16x8, both);
    };
    auto ring_load = [&](int t, const uint8_t* base) {
        if (t < KS) {
            fa[t & 3] = *(const h16x8*)(base + OFF_XH + xoff + t * 32);
            fb[t & 3] = *(const h16x8*)(base + OFF_XL + xoff + t * 32);
        } else {
            const int c = (t - KS) >> 1, j = (t - KS) & 1;
            fa[t & 3] = tr8(base + OFF_XH, c, j);
            fb[t & 3] = tr8(base + OFF_XL, c, j);
        }
    };
    f32x16 s_cur[2], s_next[2];
    auto plain_first_product = [&](const uint8_t* base) {
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) s_cur[g][r] = 0.f;
#pragma unroll
        for (int t = 0; t < KS; ++t) {
            const h16x8 a = *(const h16x8*)(base + OFF_XH + xoff + t * 32);
            const h16x8 l = *(const h16x8*)(base + OFF_XL + xoff + t * 32);
#pragma unroll
            for (int g = 0; g < 2; ++g) s_cur[g] = mfma16(l, qh[g][t], s_cur[g]);
#pragma unroll
            for (int g = 0; g < 2; ++g) s_cur[g] = mfma16(a, ql[g][t], s_cur[g]);
#pragma unroll
            for (int g = 0; g < 2; ++g) s_cur[g] = mfma16(a, qh[g][t], s_cur[g]);
        }
    };

    f32x16 o[2][NT];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int c = 0; c < NT; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[g][c][r] = 0.f;
    float rsum[2] = {0.f, 0.f};
    int buf = 0;
    i32x4 phv[2][2], plv[2][2];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) { s_cur[g][r] = 0.f; s_next[g][r] = 0.f; }
    if (total > 0) plain_first_product(lds);
    if (total > 1) {
#pragma unroll
        for (int t = 0; t < RD; ++t) ring_load(t, lds + STAGE);
    }

    // One block of the pipeline. HAS_NEXT: the block is followed by another one of the same sweep (whose first product runs here,
    // under this block's weights). The blocks of a sweep form the INNER loop and the row update sits between sweeps, outside it:
    // with everything in one flat loop the register allocator weighed the row update like the hot path and spilled inside it.
    int n = 0;                                            // blocks done (the DMA runs three blocks ahead of it)
    auto block = [&](auto has_next_c) __attribute__((always_inline)) {
        constexpr bool has_next = decltype(has_next_c)::value;
        const uint8_t* base = lds + buf * STAGE;
        const int nbuf = buf == NBUF - 1 ? 0 : buf + 1;
        const uint8_t* nbase = lds + nbuf * STAGE;
        const uint8_t* n2base = lds + (nbuf == NBUF - 1 ? 0 : nbuf + 1) * STAGE;
        const int key0 = st_cur * 32;
        const bool tail = key0 + 32 > N;
        refresh_offsets();

        auto weights2 = [&](int g, int t, auto tail_c) {
            float p[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int r = 2 * t + u;
                p[u] = __builtin_amdgcn_exp2f(fmaxf(fmaf(s_cur[g][r], K1, K0), TMIN));
                if (decltype(tail_c)::value && key0 + sigma_row(mfma_row(r, hi)) >= N) p[u] = 0.f;
                if (PL) rsum[g] += p[u];
            }
            const h16x2 h = {(h16)p[0], (h16)p[1]};
            phv[g][t >> 2][t & 3] = __builtin_bit_cast(int, h);
            if (PL) {
                const h16x2 l = {(h16)(p[0] - (float)h[0]), (h16)(p[1] - (float)h[1])};
                plv[g][t >> 2][t & 3] = __builtin_bit_cast(int, l);
            } else {
                rsum[g] += (float)h[0] + (float)h[1];
            }
        };

        // ---- phase 1: first product of block n + 1 (both query groups) with the weights of block n between its MFMAs
        auto phase1 = [&](auto tail_c) {
#pragma unroll
            for (int t = 0; t < KS; ++t) {
                if (t == 0) {
                    f32x16 z;
#pragma unroll
                    for (int r = 0; r < 16; ++r) z[r] = 0.f;
                    s_next[0] = mfma16(fb[0], qh[0][0], z);
                    s_next[1] = mfma16(fb[0], qh[1][0], z);
                } else {
                    s_next[0] = mfma16(fb[t & 3], qh[0][t], s_next[0]);
                    s_next[1] = mfma16(fb[t & 3], qh[1][t], s_next[1]);
                }
                if (t < 8) weights2(0, t, tail_c);              // 16 accumulator rows = 8 pairs
                s_next[0] = mfma16(fa[t & 3], ql[0][t], s_next[0]);
                s_next[1] = mfma16(fa[t & 3], ql[1][t], s_next[1]);
                if (t < 8) weights2(1, t, tail_c);
                s_next[0] = mfma16(fa[t & 3], qh[0][t], s_next[0]);
                s_next[1] = mfma16(fa[t & 3], qh[1][t], s_next[1]);
                if (t + RD < KS) ring_load(t + RD, nbase);
                else ring_load(t + RD, base);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if constexpr (has_next) {
            if (tail) phase1(std::true_type{});
            else phase1(std::false_type{});
        } else {
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int r = 0; r < 16; ++r) s_next[g][r] = 0.f;
#pragma unroll
            for (int t = 0; t < 8; ++t) {                          // (16 accumulator rows = 8 pairs, whatever the feature width)
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    if (tail) weights2(g, t, std::true_type{});
                    else weights2(g, t, std::false_type{});
                }
            }
#pragma unroll
            for (int t = 0; t < KS; ++t)
                if (t + RD >= KS) ring_load(t + RD, base);
            __builtin_amdgcn_sched_barrier(0);
        }

        // ---- phase 2: second product of block n, both query groups per operand read
#pragma unroll
        for (int t = KS; t < NSTEP; ++t) {
            const int c = (t - KS) >> 1, j = (t - KS) & 1;
#pragma unroll
            for (int g = 0; g < 2; ++g) o[g][c] = mfma16(fb[t & 3], __builtin_bit_cast(h16x8, phv[g][j]), o[g][c]);
            if (PL) {
#pragma unroll
                for (int g = 0; g < 2; ++g) o[g][c] = mfma16(fa[t & 3], __builtin_bit_cast(h16x8, plv[g][j]), o[g][c]);
            }
#pragma unroll
            for (int g = 0; g < 2; ++g) o[g][c] = mfma16(fa[t & 3], __builtin_bit_cast(h16x8, phv[g][j]), o[g][c]);
            if (t + RD < NSTEP) ring_load(t + RD, base);
            else if (n + 2 < total) ring_load(t + RD - NSTEP, n2base);
            __builtin_amdgcn_sched_barrier(0);
            if (t == NSTEP - 1 - RD) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (n + 3 < total) stage_dma(st_dma, buf);
                advance(st_dma, fwd_dma);
            }
        }

        advance(st_cur, fwd_cur);
        buf = nbuf;
        ++n;
        if constexpr (has_next) {
            s_cur[0] = s_next[0];
            s_cur[1] = s_next[1];
        }
    };

    const int nsweep = CHUNKED ? 1 : iters, len = CHUNKED ? s1 - s0 : nst;
    for (int it = 0; it < nsweep; ++it) {
#ifdef F16W_PIN_QL
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int t = 0; t < KS; ++t) asm volatile("" : "+a"(ql[g][t]));
#endif
        for (int i = 0; i + 1 < len; ++i) block(std::true_type{});
        if (len > 0) block(std::false_type{});
        if (CHUNKED) break;
        // ---- end of a sweep: row update (mean_shift.py:70-77), one query group after the other
        bool low = false;
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {
            // (the two groups are independent; the order only steers hipcc 7.2's register allocator: with this one both
            // instantiations come out at 0 spilled registers / 0 bytes of scratch, with the other one 24 resp. 48 are spilled)
            const int g = PL ? 1 - gi : gi;
            const float rs = rsum[g] + xor32(rsum[g]);
            const float Dinv = UNSCALE_O / rs;
            // (one feature tile at a time: 16 values of the current Q live beside the accumulators, not 64 -- the kernel must not
            // spill; the additions into n2 keep the order of the 8-wave kernel: tile by tile, register by register)
            float n2 = 0.f;
#pragma unroll
            for (int c = 0; c < NT; ++c) {
                float qacc[16];
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float e0 = ((float)qh[g][2 * c + j][u] + (float)ql[g][2 * c + j][u]) * UNSCALE_Q;
                        const float e1 = ((float)qh[g][2 * c + j][4 + u] + (float)ql[g][2 * c + j][4 + u]) * UNSCALE_Q;
                        const float keep = hi ? e1 : e0, send = hi ? e0 : e1;
                        const float recv = __shfl_xor(send, 32, 64);
                        qacc[8 * j + u] = hi ? recv : keep;
                        qacc[8 * j + 4 + u] = hi ? keep : recv;
                    }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float q = qacc[r];
                    const float m = o[g][c][r] * Dinv - q;
                    const float nq = q + m;
                    o[g][c][r] = nq;
                    n2 += nq * nq;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            n2 += xor32(n2);
            const float nrm = sqrtf(n2);
            if (nrm < 0.5f) low = true;
            if (n == total) {
                // (row index and output address recomputed from a lane id the compiler cannot hoist: addresses formed at kernel
                // entry would live -- and spill -- across the whole launch)
                int l;
                asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
                const int qr = bx * 256 + wave * 64 + g * 32 + (l & 31), hl = l >> 5;
                if (qr < N) {
                    float* out = newX + ((size_t)cloud * N + qr) * D;
#pragma unroll
                    for (int c = 0; c < NT; ++c)
#pragma unroll
                        for (int q4 = 0; q4 < 4; ++q4) {
                            f32x4 v = {o[g][c][4 * q4] / nrm, o[g][c][4 * q4 + 1] / nrm, o[g][c][4 * q4 + 2] / nrm,
                                       o[g][c][4 * q4 + 3] / nrm};
                            *(f32x4*)(out + 32 * c + 8 * q4 + 4 * hl) = v;
                        }
                }
            } else {
#pragma unroll
                for (int c = 0; c < NT; ++c)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        float v[8];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const float a = (o[g][c][8 * j + u] / nrm) * SCALE_X, bq = (o[g][c][8 * j + 4 + u] / nrm) * SCALE_X;
                            const float keep = hi ? bq : a, send = hi ? a : bq;
                            const float recv = __shfl_xor(send, 32, 64);
                            v[u] = hi ? recv : keep;
                            v[4 + u] = hi ? keep : recv;
                        }
                        split_q(g, 2 * c + j, v);
                    }
#pragma unroll
                for (int c = 0; c < NT; ++c)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[g][c][r] = 0.f;
                rsum[g] = 0.f;
            }
            __builtin_amdgcn_sched_barrier(0);             // one query group after the other: nothing of the second is started early
        }
        if (!PL && lowq != nullptr && low) lowq[cloud] = 1;
        if (n != total) plain_first_product(lds + buf * STAGE);          // first block of the next sweep, new Q
    }
    if (CHUNKED) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const float rs = rsum[g] + xor32(rsum[g]);
            if (qrow[g] < N) {
                const size_t slot = ((size_t)cloud * N + qrow[g]) * nchunk + chunk;
                float* out = partO + slot * D;
#pragma unroll
                for (int c = 0; c < NT; ++c)
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        constexpr float U = 1.0f / 33554432.0f;          // 2^-25
                        f32x4 v = {o[g][c][4 * q4] * U, o[g][c][4 * q4 + 1] * U, o[g][c][4 * q4 + 2] * U, o[g][c][4 * q4 + 3] * U};
                        *(f32x4*)(out + 32 * c + 8 * q4 + 4 * hi) = v;
                    }
                if (hi == 0) partS[slot] = rs * (1.0f / 16384.0f);
            }
        }
        return;
    }
    // (iters == 0 never reaches this kernel: sed_ms_iterate_ws_f32 only plans the split-fp16 schedules for iters > 0)
}

#ifndef F16S_NBUF
#define F16S_NBUF 3                                    // stage buffers of the sparse kernel (4 fit -- 4 x 37 KiB + 7 KiB of tables <= 160 KiB -- and change nothing: 51.1 vs 50.5 ms)
#endif
#ifndef F16S_NBUF_RM_N
#define F16S_NBUF_RM_N 6
#endif
constexpr int F16S_NBUF_RM = F16S_NBUF_RM_N;       // stage buffers on row-major images (6 x 17 KiB)
constexpr int F16S_MAXW = 8;                      // 64-bit words of a stage mask: 512 stages = 16 384 points

constexpr int F16S_REFGROUP = 12;                 // reference images per LDS load: 12 x 9 KiB head planes <= 3 stage buffers
#ifndef F16S_DELTA_V
#define F16S_DELTA_V 0.005f
#endif
constexpr float F16S_DELTA = F16S_DELTA_V;              // masks stay valid while no query has turned by more than this (rad)

// ------------------------------------------------------------------------------------------------------------
// Block-sparse schedule, round 3 form (ms_iterate_d128_f16x_kernel): the same skipping RULE as ms_iterate_d128_f16s_kernel below
// -- every tile has two unit references with cos(alpha); a key tile is visited by a workgroup only if a query of the workgroup is
// within theta + alpha + margin of one of the tile's references -- on the pipeline of the dense kernel ms_iterate_f16w_kernel:
// 64 queries per wave, row-major 17 KiB stage images, software pipeline inside the wave, one barrier per listed stage. What
// changes against the dense kernel is only WHICH stages a workgroup copies and computes: the entries of its stage list (rebuilt
// when a query has moved), walked in alternating direction; what changes against the round-2 kernel: small workgroups (NW = 2
// waves = 128 queries, two per CU) whose waves ALL compute every listed stage (no per-wave skipping inside the list: a skipped
// block saved its MFMAs but left the pipeline in pieces -- 0.35 of the matrix roof against the dense pipeline's 0.56).
// Work queue of the persistent block-sparse kernels. sched (ints): [0 .. 2] heads of natural-order queues (counting launch; item_list == NULL),
// [8 .. 15] / [16 .. 23] heads of the per-XCD queues (iteration launch / its (h, l) redo), [24 .. 31] start and [32 .. 39] length of
// XCD x's queue inside item_list. A workgroup takes the next item of ITS XCD's queue -- whole clouds, heaviest first, a cloud's
// items longest first: the workgroups of an XCD work on one or two clouds at a time and their stage images stay in that XCD's
// L2 (with one global length-sorted list every XCD streamed every cloud: 3.4 TB/s of L2 misses, 20 x the dense kernel's) -- and
// when that queue is empty, the next item of the following XCDs' queues. An item = (cloud << 8) | block of query rows.
constexpr int MS_SCHED_INTS = 64;
__device__ __forceinline__ int ms_next_item(int* __restrict__ sched, const int* __restrict__ item_list, int head0, int nitems, int nbx) {
    if (item_list == nullptr) {
        const int j = atomicAdd(sched + head0, 1);          // natural order: head0 = the launch's own counter (0, 1, 2)
        return j >= nitems ? -1 : ((j / nbx) << 8) | (j % nbx);
    }
    const int xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7;          // HW_REG_XCC_ID[3:0]
    for (int s = 0; s < 8; ++s) {
        const int x = (xcc + s) & 7;
        const int len = sched[32 + x];
        if (len > 0 && __hip_atomic_load(sched + head0 + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < len) {
            const int j = atomicAdd(sched + head0 + x, 1);
            if (j < len) return item_list[sched[24 + x] + j];
        }
    }
    return -1;
}

constexpr int F16X_NBUF = 3;
template <int NW, bool PL = true>
__global__ __launch_bounds__(64 * NW, 1) void ms_iterate_d128_f16x_kernel(
    const float* __restrict__ X, const uint8_t* __restrict__ blob, float* __restrict__ newX,
    const float* __restrict__ bw, const int* __restrict__ flags, int N, int iters, float skip_below,
    const uint8_t* __restrict__ refblob, const float* __restrict__ tile_cosalpha, float margin,
    unsigned long long* __restrict__ stats, int* __restrict__ lowq, int nitems, const int* __restrict__ item_list,
    int* __restrict__ sched, int head0, int* __restrict__ item_stages) {
    constexpr int NT = 4;
    constexpr int MAXW = F16S_MAXW, QB = 64 * NW;         // query rows per workgroup
    constexpr int REFB = 9216;                            // the first 9 DMA pieces of an image hold its 8704-byte head plane
    __shared__ unsigned long long wmask[NW][MAXW];
    __shared__ int slist[512];
    __shared__ float wmoved[NW];
    __shared__ int ns_sh, item_sh;
    __shared__ __attribute__((aligned(16))) float thr[2 * 64 * MAXW];
    using L = StageLayoutD<NT>;
    constexpr int D = L::D, KS = 2 * NT, NSTEP = 4 * NT;  // k-steps of the first product, operand steps of a block
    constexpr int XROW = L::XROW, STAGE = L::STAGE, NPIECE = L::STAGE / 1024;
    constexpr int NBUF = 3;
    constexpr int RD = F16W_RING_DISTANCE;             // one step ahead = 6 / 4 MFMAs (192 / 128 matrix cycles) per operand pair
    constexpr int OFF_XH = L::OFF_XH, OFF_XL = L::OFF_XL;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];    // [3][STAGE]
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int li = lane & 31, hi = lane >> 5;
    // Persistent workgroups (the grid is the number of resident workgroups, not the number of work items): an item = 64 NW
    // query rows of one cloud for all iterations, and items take 6 .. 40 ms depending on how many stages their queries see --
    // in EVERY cloud (the queries inside its largest cluster list most of it). Left to the hardware dispatcher, which hands
    // workgroups out in order, slots sat idle for milliseconds behind a busy shader engine and the launch ended on a 35 ms
    // tail of long items (84 % of the slots busy on trained embeddings). Here a slot that has finished takes the next item of
    // `item_list` itself: items sorted by descending length of their first stage list (ms_sparse_item_order_kernel; counted by
    // a first launch of this kernel with item_stages != NULL, which stops after building the lists), so the launch ends on
    // its shortest items. item_list == NULL: items in natural order.
    const int nbx = (N + QB - 1) / QB;
    for (;;) {
    __syncthreads();                                      // every wave is done with the previous item (shared tables, item_sh)
    if (tid == 0) item_sh = ms_next_item(sched, item_list, head0, nitems, nbx);
    __syncthreads();
    const int item = __builtin_amdgcn_readfirstlane(item_sh);
    if (item < 0) break;
    const int bx = item & 0xff;
    const int cloud = item >> 8;
    [&]() __attribute__((always_inline)) {
#ifdef F16X_CLOCKS
    const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
#endif
    if (flags[cloud]) return;
    if (PL && lowq != nullptr && !lowq[cloud]) return;
    const float* Xc = X + (size_t)cloud * N * D;
    const int nst = (N + 31) >> 5;
    const int nrs = 2 * ((nst + 31) >> 5);               // reference images: image 2 k + w = w-th references of tiles 32 k ..
    const uint8_t* ref_c = refblob + (size_t)cloud * nrs * STAGE;
    const uint8_t* blob_c = blob + (size_t)cloud * nst * STAGE;
    constexpr int REFG = NBUF * STAGE / REFB;             // reference head planes per LDS load
    int qrow[2], qrow_c[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        qrow[g] = bx * QB + wave * 64 + g * 32 + li;
        qrow_c[g] = qrow[g] < N ? qrow[g] : N - 1;
    }

    const float b = bw[cloud];
    const float inv_b2_l2e = 1.44269504088896340736f / (b * b);
    const float K1 = inv_b2_l2e * (1.0f / 4194304.0f);
    const float K0 = LOG2_SCALE_P - inv_b2_l2e;
    const float TMIN = LOG2_SCALE_P - 75.0f * 1.44269504088896340736f;
    {   // thresholds: reference rho is "near" a query with  q . m_rho > cos(theta + alpha_rho + margin) - slack
        const float Dthr = -2.0f * skip_below * b * b;   // dist >= Dthr  <=>  weight <= e^skip
        const float theta = Dthr < 3.99f ? acosf(1.0f - 0.5f * Dthr) + margin + F16S_DELTA : 1.0e9f;
        for (int rho = tid; rho < 2 * 64 * MAXW; rho += 64 * NW) {
            float v = 3.0e38f;                           // references of tiles past the end: never near
            const int t = (rho >> 6) * 32 + (rho & 31);  // image rho / 32 = 2 (t / 32) + which reference
            if (t < nst) {
                const float ca = fminf(fmaxf(tile_cosalpha[(size_t)cloud * nrs * 32 + rho], -1.0f), 1.0f);
                const float ang = theta + acosf(ca);
                v = ang < 3.14f ? (cosf(ang) - 1.0e-3f) * (SCALE_X * SCALE_X) : -3.0e38f;        // -3e38: always near
            }
            thr[rho] = v;
        }
    }

    h16x8 qh[2][KS], ql[2][KS];
    auto split_q = [&](int g, int ks, const float* v) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const h16 h = (h16)v[i];
            qh[g][ks][i] = h;
            ql[g][ks][i] = (h16)(v[i] - (float)h);
        }
    };
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            float v[8];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const f32x4 t = *(const f32x4*)(Xc + (size_t)qrow_c[g] * D + 16 * ks + 8 * hi + 4 * q);
#pragma unroll
                for (int u = 0; u < 4; ++u) v[4 * q + u] = t[u] * SCALE_X;
            }
            split_q(g, ks, v);
        }

    // the 17 DMA pieces (1 KiB each) of a stage image are dealt round-robin to the waves
    static_assert(NPIECE == 17, "17 pieces");
    const unsigned lane16 = lane * 16;
    auto stage_dma = [&](int st, int buf) {
        const uint8_t* src = blob_c + (size_t)st * STAGE;
        uint8_t* dst = lds + buf * STAGE;
#pragma unroll
        for (int i = 0; i < (NPIECE + NW - 1) / NW; ++i) {
            const int pc = wave + i * NW;
            if (pc < NPIECE)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + pc * 1024 + lane16),
                                                 (__attribute__((address_space(3))) void*)(dst + pc * 1024), 16, 0, 0);
        }
    };
    h16x8 fa[4], fb[4];
    int xoff, toff;                                       // see ms_iterate_d128_f16r_kernel: same image rows, same permutation
    auto refresh_offsets = [&]() {
        int l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        const int m = l & 31, h = l >> 5;
        const int sig = 16 * (m >> 4) + 4 * (m & 3) + ((m >> 2) & 3);
        xoff = sig * XROW + h * 16;
        const int i16 = l & 15;
        toff = (4 * (i16 >> 2) + h) * XROW + 32 * ((l >> 4) & 1) + 8 * (i16 & 3);
    };
    refresh_offsets();
    typedef short v4s __attribute__((__vector_size__(4 * sizeof(short))));
    auto tr8 = [&](const uint8_t* plane, int c, int j) {
        const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) v4s*)(plane + toff + (16 * j) * XROW + 64 * c));
        const v4s hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) v4s*)(plane + toff + (16 * j + 2) * XROW + 64 * c));
        typedef short v8s __attribute__((__vector_size__(8 * sizeof(short))));
        const v8s both = __builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(h16x8, both);
    };
    auto ring_load = [&](int t, const uint8_t* base) {
        if (t < KS) {
            fa[t & 3] = *(const h16x8*)(base + OFF_XH + xoff + t * 32);
            fb[t & 3] = *(const h16x8*)(base + OFF_XL + xoff + t * 32);
        } else {
            const int c = (t - KS) >> 1, j = (t - KS) & 1;
            fa[t & 3] = tr8(base + OFF_XH, c, j);
            fb[t & 3] = tr8(base + OFF_XL, c, j);
        }
    };
    f32x16 s_cur[2], s_next[2];
    auto plain_first_product = [&](const uint8_t* base) {
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) s_cur[g][r] = 0.f;
#pragma unroll
        for (int t = 0; t < KS; ++t) {
            const h16x8 a = *(const h16x8*)(base + OFF_XH + xoff + t * 32);
            const h16x8 l = *(const h16x8*)(base + OFF_XL + xoff + t * 32);
#pragma unroll
            for (int g = 0; g < 2; ++g) s_cur[g] = mfma16(l, qh[g][t], s_cur[g]);
#pragma unroll
            for (int g = 0; g < 2; ++g) s_cur[g] = mfma16(a, ql[g][t], s_cur[g]);
#pragma unroll
            for (int g = 0; g < 2; ++g) s_cur[g] = mfma16(a, qh[g][t], s_cur[g]);
        }
    };

    f32x16 o[2][NT];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int c = 0; c < NT; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[g][c][r] = 0.f;
    float rsum[2] = {0.f, 0.f};
    int buf = 0;
    i32x4 phv[2][2], plv[2][2];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) { s_cur[g][r] = 0.f; s_next[g][r] = 0.f; }
    unsigned long long n_listed = 0, n_remake = 0;        // statistics only
    int ns = 0;                                           // entries of the workgroup's stage list
    bool fwd = true;                                      // the list is walked in alternating direction (L2 reuse)
    // list entries: one LDS word per block, fetched four blocks ahead
    auto entry_raw = [&](int j) { return slist[fwd ? j : ns - 1 - j]; };
    auto entry = [&](int j) { return __builtin_amdgcn_readfirstlane(entry_raw(j)); };
    int q0 = 0, q1 = 0, q2 = 0, q3 = 0;                   // entries n .. n + 3 of the running sweep (scalar registers)

    // One block of the pipeline. HAS_NEXT: the block is followed by another one of the same sweep (whose first product runs here,
    // under this block's weights). The blocks of a sweep form the INNER loop and the row update sits between sweeps, outside it:
    // with everything in one flat loop the register allocator weighed the row update like the hot path and spilled inside it.
    int n = 0;                                            // entries of this sweep done (the DMA runs three entries ahead)
    auto block = [&](auto has_next_c) __attribute__((always_inline)) {
        constexpr bool has_next = decltype(has_next_c)::value;
        const uint8_t* base = lds + buf * STAGE;
        const int nbuf = buf == NBUF - 1 ? 0 : buf + 1;
        const uint8_t* nbase = lds + nbuf * STAGE;
        const uint8_t* n2base = lds + (nbuf == NBUF - 1 ? 0 : nbuf + 1) * STAGE;
        const int key0 = q0 * 32;
        const bool tail = key0 + 32 > N;
        const int raw4 = n + 4 < ns ? entry_raw(n + 4) : 0;   // consumed at the end of the block
        refresh_offsets();

        auto weights2 = [&](int g, int t, auto tail_c) {
            float p[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int r = 2 * t + u;
                p[u] = __builtin_amdgcn_exp2f(fmaxf(fmaf(s_cur[g][r], K1, K0), TMIN));
                if (decltype(tail_c)::value && key0 + sigma_row(mfma_row(r, hi)) >= N) p[u] = 0.f;
                if (PL) rsum[g] += p[u];
            }
            const h16x2 h = {(h16)p[0], (h16)p[1]};
            phv[g][t >> 2][t & 3] = __builtin_bit_cast(int, h);
            if (PL) {
                const h16x2 l = {(h16)(p[0] - (float)h[0]), (h16)(p[1] - (float)h[1])};
                plv[g][t >> 2][t & 3] = __builtin_bit_cast(int, l);
            } else {
                rsum[g] += (float)h[0] + (float)h[1];
            }
        };

        // ---- phase 1: first product of block n + 1 (both query groups) with the weights of block n between its MFMAs
        auto phase1 = [&](auto tail_c) {
#pragma unroll
            for (int t = 0; t < KS; ++t) {
                if (t == 0) {
                    f32x16 z;
#pragma unroll
                    for (int r = 0; r < 16; ++r) z[r] = 0.f;
                    s_next[0] = mfma16(fb[0], qh[0][0], z);
                    s_next[1] = mfma16(fb[0], qh[1][0], z);
                } else {
                    s_next[0] = mfma16(fb[t & 3], qh[0][t], s_next[0]);
                    s_next[1] = mfma16(fb[t & 3], qh[1][t], s_next[1]);
                }
                if (t < 8) weights2(0, t, tail_c);              // 16 accumulator rows = 8 pairs
                s_next[0] = mfma16(fa[t & 3], ql[0][t], s_next[0]);
                s_next[1] = mfma16(fa[t & 3], ql[1][t], s_next[1]);
                if (t < 8) weights2(1, t, tail_c);
                s_next[0] = mfma16(fa[t & 3], qh[0][t], s_next[0]);
                s_next[1] = mfma16(fa[t & 3], qh[1][t], s_next[1]);
                if (t + RD < KS) ring_load(t + RD, nbase);
                else ring_load(t + RD, base);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if constexpr (has_next) {
            if (tail) phase1(std::true_type{});
            else phase1(std::false_type{});
        } else {
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int r = 0; r < 16; ++r) s_next[g][r] = 0.f;
#pragma unroll
            for (int t = 0; t < 8; ++t) {                          // (16 accumulator rows = 8 pairs, whatever the feature width)
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    if (tail) weights2(g, t, std::true_type{});
                    else weights2(g, t, std::false_type{});
                }
            }
#pragma unroll
            for (int t = 0; t < KS; ++t)
                if (t + RD >= KS) ring_load(t + RD, base);
            __builtin_amdgcn_sched_barrier(0);
        }

        // ---- phase 2: second product of block n, both query groups per operand read. (Every wave computes every listed stage.
        // Skipping the MFMAs of a wave whose weights all rounded to zero -- per 32-query group, per operand step or for the whole
        // phase -- was tried three ways and lost every time, 324 .. 390 ms against 260: hipcc 7.2 then moves the accumulators
        // between the two register files around the conditional MFMAs and spills inside the loop.)
#pragma unroll
        for (int t = KS; t < NSTEP; ++t) {
            const int c = (t - KS) >> 1, j = (t - KS) & 1;
#pragma unroll
            for (int g = 0; g < 2; ++g) o[g][c] = mfma16(fb[t & 3], __builtin_bit_cast(h16x8, phv[g][j]), o[g][c]);
            if (PL) {
#pragma unroll
                for (int g = 0; g < 2; ++g) o[g][c] = mfma16(fa[t & 3], __builtin_bit_cast(h16x8, plv[g][j]), o[g][c]);
            }
#pragma unroll
            for (int g = 0; g < 2; ++g) o[g][c] = mfma16(fa[t & 3], __builtin_bit_cast(h16x8, phv[g][j]), o[g][c]);
            if (t + RD < NSTEP) ring_load(t + RD, base);
            else if (n + 2 < ns) ring_load(t + RD - NSTEP, n2base);
            __builtin_amdgcn_sched_barrier(0);
            if (t == NSTEP - 1 - RD) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (n + 3 < ns) stage_dma(q3, buf);
            }
        }

        buf = nbuf;
        ++n;
        q0 = q1; q1 = q2; q2 = q3;
        q3 = __builtin_amdgcn_readfirstlane(raw4);
        if constexpr (has_next) {
            s_cur[0] = s_next[0];
            s_cur[1] = s_next[1];
        }
    };

    // Every workgroup walks the UNION of its waves' stage lists with the dense kernel's pipeline (all its waves compute every listed
    // stage: a stage one wave needs and the other does not costs that wave a block of MFMAs whose weights come out <= e^skip, but
    // the pipeline stays the dense kernel's -- 0.56 of the matrix roof against 0.35 for the per-wave skipping of the round-2 kernel;
    // with 128-query workgroups the union holds 56 % of the stages on trained embeddings where a 32-query wave needs 48 %).
    // Masks and list are reused while no query of the workgroup has turned by more than F16S_DELTA since they were made (the
    // thresholds carry that much extra slack). The rows at mask time are parked in the output rows.
    for (int it = 0; it < iters; ++it) {
        __syncthreads();                                 // every wave is out of the previous sweep's stage buffers
        bool remake = it == 0;
        if (it > 0) {
            float mx = 0.f;
#pragma unroll
            for (int w2 = 0; w2 < NW; ++w2) mx = fmaxf(mx, wmoved[w2]);
            remake = !(mx <= F16S_DELTA);
        }
        if (remake) {
#pragma unroll
            for (int g = 0; g < 2; ++g)
                if (qrow[g] < N) {                       // remember where the masks were made: the row's slot of the output
                    float* keep = newX + ((size_t)cloud * N + qrow[g]) * 128;
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks) {
                        f32x4 v0, v1;
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            v0[u] = ((float)qh[g][ks][u] + (float)ql[g][ks][u]) * UNSCALE_Q;
                            v1[u] = ((float)qh[g][ks][4 + u] + (float)ql[g][ks][4 + u]) * UNSCALE_Q;
                        }
                        *(f32x4*)(keep + 16 * ks + 8 * hi) = v0;          // Q operand of k-step ks: features 16 ks + 8 hi + i; the same
                        *(f32x4*)(keep + 16 * ks + 8 * hi + 4) = v1;      // lane reads them back in this order at the row update
                    }
                }
            // ---- this wave's 64 queries against all tile references -> its stage mask
            for (int g0 = 0; g0 < nrs; g0 += REFG) {
                const int ng = min(REFG, nrs - g0);
                if (g0 > 0) __syncthreads();                  // every wave is done with the previous group's planes
                for (int pc = wave; pc < ng * 9; pc += NW) {  // 1 KiB pieces: image pc / 9, piece pc % 9
                    const int im = pc / 9, piece = pc - 9 * im;
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void*)(ref_c + (size_t)(g0 + im) * STAGE + piece * 1024 + lane16),
                        (__attribute__((address_space(3))) void*)(lds + im * REFB + piece * 1024), 16, 0, 0);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                for (int im = 0; im < ng; ++im) {
                    const uint8_t* rbase = lds + im * REFB + OFF_XH + li * XROW + hi * 16;     // references in natural row order
                    unsigned word = 0;
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        f32x16 sr;
#pragma unroll
                        for (int r = 0; r < 16; ++r) sr[r] = 0.f;
#pragma unroll
                        for (int t = 0; t < 8; ++t) sr = mfma16(*(const h16x8*)(rbase + t * 32), qh[g][t], sr);
#pragma unroll
                        for (int q4 = 0; q4 < 4; ++q4) {
                            const f32x4 th = *(const f32x4*)(thr + (g0 + im) * 32 + 8 * q4 + 4 * hi);
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const unsigned long long bal = __builtin_amdgcn_ballot_w64(sr[4 * q4 + u] > th[u]);
                                word |= ((unsigned)bal != 0u ? 1u : 0u) << (8 * q4 + u);              // tile row of lane half 0
                                word |= ((unsigned)(bal >> 32) != 0u ? 1u : 0u) << (8 * q4 + u + 4);  // ... of lane half 1
                            }
                        }
                    }
                    if (lane == 0) {                                  // a tile is needed if either of its references is near
                        unsigned* wm = (unsigned*)wmask[wave] + ((g0 + im) >> 1);
                        *wm = ((g0 + im) & 1) ? (*wm | word) : word;
                    }
                }
            }
            if (lane == 0 && ((nrs >> 1) & 1)) ((unsigned*)wmask[wave])[nrs >> 1] = 0u;      // upper half of the last 64-bit word
            __syncthreads();
            // ---- the workgroup's stage list, ascending (wave 0: one 64-bit word of the union at a time)
            if (wave == 0) {
                int base = 0;
                for (int w2 = 0; w2 < (nst + 63) >> 6; ++w2) {
                    unsigned long long any = 0ull;
#pragma unroll
                    for (int v = 0; v < NW; ++v) any |= wmask[v][w2];
                    if ((any >> lane) & 1ull) slist[base + __builtin_popcountll(any & ((1ull << lane) - 1ull))] = 64 * w2 + lane;
                    base += __builtin_popcountll(any);
                }
                if (lane == 0) ns_sh = base;
            }
            __syncthreads();
            ns = __builtin_amdgcn_readfirstlane(ns_sh);
            ++n_remake;
            if (item_stages != nullptr) {                 // counting launch: the length of the first list is all that is wanted
                if (tid == 0) item_stages[cloud * nbx + bx] = ns;
                return;
            }
        }   // remake
        n_listed += ns;
        // ---- prime the copy pipeline of this sweep: entries 0 .. 2, first product of entry 0, operands of entry 1
        fwd = (it & 1) == 0;
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (j < ns) stage_dma(entry(j), j);
        q0 = ns > 0 ? entry(0) : 0;
        q1 = ns > 1 ? entry(1) : 0;
        q2 = ns > 2 ? entry(2) : 0;
        q3 = ns > 3 ? entry(3) : 0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        refresh_offsets();
        buf = 0;
        n = 0;
        if (ns > 0) plain_first_product(lds);
        if (ns > 1) {
#pragma unroll
            for (int t = 0; t < RD; ++t) ring_load(t, lds + STAGE);
        }
        for (int i = 0; i + 1 < ns; ++i) block(std::true_type{});
        if (ns > 0) block(std::false_type{});
        float wm = 0.f;
        // ---- end of a sweep: row update (mean_shift.py:70-77), one query group after the other
        bool low = false;
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {
            // (the two groups are independent; the order only steers hipcc 7.2's register allocator: with this one both
            // instantiations come out at 0 spilled registers / 0 bytes of scratch, with the other one 24 resp. 48 are spilled)
            const int g = PL ? 1 - gi : gi;
            const float rs = rsum[g] + xor32(rsum[g]);
            const float Dinv = UNSCALE_O / rs;
            // (one feature tile at a time: 16 values of the current Q live beside the accumulators, not 64 -- the kernel must not
            // spill; the additions into n2 keep the order of the 8-wave kernel: tile by tile, register by register)
            float n2 = 0.f;
#pragma unroll
            for (int c = 0; c < NT; ++c) {
                float qacc[16];
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float e0 = ((float)qh[g][2 * c + j][u] + (float)ql[g][2 * c + j][u]) * UNSCALE_Q;
                        const float e1 = ((float)qh[g][2 * c + j][4 + u] + (float)ql[g][2 * c + j][4 + u]) * UNSCALE_Q;
                        const float keep = hi ? e1 : e0, send = hi ? e0 : e1;
                        const float recv = __shfl_xor(send, 32, 64);
                        qacc[8 * j + u] = hi ? recv : keep;
                        qacc[8 * j + 4 + u] = hi ? keep : recv;
                    }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float q = qacc[r];
                    const float m = o[g][c][r] * Dinv - q;
                    const float nq = q + m;
                    o[g][c][r] = nq;
                    n2 += nq * nq;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            n2 += xor32(n2);
            const float nrm = sqrtf(n2);
            if (nrm < 0.5f) low = true;
            if (it == iters - 1) {
                // (row index and output address recomputed from a lane id the compiler cannot hoist: addresses formed at kernel
                // entry would live -- and spill -- across the whole launch)
                int l;
                asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
                const int qr = bx * QB + wave * 64 + g * 32 + (l & 31), hl = l >> 5;
                if (qr < N) {
                    float* out = newX + ((size_t)cloud * N + qr) * D;
#pragma unroll
                    for (int c = 0; c < NT; ++c)
#pragma unroll
                        for (int q4 = 0; q4 < 4; ++q4) {
                            f32x4 v = {o[g][c][4 * q4] / nrm, o[g][c][4 * q4 + 1] / nrm, o[g][c][4 * q4 + 2] / nrm,
                                       o[g][c][4 * q4 + 3] / nrm};
                            *(f32x4*)(out + 32 * c + 8 * q4 + 4 * hl) = v;
                        }
                }
            } else {
                // new Q operand + how far the new row is from where the masks were made (angle <= 1.06 chord for chords <= 0.6)
                float ch2 = 0.f;
                int l2;
                asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l2));
                const int qr2 = bx * QB + wave * 64 + g * 32 + (l2 & 31);
                const float* kept = newX + ((size_t)cloud * N + (qr2 < N ? qr2 : N - 1)) * 128 + 8 * (l2 >> 5);
#pragma unroll
                for (int c = 0; c < NT; ++c)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        float v[8];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const float a = (o[g][c][8 * j + u] / nrm) * SCALE_X, bq = (o[g][c][8 * j + 4 + u] / nrm) * SCALE_X;
                            const float keep = hi ? bq : a, send = hi ? a : bq;
                            const float recv = __shfl_xor(send, 32, 64);
                            v[u] = hi ? recv : keep;
                            v[4 + u] = hi ? keep : recv;
                        }
                        const f32x4 k0 = *(const f32x4*)(kept + 16 * (2 * c + j));
                        const f32x4 k1 = *(const f32x4*)(kept + 16 * (2 * c + j) + 4);
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const float d0 = v[u] * UNSCALE_Q - k0[u], d1 = v[4 + u] * UNSCALE_Q - k1[u];
                            ch2 = fmaf(d0, d0, fmaf(d1, d1, ch2));
                        }
                        split_q(g, 2 * c + j, v);
                    }
                if (qr2 >= N) ch2 = 0.f;
                ch2 += xor32(ch2);
                float w1 = ch2 <= 0.36f ? 1.06f * sqrtf(ch2) : 1.0e9f;          // NaN -> 1e9
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) w1 = fmaxf(w1, __shfl_xor(w1, off, 64));
                wm = fmaxf(wm, w1);
#pragma unroll
                for (int c = 0; c < NT; ++c)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[g][c][r] = 0.f;
                rsum[g] = 0.f;
            }
            __builtin_amdgcn_sched_barrier(0);             // one query group after the other: nothing of the second is started early
        }
        if (!PL && lowq != nullptr && low) lowq[cloud] = 1;
        if (it + 1 < iters && lane == 0) wmoved[wave] = wm;   // read after the barrier that opens the next sweep
    }
    if (stats && lane == 0) {
        // [0] stage visits of workgroups, [1] first and [2] second products of 32-query groups (this kernel: every wave computes
        // every listed stage), [3] the dense count of 32-query groups x stages x iterations, [4] mask / list constructions
        if (wave == 0) atomicAdd(stats + 0, n_listed);
        atomicAdd(stats + 1, 2ull * n_listed);
        atomicAdd(stats + 2, 2ull * n_listed);
        atomicAdd(stats + 3, 2ull * (unsigned long long)nst * (unsigned long long)iters);
        if (wave == 0) atomicAdd(stats + 4, n_remake);
#ifdef F16X_CLOCKS
        if (wave == 0) { atomicAdd(stats + 5, __builtin_amdgcn_s_memrealtime() - t_start); atomicMax(stats + 6, __builtin_amdgcn_s_memrealtime()); atomicMin(stats + 7, t_start);
            const int orig = cloud * nbx + bx;
            stats[8 + 4 * orig] = t_start; stats[9 + 4 * orig] = __builtin_amdgcn_s_memrealtime();
            stats[10 + 4 * orig] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)); stats[11 + 4 * orig] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11)); }
#endif
    }
    }();
    }   // work items
}


// ------------------------------------------------------------------------------------------------------------
// Block-sparse schedule on the pipelined split-fp16 kernel (round 2; its fp32 predecessor: tools/experiments/ms_sparse_fp32.hip).
// Rows arrive sorted so that 32-row tiles -- here: stage images -- are cluster-pure, together with two unit reference
// vectors per tile (normalised means of two groups of its rows) and cos(alpha) of each, alpha = the widest angle between
// the reference and a row of its group. Every iteration
//   (1) every wave measures its 32 current queries against ALL tile references: S = M Q^T on the matrix pipe (fp16 head
//       parts only: |error| <= 5e-4 in the dot product, covered by the threshold's slack), 8 MFMAs per 32 references;
//   (2) it marks the stages it needs: by the triangle inequality on the unit sphere angle(q, x) >= angle(q, m) - alpha for
//       every key x within alpha of a reference m, so a tile all of whose rows lie in caps with
//       q . m <= cos(theta + alpha + margin) - slack  for all 32 queries carries only weights <= e^skip for them (theta = the
//       angle at which the kernel weight drops to e^skip). A tile has TWO references, each covering a part of its rows: the
//       tile at the border between two clusters of the sorted order would otherwise be wide open and needed by everybody;
//   (3) the workgroup compacts the union of its 8 waves' marks into a stage list (thread s owns stage s: nst <= 512);
//   (4) the pipeline of ms_iterate_d128_f16p_kernel runs over that list only -- stages nobody needs are never copied to
//       LDS; a wave that does not need a listed stage only takes part in its barrier; a wave whose weights of a stage all
//       round to zero in fp16 (p 2^14 <= 2^-25: exactly the blocks whose O-contribution is 0 in the dense kernel too)
//       skips the second product.
// The references are staged like keys: ms_split_kernel lays M out as stage images (32 references per image), and (1)
// copies the head planes of up to 12 images at a time into the (then idle) stage buffers.
// The pipeline is primed and drained once per iteration (2 stage copies exposed); lists are walked in alternating
// direction so that an iteration starts on the stages the previous one left in L2.
// What is dropped relative to the dense kernel: weights <= e^skip in whole blocks, <= N e^skip of a row sum (>= 1).
// RM: row-major stage images (StageLayoutN: 17 KiB instead of the four-plane 37 KiB -- half the L2 / fabric traffic and DMA issue,
// six stage buffers instead of three), second-product operands by transpose reads like ms_iterate_f16w_kernel
// NW: waves per workgroup: 8 (256 query rows, one workgroup per CU) or 4 (128 rows, two per CU: smaller unions of the waves' stage
// lists, two independent barrier domains per CU, twice the stage copies)
// NT: 32-feature tiles of a row: 4 (d = 128) or 5 (d = 160: the HPNet-widened embedding; row-major images only)
template <bool STAGGER, bool PL = true, bool RM = false, int NW = 8, int NT = 4>     // PL = false: fp16 heads of the weights only (see ms_iterate_d128_f16q_kernel)
__global__ __launch_bounds__(64 * NW, 8 / NW) void ms_iterate_d128_f16s_kernel(
    const float* __restrict__ X, const uint8_t* __restrict__ blob, float* __restrict__ newX,
    const float* __restrict__ bw, const int* __restrict__ flags, int N, int iters, float skip_below,
    const uint8_t* __restrict__ refblob, const float* __restrict__ tile_cosalpha, float margin,
    unsigned long long* __restrict__ stats, int* __restrict__ lowq, int nitems, const int* __restrict__ item_list,
    int* __restrict__ sched, int head0, int* __restrict__ item_stages) {
    using L = StageLayout<32>;
    using LR = StageLayoutD<NT>;
    static_assert(RM || NT == 4, "four-plane images: d = 128 only");
    constexpr int D = 32 * NT, KS = 2 * NT, NSTEP = 4 * NT;      // feature width, k-steps of the first product, operand steps of a block
    constexpr int XROW = RM ? LR::XROW : L::XROW, TROW = L::TROW, STAGE = RM ? LR::STAGE : L::STAGE, NPIECE = STAGE / 1024;
    constexpr int OFF_XH = 0, OFF_XL = RM ? LR::OFF_XL : L::OFF_XL, OFF_TH = L::OFF_TH, OFF_TL = L::OFF_TL;
    static_assert(StageLayoutN::XROW == L::XROW && StageLayoutN::OFF_XL == L::OFF_XL, "the X planes of both d = 128 layouts coincide");
    constexpr int MAXW = F16S_MAXW;
    constexpr int NBUF = RM ? (NW == 8 ? F16S_NBUF_RM : (NT == 4 ? 4 : 3)) : F16S_NBUF;
    constexpr int REFP = (32 * XROW + 1023) / 1024, REFB = REFP * 1024;       // DMA pieces / bytes that cover an image's head plane
    constexpr int REFG = NBUF * STAGE / REFB < F16S_REFGROUP ? NBUF * STAGE / REFB : F16S_REFGROUP;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];    // [NBUF][STAGE]
    __shared__ unsigned long long wmask[NW][MAXW];
    __shared__ int slist[512];
    __shared__ int wcount[NW];
    __shared__ int item_sh;
    __shared__ float wmoved[NW];
    __shared__ __attribute__((aligned(16))) float thr[2 * 64 * MAXW]; // per reference: q . m (scaled 2^22) above which it is near
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int li = lane & 31, hi = lane >> 5;
    const bool late = STAGGER && wave >= NW / 2;
    // persistent workgroups over a sorted item list: see ms_iterate_d128_f16x_kernel (an item here = 256 query rows of a cloud)
    constexpr int QB = 32 * NW;                           // query rows per workgroup
    const int nbx = (N + QB - 1) / QB;
    for (;;) {
    __syncthreads();                                      // every wave is done with the previous item (shared tables, item_sh)
    if (tid == 0) item_sh = ms_next_item(sched, item_list, head0, nitems, nbx);
    __syncthreads();
    const int item = __builtin_amdgcn_readfirstlane(item_sh);
    if (item < 0) break;
    const int bx = item & 0xff;
    const int cloud = item >> 8;
    [&]() __attribute__((always_inline)) {
    if (flags[cloud]) return;
    if (PL && lowq != nullptr && !lowq[cloud]) return;     // second pass: only the clouds the heads-only pass has flagged
    const float* Xc = X + (size_t)cloud * N * D;
    const int nst = (N + 31) >> 5;
    const int nrs = 2 * ((nst + 31) >> 5);               // reference images: image 2 k + w = w-th references of tiles 32 k ..
    const uint8_t* ref_c = refblob + (size_t)cloud * nrs * STAGE;
    const uint8_t* blob_c = blob + (size_t)cloud * nst * STAGE;
    const int qrow = bx * QB + wave * 32 + li;
    const int qrow_c = qrow < N ? qrow : N - 1;

    const float b = bw[cloud];
    const float inv_b2_l2e = 1.44269504088896340736f / (b * b);
    const float K1 = inv_b2_l2e * (1.0f / 4194304.0f);
    const float K0 = LOG2_SCALE_P - inv_b2_l2e;
    const float TMIN = LOG2_SCALE_P - 75.0f * 1.44269504088896340736f;
    {   // thresholds: reference rho is "near" a query with  q . m_rho > cos(theta + alpha_rho + margin) - slack
        const float Dthr = -2.0f * skip_below * b * b;   // dist >= Dthr  <=>  weight <= e^skip
        const float theta = Dthr < 3.99f ? acosf(1.0f - 0.5f * Dthr) + margin + F16S_DELTA : 1.0e9f;
        for (int rho = tid; rho < 2 * 64 * MAXW; rho += 64 * NW) {
            float v = 3.0e38f;                           // references of tiles past the end: never near
            const int t = (rho >> 6) * 32 + (rho & 31);  // image rho / 32 = 2 (t / 32) + which reference
            if (t < nst) {
                const float ca = fminf(fmaxf(tile_cosalpha[(size_t)cloud * nrs * 32 + rho], -1.0f), 1.0f);
                const float ang = theta + acosf(ca);
                v = ang < 3.14f ? (cosf(ang) - 1.0e-3f) * (SCALE_X * SCALE_X) : -3.0e38f;        // -3e38: always near
            }
            thr[rho] = v;
        }
    }
    const float dead_below = 2.98023223876953125e-8f * 0.5f * __expf(-4.0f * F16S_DELTA / (b * b));
    h16x8 qh[KS], ql[KS];
    auto split_q = [&](int ks, const float* v) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const h16 h = (h16)v[i];
            qh[ks][i] = h;
            ql[ks][i] = (h16)(v[i] - (float)h);
        }
    };
    // Q operand of k-step ks on lane half hi: four-plane images: features 16 ks + 4 hi + {0..3, 8..11} (the order the accumulator
    // rows come in: the row update needs no exchange); row-major images: features 16 ks + 8 hi + 0..7 (the image's own order: the
    // row update exchanges four values per k-step with the other lane half, like ms_iterate_f16w_kernel)
#pragma unroll
    for (int c = 0; c < NT; ++c)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float v[8];
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const f32x4 t = *(const f32x4*)(Xc + (size_t)qrow_c * D + 32 * c + (RM ? 16 * j + 8 * hi + 4 * g : 8 * (2 * j + g) + 4 * hi));
#pragma unroll
                for (int u = 0; u < 4; ++u) v[4 * g + u] = t[u] * SCALE_X;
            }
            split_q(2 * c + j, v);
        }

    static_assert(RM || NPIECE == 37, "the four-plane piece distribution below is written for 37 pieces");
    const unsigned lane16 = lane * 16;
    auto stage_dma = [&](int st, int buf) {
        const uint8_t* src = blob_c + (size_t)st * STAGE;
        uint8_t* dst = lds + buf * STAGE;
        if constexpr (RM) {                               // 17 / 21 pieces dealt round-robin to the waves
#pragma unroll
            for (int i = 0; i < (NPIECE + NW - 1) / NW; ++i) {
                const int pc = wave + NW * i;
                if (pc < NPIECE)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + pc * 1024 + lane16),
                                                     (__attribute__((address_space(3))) void*)(dst + pc * 1024), 16, 0, 0);
            }
            return;
        }
        const auto g = (const __attribute__((address_space(1))) void*)(src + wave * 4096 + lane16);
        const auto l = (__attribute__((address_space(3))) void*)(dst + wave * 4096);
        __builtin_amdgcn_global_load_lds(g, l, 16, 0, 0);
        __builtin_amdgcn_global_load_lds(g, l, 16, 1024, 0);
        __builtin_amdgcn_global_load_lds(g, l, 16, 2048, 0);
        __builtin_amdgcn_global_load_lds(g, l, 16, 3072, 0);
        if (wave < 5)
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(src + (32 + wave) * 1024 + lane16),
                (__attribute__((address_space(3))) void*)(dst + (32 + wave) * 1024), 16, 0, 0);
    };

    h16x8 fa[4], fb[4];
    const int xoff_nat = li * XROW + hi * 16;             // natural row order (reference planes)
    // RM: key rows in sigma order for the first product, transpose reads for the second (ms_iterate_d128_f16r_kernel's scheme)
    const int xoff = RM ? (16 * (li >> 4) + 4 * (li & 3) + ((li >> 2) & 3)) * XROW + hi * 16 : xoff_nat;
    const int toff = RM ? (4 * ((lane & 15) >> 2) + hi) * XROW + 32 * ((lane >> 4) & 1) + 8 * (lane & 3) : li * TROW + hi * 16;
    typedef short v4s __attribute__((__vector_size__(4 * sizeof(short))));
    auto tr8 = [&](const uint8_t* plane, int c, int j) {
        const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) v4s*)(plane + toff + (16 * j) * XROW + 64 * c));
        const v4s hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) v4s*)(plane + toff + (16 * j + 2) * XROW + 64 * c));
        typedef short v8s __attribute__((__vector_size__(8 * sizeof(short))));
        const v8s both = __builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(h16x8, both);
    };
    auto ring_load = [&](int t, const uint8_t* base) {
        if (t < KS) {
            fa[t & 3] = *(const h16x8*)(base + OFF_XH + xoff + t * 32);
            fb[t & 3] = *(const h16x8*)(base + OFF_XL + xoff + t * 32);
        } else if constexpr (RM) {
            const int c = (t - KS) >> 1, j = (t - KS) & 1;
            fa[t & 3] = tr8(base + OFF_XH, c, j);
            fb[t & 3] = tr8(base + OFF_XL, c, j);
        } else {
            const int c = (t - KS) >> 1, j = (t - KS) & 1;
            fa[t & 3] = *(const h16x8*)(base + OFF_TH + toff + c * 32 * TROW + j * 32);
            fb[t & 3] = *(const h16x8*)(base + OFF_TL + toff + c * 32 * TROW + j * 32);
        }
    };

    f32x16 o[NT];
#pragma unroll
    for (int c = 0; c < NT; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[c][r] = 0.f;
    float rsum = 0.f;
    h16x8 ph[2], pl[2];
    unsigned long long n_listed = 0, n_first = 0, n_second = 0, n_remake = 0;      // per-wave counts (statistics only)

    // Masks and list are reused while no query of the workgroup has turned by more than F16S_DELTA since they were made (the
    // thresholds carry that much extra slack): mean-shift moves rows in its first few iterations and then barely at all.
    // The rows at mask time are parked in the output rows (row-private; overwritten by the result at the end).
    int ns = 0;
    for (int it = 0; it < iters; ++it) {
        __syncthreads();                                 // every wave is out of the previous iteration's stage buffers
        bool remake = it == 0;
        if (it > 0) {
            float mx = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) mx = fmaxf(mx, wmoved[w]);
            remake = !(mx <= F16S_DELTA);
        }
        if (remake) {
        if (qrow < N) {                                  // remember where the masks were made: the row's slot of the output
            float* keep = newX + ((size_t)cloud * N + qrow) * D;
#pragma unroll
            for (int c = 0; c < NT; ++c)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int r = 4 * g + u;
                        v[u] = ((float)qh[2 * c + (r >> 3)][r & 7] + (float)ql[2 * c + (r >> 3)][r & 7]) * UNSCALE_Q;
                    }
                    // (RM: Q-operand order -- element i of k-step 2 c + (g >> 1) -- read back by the same lane in the same order)
                    *(f32x4*)(keep + (RM ? 32 * c + 16 * (g >> 1) + 8 * hi + 4 * (g & 1) : 32 * c + 8 * g + 4 * hi)) = v;
                }
        }
        // ---- (1) + (2): this wave's queries against all tile references -> its stage mask
        for (int g0 = 0; g0 < nrs; g0 += REFG) {
            const int ng = min(REFG, nrs - g0);
            if (g0 > 0) __syncthreads();                      // every wave is done with the previous group's planes
            for (int pc = wave; pc < ng * REFP; pc += NW) {   // 1 KiB pieces: image pc / REFP, piece pc % REFP
                const int im = pc / REFP, piece = pc - REFP * im;
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(ref_c + (size_t)(g0 + im) * STAGE + piece * 1024 + lane16),
                    (__attribute__((address_space(3))) void*)(lds + im * REFB + piece * 1024), 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            for (int im = 0; im < ng; ++im) {
                const uint8_t* rbase = lds + im * REFB + OFF_XH + xoff_nat;
                f32x16 sr;
#pragma unroll
                for (int r = 0; r < 16; ++r) sr[r] = 0.f;
#pragma unroll
                for (int t = 0; t < KS; ++t) sr = mfma16(*(const h16x8*)(rbase + t * 32), qh[t], sr);
                unsigned word = 0;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 th = *(const f32x4*)(thr + (g0 + im) * 32 + 8 * g + 4 * hi);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const unsigned long long bal = __builtin_amdgcn_ballot_w64(sr[4 * g + u] > th[u]);
                        word |= ((unsigned)bal != 0u ? 1u : 0u) << (8 * g + u);              // tile row of lane half 0
                        word |= ((unsigned)(bal >> 32) != 0u ? 1u : 0u) << (8 * g + u + 4);  // ... of lane half 1
                    }
                }
                if (lane == 0) {                                  // a tile is needed if either of its references is near
                    unsigned* wm = (unsigned*)wmask[wave] + ((g0 + im) >> 1);
                    *wm = ((g0 + im) & 1) ? (*wm | word) : word;
                }
            }
        }
        if (lane == 0 && ((nrs >> 1) & 1)) ((unsigned*)wmask[wave])[nrs >> 1] = 0u;      // upper half of the last 64-bit word
        __syncthreads();
        ++n_remake;
        }   // remake
        // ---- (3) the workgroup's stage list, ascending: thread s owns stage s (s + 64 NW, .. in further passes). Made in EVERY
        // sweep (two barriers): between mask rebuilds the waves take dead stages out of their masks (below), and a stage no
        // wave needs any more leaves the list -- no copy, no barrier for it.
        ns = 0;
        for (int s0 = 0; s0 < nst; s0 += 64 * NW) {
            const int st = s0 + tid;
            bool need = false;
            if (st < nst) {
                const int w = st >> 6, sh = st & 63;
                unsigned long long any = 0ull;
#pragma unroll
                for (int v = 0; v < NW; ++v) any |= wmask[v][w];
                need = (any >> sh) & 1ull;
            }
            const unsigned long long bal = __builtin_amdgcn_ballot_w64(need);
            if (s0 > 0) __syncthreads();                  // wcount of the previous pass has been read
            if (lane == 0) wcount[wave] = __builtin_popcountll(bal);
            __syncthreads();
            int base = ns;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const int cnt = wcount[w];
                if (w < wave) base += cnt;
                ns += cnt;
            }
            if (need) slist[base + __builtin_popcountll(bal & ((1ull << lane) - 1ull))] = st;
        }
        __syncthreads();
        ns = __builtin_amdgcn_readfirstlane(ns);
        if (item_stages != nullptr) {                     // counting launch: the length of the first list is all that is wanted
            if (tid == 0) item_stages[cloud * nbx + bx] = ns;
            return;
        }
        n_listed += ns;

        // ---- (4) the pipeline over the list
        const bool fwd = (it & 1) == 0;
        auto entry = [&](int j) { return __builtin_amdgcn_readfirstlane(slist[fwd ? j : ns - 1 - j]); };
#pragma unroll
        for (int j0 = 0; j0 < NBUF - 1; ++j0)
            if (j0 < ns) stage_dma(entry(j0), j0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (ns > 0) {
#pragma unroll
            for (int t = 0; t < 3; ++t) ring_load(t, lds);
        }
        int buf = 0;
        for (int j = 0; j < ns; ++j) {
            const uint8_t* base = lds + buf * STAGE;
            const int nbuf = buf == NBUF - 1 ? 0 : buf + 1;
            const uint8_t* nbase = lds + nbuf * STAGE;
            const int st = entry(j);
            const int key0 = st * 32;
            const bool need =
                __builtin_amdgcn_readfirstlane((int)((wmask[wave][st >> 6] >> (st & 63)) & 1ull)) != 0;
            bool live = false;

            auto first_product_and_weights = [&]() {
                f32x16 s;
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
                for (int t = 0; t < KS; ++t) {
                    s = mfma16(fb[t & 3], qh[t], s);
                    s = mfma16(fa[t & 3], ql[t], s);
                    s = mfma16(fa[t & 3], qh[t], s);
                    ring_load(t + 3, base);
                    __builtin_amdgcn_sched_barrier(0);
                }
                float p[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) p[r] = __builtin_amdgcn_exp2f(fmaxf(fmaf(s[r], K1, K0), TMIN));
                if (key0 + 32 > N) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (key0 + (RM ? sigma_row(mfma_row(r, hi)) : mfma_row(r, hi)) >= N) p[r] = 0.f;
                }
                float pmax = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    pmax = fmaxf(pmax, p[r]);
                    const h16 h = (h16)p[r];
                    ph[r >> 3][r & 7] = h;
                    if (PL) {
                        rsum += p[r];
                        pl[r >> 3][r & 7] = (h16)(p[r] - (float)h);
                    } else {
                        rsum += (float)h;
                    }
                }
                // p 2^14 <= 2^-25 rounds to (h, l) = (0, 0): the second product of such a block adds exactly nothing
                live = __builtin_amdgcn_ballot_w64(pmax > 2.98023223876953125e-8f) != 0ull;
                // ... and a block whose largest weight is below 2^-25 e^(-4 delta / b^2) stays that way until the masks are remade:
                // every query is within delta of where the masks were made, hence within 2 delta of where it is now; a chord
                // (<= 2) then changes by <= 2 delta and the exponent -chord^2 / 2 b^2 by <= 4 delta / b^2. The wave drops the stage
                // from its OWN mask: no first product for it in the sweeps that follow (exactly the blocks whose second product
                // would be skipped anyway).
                if (__builtin_amdgcn_ballot_w64(pmax > dead_below) == 0ull && lane == 0)
                    wmask[wave][st >> 6] &= ~(1ull << (st & 63));
                ++n_first;
            };

            if (!late && need) first_product_and_weights();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                                  // B_j
            // entry j + NBUF - 1 goes into the buffer entry j - 1 has left (every wave is past it: it passed B_j)
            if (j + NBUF - 1 < ns) stage_dma(entry(j + NBUF - 1), buf == 0 ? NBUF - 1 : buf - 1);
            if (late && need) first_product_and_weights();

            if (live) {
                ++n_second;
#pragma unroll
                for (int t = KS; t < NSTEP; ++t) {
                    const int c = (t - KS) >> 1, jj = (t - KS) & 1;
                    o[c] = mfma16(fb[t & 3], ph[jj], o[c]);
                    if (PL) o[c] = mfma16(fa[t & 3], pl[jj], o[c]);
                    o[c] = mfma16(fa[t & 3], ph[jj], o[c]);
                    if (t + 3 < NSTEP) ring_load(t + 3, base);
                    else if (j + 1 < ns) ring_load(t + 3 - NSTEP, nbase);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else if (j + 1 < ns) {
#pragma unroll
                for (int t = 0; t < 3; ++t) ring_load(t, nbase);
            }
            buf = nbuf;
        }

        // ---- row update (mean_shift.py:70-77)
        const float rs = rsum + xor32(rsum);
        const float Dinv = UNSCALE_O / rs;
        float n2 = 0.f;
#pragma unroll
        for (int c = 0; c < NT; ++c) {
            float qacc[16];                               // the current row in accumulator order
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float e0 = ((float)qh[2 * c + j][u] + (float)ql[2 * c + j][u]) * UNSCALE_Q;
                    const float e1 = ((float)qh[2 * c + j][4 + u] + (float)ql[2 * c + j][4 + u]) * UNSCALE_Q;
                    if (RM) {
                        const float keep_ = hi ? e1 : e0, send = hi ? e0 : e1;
                        const float recv = __shfl_xor(send, 32, 64);
                        qacc[8 * j + u] = hi ? recv : keep_;
                        qacc[8 * j + 4 + u] = hi ? keep_ : recv;
                    } else {
                        qacc[8 * j + u] = e0;
                        qacc[8 * j + 4 + u] = e1;
                    }
                }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float q = qacc[r];
                const float m = o[c][r] * Dinv - q;
                const float nq = q + m;
                o[c][r] = nq;
                n2 += nq * nq;
            }
        }
        n2 += xor32(n2);
        const float nrm = sqrtf(n2);
        if (!PL && lowq != nullptr && nrm < 0.5f) lowq[cloud] = 1;       // see ms_iterate_d128_f16q_kernel
        if (it + 1 < iters) {   // how far is the new row from where the masks were made (angle <= 1.06 chord for chords <= 0.6)
            float ch2 = 0.f;
            if (RM) {                                     // new Q operand (exchange with the other lane half) and its distance to the parked row
                const float* kept = newX + ((size_t)cloud * N + qrow_c) * D + 8 * hi;
#pragma unroll
                for (int c = 0; c < NT; ++c)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        float v[8];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const float a_ = (o[c][8 * j + u] / nrm) * SCALE_X, b_ = (o[c][8 * j + 4 + u] / nrm) * SCALE_X;
                            const float keep_ = hi ? b_ : a_, send = hi ? a_ : b_;
                            const float recv = __shfl_xor(send, 32, 64);
                            v[u] = hi ? recv : keep_;
                            v[4 + u] = hi ? keep_ : recv;
                        }
                        const f32x4 k0 = *(const f32x4*)(kept + 16 * (2 * c + j));
                        const f32x4 k1 = *(const f32x4*)(kept + 16 * (2 * c + j) + 4);
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const float d0 = v[u] * UNSCALE_Q - k0[u], d1 = v[4 + u] * UNSCALE_Q - k1[u];
                            ch2 = fmaf(d0, d0, fmaf(d1, d1, ch2));
                        }
                        split_q(2 * c + j, v);
                    }
                if (qrow >= N) ch2 = 0.f;
            } else if (qrow < N) {
                const float* keep = newX + ((size_t)cloud * N + qrow) * D;
                const float inv = 1.0f / nrm;
#pragma unroll
                for (int c = 0; c < NT; ++c)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 k = *(const f32x4*)(keep + 32 * c + 8 * g + 4 * hi);
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const float dlt = o[c][4 * g + u] * inv - k[u];
                            ch2 = fmaf(dlt, dlt, ch2);
                        }
                    }
            }
            ch2 += xor32(ch2);
            float wm = ch2 <= 0.36f ? 1.06f * sqrtf(ch2) : 1.0e9f;          // NaN -> 1e9
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) wm = fmaxf(wm, __shfl_xor(wm, off, 64));
            if (lane == 0) wmoved[wave] = wm;            // read after the barrier that opens the next iteration
        }
        if (it == iters - 1) {
            if (qrow < N) {
                float* out = newX + ((size_t)cloud * N + qrow) * D;
#pragma unroll
                for (int c = 0; c < NT; ++c)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f32x4 v = {o[c][4 * g] / nrm, o[c][4 * g + 1] / nrm, o[c][4 * g + 2] / nrm,
                                   o[c][4 * g + 3] / nrm};
                        *(f32x4*)(out + 32 * c + 8 * g + 4 * hi) = v;
                    }
            }
        } else {
            if (!RM) {
#pragma unroll
                for (int c = 0; c < NT; ++c)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        float v[8];
#pragma unroll
                        for (int i = 0; i < 8; ++i) v[i] = (o[c][8 * j + i] / nrm) * SCALE_X;
                        split_q(2 * c + j, v);
                    }
            }
#pragma unroll
            for (int c = 0; c < NT; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[c][r] = 0.f;
            rsum = 0.f;
        }
    }
    if (stats && lane == 0) {
        // [0] stage visits of workgroups (listed), [1] first products of waves, [2] second products of waves,
        // [3] dense count: waves x stages x iterations, [4] mask / list constructions of workgroups
        if (wave == 0) atomicAdd(stats + 0, n_listed);
        atomicAdd(stats + 1, n_first);
        atomicAdd(stats + 2, n_second);
        atomicAdd(stats + 3, (unsigned long long)nst * (unsigned long long)iters);
        if (wave == 0) atomicAdd(stats + 4, n_remake);
    }
    }();
    }   // work items
}

}  // namespace

// ---- entry points used by ms_iterate.hip's planner ----------------------------------------------------------
// `digits` = fp16 digits of the kernel weights in the second product (sed_ms_options_t.weight_digits): 1 = fp16 heads, clouds whose
// weighted means cancel redone with (h, l) weights by a second launch; 2 = (h, l) weights everywhere. No state is kept between
// calls (the function-local `attr` flags only remember that a kernel's dynamic-LDS limit has been raised once).

static size_t f16_flag_bytes(int B) { return (((size_t)B * sizeof(int) + 255) / 256) * 256; }
static size_t f16_blob_bytes_n(int B, int N, int d) {                                      // row-major images
    return (size_t)B * ((N + 31) / 32) * (d == 160 ? StageLayoutD<5>::STAGE : StageLayoutD<4>::STAGE);
}
static size_t f16_blob_bytes_4(int B, int N) { return (size_t)B * ((N + 31) / 32) * StageLayout<32>::STAGE; }    // four-plane images

// stage images | "rows not unit" flags | "weighted means cancel" flags (both per cloud, 256-byte blocks)
size_t ms_f16_workspace_bytes(int B, int N, int d) { return f16_blob_bytes_n(B, N, d) + 2 * f16_flag_bytes(B); }

template <int NT>
static int f16w_attr() {
    static bool attr = false;
    if (attr) return SED_OK;
    constexpr int sm = 3 * StageLayoutD<NT>::STAGE;
    hipError_t e = hipFuncSetAttribute((const void*)ms_iterate_f16w_kernel<NT, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, sm);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void*)ms_iterate_f16w_kernel<NT, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, sm);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void*)ms_iterate_f16w_kernel<NT, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, sm);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void*)ms_iterate_f16w_kernel<NT, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, sm);
    if (e != hipSuccess) return (int)e;
    attr = true;
    return SED_OK;
}

static int f16r_attr() {
    static bool attr = false;
    if (attr) return SED_OK;
    constexpr int sm = 3 * StageLayoutN::STAGE;
    hipError_t e = hipFuncSetAttribute((const void*)ms_iterate_d128_f16r_kernel<false, false>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, sm);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void*)ms_iterate_d128_f16r_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, sm);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void*)ms_iterate_d128_f16r_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, sm);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void*)ms_iterate_d128_f16r_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, sm);
    if (e != hipSuccess) return (int)e;
    attr = true;
    return SED_OK;
}

// the 64-queries-per-wave kernel (any of its four forms) for feature width 32 NT
template <int NT>
static void f16w_run(bool chunked, bool pl, dim3 grid, const float* X, const uint8_t* blob, float* newX, const float* bw,
                     const int* flags, int N, int iters, const float* Q, float* partO, float* partS, int* lowq, hipStream_t stream) {
    constexpr int sm = 3 * StageLayoutD<NT>::STAGE;
    if (chunked && pl) ms_iterate_f16w_kernel<NT, true, true><<<grid, 256, sm, stream>>>(X, blob, newX, bw, flags, N, iters, Q, partO, partS, lowq);
    else if (chunked) ms_iterate_f16w_kernel<NT, true, false><<<grid, 256, sm, stream>>>(X, blob, newX, bw, flags, N, iters, Q, partO, partS, lowq);
    else if (pl) ms_iterate_f16w_kernel<NT, false, true><<<grid, 256, sm, stream>>>(X, blob, newX, bw, flags, N, iters, Q, partO, partS, lowq);
    else ms_iterate_f16w_kernel<NT, false, false><<<grid, 256, sm, stream>>>(X, blob, newX, bw, flags, N, iters, Q, partO, partS, lowq);
}
static void f16w_any(int d, bool chunked, bool pl, dim3 grid, const float* X, const uint8_t* blob, float* newX, const float* bw,
                     const int* flags, int N, int iters, const float* Q, float* partO, float* partS, int* lowq, hipStream_t stream) {
    if (d == 160) f16w_run<5>(chunked, pl, grid, X, blob, newX, bw, flags, N, iters, Q, partO, partS, lowq, stream);
    else f16w_run<4>(chunked, pl, grid, X, blob, newX, bw, flags, N, iters, Q, partO, partS, lowq, stream);
}

// one launch, all iterations; flags live behind the stage images; *flags_out = the per-cloud "rows not unit" flags the exact
// fp32 kernel reads. d = 128, or 160 (the HPNet-widened embedding; 64-queries-per-wave kernel only)
int ms_f16_launch(int B, int N, int d, int iters, const float* bw, const float* X, float* newX, void* workspace, int** flags_out,
                  int digits, int wq, hipStream_t stream) {
    using L = StageLayoutN;
    uint8_t* blob = (uint8_t*)workspace;
    int* flags = (int*)(blob + f16_blob_bytes_n(B, N, d));
    int* lowq = (int*)((uint8_t*)flags + f16_flag_bytes(B));
    *flags_out = flags;
    hipError_t e = hipMemsetAsync(flags, 0, 2 * f16_flag_bytes(B), stream);
    if (e != hipSuccess) return (int)e;
    int rc = f16r_attr();
    if (rc == SED_OK) rc = f16w_attr<4>();
    if (rc == SED_OK) rc = f16w_attr<5>();
    if (rc != SED_OK) return rc;
    const int nst = (N + 31) / 32;
    const dim3 grid((N + 255) / 256, B);
    if (d == 160) ms_split_d_kernel<5><<<dim3(nst, B), 256, 0, stream>>>(X, bw, blob, flags, N, nst);
    else ms_split_n_kernel<<<dim3(nst, B), 256, 0, stream>>>(X, bw, blob, flags, N, nst);
    if (wq == 64 || d == 160) {  // 64 queries per wave: 4-wave workgroups, one wave per SIMD (same bits as the 8-wave kernel at d = 128)
        if (digits == 2) {
            f16w_any(d, false, true, grid, X, blob, newX, bw, flags, N, iters, nullptr, nullptr, nullptr, nullptr, stream);
        } else {
            f16w_any(d, false, false, grid, X, blob, newX, bw, flags, N, iters, nullptr, nullptr, nullptr, lowq, stream);
            f16w_any(d, false, true, grid, X, blob, newX, bw, flags, N, iters, nullptr, nullptr, nullptr, lowq, stream);
        }
    } else if (digits == 2) {
        ms_iterate_d128_f16r_kernel<false, true><<<grid, 512, 3 * L::STAGE, stream>>>(X, blob, newX, bw, flags, N, iters);
    } else {
        ms_iterate_d128_f16r_kernel<false, false><<<grid, 512, 3 * L::STAGE, stream>>>(X, blob, newX, bw, flags, N, iters,
                                                                                       nullptr, nullptr, nullptr, lowq);
        ms_iterate_d128_f16r_kernel<false, true><<<grid, 512, 3 * L::STAGE, stream>>>(X, blob, newX, bw, flags, N, iters,
                                                                                      nullptr, nullptr, nullptr, lowq);
    }
    SED_LAUNCH_CHECK();
    return SED_OK;
}

// key-chunked split-fp16 schedule: chunk count from N only (results do not depend on how many clouds share a launch):
// as many chunks as fill the 256 CUs with ONE cloud's workgroups, at least 8 stages per chunk; 0 = not worth it
int ms_f16_chunks(int N) {
    const int nbx = (N + 255) / 256, nst = (N + 31) / 32;
    if (N < 2560) return 0;
    int S = 256 / nbx;
    if (S > nst / 8) S = nst / 8;
    return S < 2 ? 0 : S;
}

static size_t f16_chunked_base_bytes(int B, int N, int d) { return f16_blob_bytes_n(B, N, d) + 2 * f16_flag_bytes(B); }

size_t ms_f16_chunked_workspace_bytes(int B, int N, int d) {
    const int S = ms_f16_chunks(N) > 1 ? ms_f16_chunks(N) : 1;
    return f16_chunked_base_bytes(B, N, d) + (size_t)B * N * S * (d + 1) * sizeof(float) + 256;
}

// one launch pair per iteration; `combine` = ms_iterate.hip's ms_combine_kernel launcher. S = key chunks per query block
// (ms_f16_chunks(N) when few clouds would leave CUs idle; 1 = whole sweeps, the form the d = 160 kernel always takes: its
// one-launch instantiation does not fit the register file)
int ms_f16_chunked_launch(int B, int N, int d, int S, int iters, const float* bw, const float* X, float* newX, void* workspace,
                          int** flags_out, int (*combine)(const float*, const float*, const float*, float*, size_t, int,
                                                          int, int, int*, hipStream_t),
                          int digits, int wq, hipStream_t stream) {
    using L = StageLayoutN;
    const int nst = (N + 31) / 32;
    uint8_t* blob = (uint8_t*)workspace;
    int* flags = (int*)(blob + f16_blob_bytes_n(B, N, d));
    float* partO = (float*)(((uintptr_t)((uint8_t*)workspace + f16_chunked_base_bytes(B, N, d)) + 255) & ~(uintptr_t)255);
    float* partS = partO + (size_t)B * N * S * d;
    int* lowq = (int*)((uint8_t*)flags + f16_flag_bytes(B));
    *flags_out = flags;
    hipError_t e = hipMemsetAsync(flags, 0, 2 * f16_flag_bytes(B), stream);
    if (e != hipSuccess) return (int)e;
    int rca = f16r_attr();
    if (rca == SED_OK) rca = f16w_attr<4>();
    if (rca == SED_OK) rca = f16w_attr<5>();
    if (rca != SED_OK) return rca;
    const bool heads = digits != 2;
    const bool wide = wq == 64 || d == 160;
    const dim3 grid((N + 255) / 256, B, S);
    if (d == 160) ms_split_d_kernel<5><<<dim3(nst, B), 256, 0, stream>>>(X, bw, blob, flags, N, nst);
    else ms_split_n_kernel<<<dim3(nst, B), 256, 0, stream>>>(X, bw, blob, flags, N, nst);
    for (int it = 0; it < iters; ++it) {
        const float* Q = it == 0 ? X : newX;
        if (wide)
            f16w_any(d, true, !heads, grid, X, blob, newX, bw, flags, N, 1, Q, partO, partS, nullptr, stream);
        else if (heads)
            ms_iterate_d128_f16r_kernel<true, false><<<grid, 512, 3 * L::STAGE, stream>>>(X, blob, newX, bw, flags, N, 1, Q, partO,
                                                                                         partS);
        else
            ms_iterate_d128_f16r_kernel<true, true><<<grid, 512, 3 * L::STAGE, stream>>>(X, blob, newX, bw, flags, N, 1, Q, partO,
                                                                                        partS);
        // the combine kernel sees the norm of every weighted mean: with heads-only weights it flags clouds whose means cancel
        const int rc = combine(partO, partS, Q, newX, (size_t)B * N, S, d, N, heads ? lowq : nullptr, stream);
        if (rc != SED_OK) return rc;
    }
    if (heads && iters > 0) {                         // flagged clouds again, (h, l) weights, all iterations in one launch
        if (wide)
            f16w_any(d, false, true, dim3((N + 255) / 256, B), X, blob, newX, bw, flags, N, iters, nullptr, nullptr, nullptr, lowq,
                     stream);
        else
            ms_iterate_d128_f16r_kernel<false, true><<<dim3((N + 255) / 256, B), 512, 3 * L::STAGE, stream>>>(
                X, blob, newX, bw, flags, N, iters, nullptr, nullptr, nullptr, lowq);
    }
    SED_LAUNCH_CHECK();
    return SED_OK;
}

// stage images of the sorted rows | flags | stage images of the tile references | scratch flags | cancel flags | work queues
size_t ms_f16_sparse_workspace_bytes(int B, int N) {
    const int nref = 2 * ((((N + 31) / 32) + 31) / 32) * 32;            // reference rows
    return f16_blob_bytes_4(B, N) + f16_blob_bytes_4(B, nref) + 3 * f16_flag_bytes(B) +
           (size_t)(MS_SCHED_INTS + 2 * (size_t)B * ((N + 127) / 128)) * sizeof(int);      // + queues, first list lengths, item list
}

// The per-XCD item queues of the persistent block-sparse kernels (layout: ms_next_item) from the first stage-list length of
// every item: clouds ranked by their total length and dealt to the 8 XCDs in snake order (equal shares of the work), every XCD's
// queue = its clouds one after the other, heaviest first, each cloud's items longest first (the launch ends on short items; what
// is left over at the end is taken by the XCDs that finish early). One workgroup; B <= MS_ORDER_MAX_CLOUDS.
constexpr int MS_ORDER_MAX_CLOUDS = 4096;
__global__ __launch_bounds__(1024) void ms_sparse_item_order_kernel(const int* __restrict__ item_stages, int B, int nbx,
                                                                    int* __restrict__ item_list, int* __restrict__ sched) {
    __shared__ int total[MS_ORDER_MAX_CLOUDS];
    __shared__ unsigned short rank_of[MS_ORDER_MAX_CLOUDS];
    __shared__ int nclouds[8], qstart[8];
    const int tid = threadIdx.x;
    for (int c = tid; c < B; c += 1024) {
        int t = 0;
        for (int b = 0; b < nbx; ++b) t += item_stages[c * nbx + b];
        total[c] = t;
    }
    __syncthreads();
    for (int c = tid; c < B; c += 1024) {                  // rank by (total descending, cloud ascending)
        const int t = total[c];
        int r = 0;
        for (int o = 0; o < B; ++o) r += (total[o] > t || (total[o] == t && o < c)) ? 1 : 0;
        rank_of[c] = (unsigned short)r;
    }
    if (tid < 8) {                                         // ranks 8 m + j go to XCD j (m even) or 7 - j (m odd)
        int n = 0;
        for (int r = 0; r < B; ++r) n += (((r >> 3) & 1) ? 7 - (r & 7) : (r & 7)) == tid ? 1 : 0;
        nclouds[tid] = n;
    }
    __syncthreads();
    if (tid == 0) {
        int acc = 0;
        for (int x = 0; x < 8; ++x) { qstart[x] = acc; acc += nclouds[x] * nbx; }
    }
    __syncthreads();
    if (tid < 8) { sched[24 + tid] = qstart[tid]; sched[32 + tid] = nclouds[tid] * nbx; }
    for (int i = tid; i < B * nbx; i += 1024) {
        const int c = i / nbx, b = i - c * nbx;
        const int r = rank_of[c], x = ((r >> 3) & 1) ? 7 - (r & 7) : (r & 7);
        const int ns = item_stages[i];
        int pos = 0;                                       // position among the cloud's items: (length descending, block ascending)
        for (int o = 0; o < nbx; ++o) {
            const int os = item_stages[c * nbx + o];
            pos += (os > ns || (os == ns && o < b)) ? 1 : 0;
        }
        item_list[qstart[x] + (r >> 3) * nbx + pos] = (c << 8) | b;
    }
}

// Block-sparse split-fp16 schedule on rows sorted into cluster-pure tiles. nref = 64 ceil(ceil(N / 32) / 32) reference rows:
// row (2 (t / 32) + w) 32 + t % 32 = w-th reference of tile t; tile_ref [B, nref, 128] unit vectors (unused rows zero),
// tile_cosalpha [B, nref] = smallest dot product of a row of the reference's group with it.
// workspace = ms_f16_sparse_workspace_bytes(B, N); stats (optional, device, 5 x u64, accumulated; the redo pass is not counted).
template <int NW>
static int f16x_launch(int B, int N, int iters, const float* bw, const float* X, float* newX, uint8_t* blob, int* flags,
                       uint8_t* refblob, int* flags2, int* lowq, float skip_below, const float* tile_ref, const float* tile_cosalpha,
                       float margin, unsigned long long* stats, int digits, int* sched, hipStream_t stream) {
    using L = StageLayoutN;
    const int nst = (N + 31) / 32, nrs = 2 * ((nst + 31) / 32);
    constexpr int sm = F16X_NBUF * L::STAGE;
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)ms_iterate_d128_f16x_kernel<NW, true>, hipFuncAttributeMaxDynamicSharedMemorySize, sm);
        if (e != hipSuccess) return (int)e;
        e = hipFuncSetAttribute((const void*)ms_iterate_d128_f16x_kernel<NW, false>, hipFuncAttributeMaxDynamicSharedMemorySize, sm);
        if (e != hipSuccess) return (int)e;
        attr = true;
    }
    const int nbx = (N + 64 * NW - 1) / (64 * NW);
    if (nbx > 255 || B > (1 << 22)) return SED_EUNSUPPORTED;
    static int slots = 0;                                  // resident workgroups: 512 registers per wave = one wave per SIMD
    if (!slots) {
        int dev = 0, cus = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (e != hipSuccess) return (int)e;
        slots = cus * (4 / NW);
    }
    const int nitems = nbx * B;
    const dim3 grid((unsigned)(nitems < slots ? nitems : slots));
    int* item_stages = sched + MS_SCHED_INTS;              // [MS_SCHED_INTS] queues (ms_next_item) | [nitems] first list lengths | [nitems] item list
    int* item_list = item_stages + nitems;
    const int* listed = B <= MS_ORDER_MAX_CLOUDS ? item_list : nullptr;      // (beyond: items in natural order)
    hipError_t e = hipMemsetAsync(sched, 0, (size_t)(MS_SCHED_INTS + nitems) * sizeof(int), stream);
    if (e != hipSuccess) return (int)e;
    ms_split_n_kernel<<<dim3(nst, B), 256, 0, stream>>>(X, bw, blob, flags, N, nst);
    ms_split_n_kernel<<<dim3(nrs, B), 256, 0, stream>>>(tile_ref, bw, refblob, flags2, nrs * 32, nrs);
    // first launch: every item builds its first stage list and reports its length; then the items are sorted by it
    ms_iterate_d128_f16x_kernel<NW, true><<<grid, 64 * NW, sm, stream>>>(X, blob, newX, bw, flags, N, iters, skip_below, refblob,
                                                                         tile_cosalpha, margin, nullptr, nullptr, nitems, nullptr,
                                                                         sched, 0, item_stages);
    if (listed) ms_sparse_item_order_kernel<<<1, 1024, 0, stream>>>(item_stages, B, nbx, item_list, sched);
    if (digits != 2) {        // heads-only weights; flagged clouds again with (h, l) weights
        ms_iterate_d128_f16x_kernel<NW, false><<<grid, 64 * NW, sm, stream>>>(X, blob, newX, bw, flags, N, iters, skip_below, refblob,
                                                                              tile_cosalpha, margin, stats, lowq, nitems, listed,
                                                                              sched, listed ? 8 : 1, nullptr);
        ms_iterate_d128_f16x_kernel<NW, true><<<grid, 64 * NW, sm, stream>>>(X, blob, newX, bw, flags, N, iters, skip_below, refblob,
                                                                             tile_cosalpha, margin, nullptr, lowq, nitems, listed,
                                                                             sched, listed ? 16 : 2, nullptr);
    } else
        ms_iterate_d128_f16x_kernel<NW, true><<<grid, 64 * NW, sm, stream>>>(X, blob, newX, bw, flags, N, iters, skip_below, refblob,
                                                                             tile_cosalpha, margin, stats, nullptr, nitems, listed,
                                                                             sched, listed ? 8 : 1, nullptr);
    SED_LAUNCH_CHECK();
    return SED_OK;
}

// forms 1 / 4: the 8-wave kernel with per-wave block skipping on four-plane (RM = false) or row-major (RM = true) stage images
#ifndef F16S_STAG
#define F16S_STAG true
#endif
template <bool RM, int NW = 8, int NT = 4>
static int f16s_launch(int B, int N, int iters, const float* bw, const float* X, float* newX, uint8_t* blob, int* flags,
                       uint8_t* refblob, int* flags2, int* lowq, float skip_below, const float* tile_ref, const float* tile_cosalpha,
                       float margin, unsigned long long* stats, int digits, int* sched, hipStream_t stream) {
    using L = StageLayout<32>;
    const int nst = (N + 31) / 32, nrs = 2 * ((nst + 31) / 32);
    constexpr int sm = RM ? (NW == 8 ? F16S_NBUF_RM : (NT == 4 ? 4 : 3)) * StageLayoutD<NT>::STAGE : F16S_NBUF * L::STAGE;
    static_assert(RM || NW == 8, "four-plane images: 8-wave workgroups only (LDS)");
    hipError_t e = hipSuccess;
    static bool attr = false;
    if (!attr) {
        e = hipFuncSetAttribute((const void*)ms_split_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, L::STAGE);
        if (e != hipSuccess) return (int)e;
        e = hipFuncSetAttribute((const void*)ms_iterate_d128_f16s_kernel<F16S_STAG, true, RM, NW, NT>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, sm);
        if (e != hipSuccess) return (int)e;
        e = hipFuncSetAttribute((const void*)ms_iterate_d128_f16s_kernel<F16S_STAG, false, RM, NW, NT>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, sm);
        if (e != hipSuccess) return (int)e;
        attr = true;
    }
    const int nbx = (N + 32 * NW - 1) / (32 * NW), nitems = nbx * B;
    if (nbx > 255 || B > (1 << 22)) return SED_EUNSUPPORTED;
    static int slots = 0;                                  // resident workgroups: 8 waves of 256 registers per CU
    if (!slots) {
        int dev = 0;
        e = hipGetDevice(&dev);
        if (e == hipSuccess) e = hipDeviceGetAttribute(&slots, hipDeviceAttributeMultiprocessorCount, dev);
        if (e != hipSuccess) return (int)e;
        slots *= 8 / NW;
    }
    const dim3 grid((unsigned)(nitems < slots ? nitems : slots));
    int* item_stages = sched + MS_SCHED_INTS;              // [MS_SCHED_INTS] queues (ms_next_item) | [nitems] first list lengths | [nitems] item list
    int* item_list = item_stages + nitems;
    const int* listed = B <= MS_ORDER_MAX_CLOUDS ? item_list : nullptr;      // (beyond: items in natural order)
    e = hipMemsetAsync(sched, 0, (size_t)(MS_SCHED_INTS + nitems) * sizeof(int), stream);
    if (e != hipSuccess) return (int)e;
    if (RM && NT != 4) {
        ms_split_d_kernel<NT><<<dim3(nst, B), 256, 0, stream>>>(X, bw, blob, flags, N, nst);
        ms_split_d_kernel<NT><<<dim3(nrs, B), 256, 0, stream>>>(tile_ref, bw, refblob, flags2, nrs * 32, nrs);
    } else if (RM) {
        ms_split_n_kernel<<<dim3(nst, B), 256, 0, stream>>>(X, bw, blob, flags, N, nst);
        ms_split_n_kernel<<<dim3(nrs, B), 256, 0, stream>>>(tile_ref, bw, refblob, flags2, nrs * 32, nrs);
    } else {
        ms_split_kernel<32><<<dim3(nst, B), 256, L::STAGE, stream>>>(X, bw, blob, flags, N, nst);
        ms_split_kernel<32><<<dim3(nrs, B), 256, L::STAGE, stream>>>(tile_ref, bw, refblob, flags2, nrs * 32, nrs);
    }
    ms_iterate_d128_f16s_kernel<F16S_STAG, true, RM, NW, NT><<<grid, 64 * NW, sm, stream>>>(
        X, blob, newX, bw, flags, N, iters, skip_below, refblob, tile_cosalpha, margin, nullptr, nullptr, nitems, nullptr, sched, 0,
        item_stages);
    if (listed) ms_sparse_item_order_kernel<<<1, 1024, 0, stream>>>(item_stages, B, nbx, item_list, sched);
    if (digits != 2) {        // heads-only weights; flagged clouds again with (h, l) weights
        ms_iterate_d128_f16s_kernel<F16S_STAG, false, RM, NW, NT><<<grid, 64 * NW, sm, stream>>>(
            X, blob, newX, bw, flags, N, iters, skip_below, refblob, tile_cosalpha, margin, stats, lowq, nitems, listed, sched,
            listed ? 8 : 1, nullptr);
        ms_iterate_d128_f16s_kernel<F16S_STAG, true, RM, NW, NT><<<grid, 64 * NW, sm, stream>>>(
            X, blob, newX, bw, flags, N, iters, skip_below, refblob, tile_cosalpha, margin, nullptr, lowq, nitems, listed, sched,
            listed ? 16 : 2, nullptr);
    } else
        ms_iterate_d128_f16s_kernel<F16S_STAG, true, RM, NW, NT><<<grid, 64 * NW, sm, stream>>>(
            X, blob, newX, bw, flags, N, iters, skip_below, refblob, tile_cosalpha, margin, stats, nullptr, nitems, listed, sched,
            listed ? 8 : 1, nullptr);
    SED_LAUNCH_CHECK();
    return SED_OK;
}

// form: 1 = round 2's 8-wave kernel on four-plane images (ms_iterate_d128_f16s_kernel); 2 / 3 = the 64-queries-per-wave kernel on
// row-major images with 2- / 4-wave workgroups (ms_iterate_d128_f16x_kernel); 4 = the 8-wave kernel on row-major images; 5 = the same with 4-wave workgroups; 0 = default
constexpr int MS_SPARSE_DEFAULT_FORM = 5;
int ms_f16_sparse_launch(int B, int N, int d, int iters, const float* bw, const float* X, float* newX, void* workspace,
                         int** flags_out, float skip_below, const float* tile_ref, const float* tile_cosalpha,
                         float margin, unsigned long long* stats, int digits, int form, hipStream_t stream) {
    const int nst = (N + 31) / 32, nrs = 2 * ((nst + 31) / 32);
    if (nst > 64 * F16S_MAXW) return SED_EUNSUPPORTED;
    if (form == 0) form = MS_SPARSE_DEFAULT_FORM;
    uint8_t* blob = (uint8_t*)workspace;
    int* flags = (int*)(blob + f16_blob_bytes_4(B, N));
    uint8_t* refblob = (uint8_t*)flags + f16_flag_bytes(B);
    int* flags2 = (int*)(refblob + f16_blob_bytes_4(B, nrs * 32));
    int* lowq = (int*)((uint8_t*)flags2 + f16_flag_bytes(B));           // clouds whose weighted means cancel (heads-only pass)
    int* sched = (int*)((uint8_t*)lowq + f16_flag_bytes(B));
    *flags_out = flags;
    hipError_t e = hipMemsetAsync(flags, 0, (size_t)B * sizeof(int), stream);
    if (e != hipSuccess) return (int)e;
    e = hipMemsetAsync(lowq, 0, (size_t)B * sizeof(int), stream);
    if (e != hipSuccess) return (int)e;
    if (form == 2)       // (the workspace is sized for the four-plane images: the row-major ones use the same carve-up)
        return f16x_launch<2>(B, N, iters, bw, X, newX, blob, flags, refblob, flags2, lowq, skip_below, tile_ref, tile_cosalpha, margin,
                              stats, digits, sched, stream);
    if (form == 3)
        return f16x_launch<4>(B, N, iters, bw, X, newX, blob, flags, refblob, flags2, lowq, skip_below, tile_ref, tile_cosalpha, margin,
                              stats, digits, sched, stream);
    if (d == 160) {                                        // the HPNet-widened embedding: five feature tiles, default form only
        if (form != 5) return SED_EUNSUPPORTED;
        return f16s_launch<true, 4, 5>(B, N, iters, bw, X, newX, blob, flags, refblob, flags2, lowq, skip_below, tile_ref, tile_cosalpha,
                                       margin, stats, digits, sched, stream);
    }
    if (d != 128) return SED_EUNSUPPORTED;
    if (form == 5)
        return f16s_launch<true, 4>(B, N, iters, bw, X, newX, blob, flags, refblob, flags2, lowq, skip_below, tile_ref, tile_cosalpha, margin,
                                    stats, digits, sched, stream);
    if (form == 4)
        return f16s_launch<true>(B, N, iters, bw, X, newX, blob, flags, refblob, flags2, lowq, skip_below, tile_ref, tile_cosalpha, margin,
                                 stats, digits, sched, stream);
    return f16s_launch<false>(B, N, iters, bw, X, newX, blob, flags, refblob, flags2, lowq, skip_below, tile_ref, tile_cosalpha, margin,
                              stats, digits, sched, stream);
}
