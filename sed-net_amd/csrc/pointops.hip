// Native point-cloud ops of the reference, rewritten for gfx950 (wave64): chamfer distance (forward + a deterministic
// backward) and the PointNet++ operator set (furthest point sampling, ball query, group / gather, three-NN,
// three-interpolate; forward). SURVEY.md section 8(a) rows a17 / a18.
//
// Replaces /root/reference/src/chamfer_distance/chamfer_distance.cu:6-205 and
// /root/reference/Fitting_patches_and_edges/pointnet2/_ext_src/src/{sampling,ball_query,group_points,interpolate}_gpu.cu.
// All of them are streaming, HBM/L2-bound integer-and-float scans (no GEMM shape): one thread per query point with the
// scanned cloud tiled through LDS where it is re-read, lanes = consecutive points for coalesced loads/stores, wave
// shuffles for the FPS arg-max. Differences by design: every op runs on the caller's stream and reports errors by
// return code (the reference launches chamfer on the default stream and printf's / exit(-1)s on errors); the chamfer
// backward gathers instead of scattering with float atomics, so it is deterministic.
#include "common.h"

namespace {

constexpr int TILE = 512;

// ---- chamfer forward: nearest neighbour of every xyz1 point in xyz2 (squared distance + index, ties -> lowest) ----
__global__ __launch_bounds__(256) void chamfer_nn_kernel(int n, const float* __restrict__ xyz1, int m,
                                                         const float* __restrict__ xyz2, float* __restrict__ dist,
                                                         int* __restrict__ idx) {
    __shared__ float buf[TILE * 3];
    const int b = blockIdx.y;
    const float* p1 = xyz1 + (size_t)b * n * 3;
    const float* p2 = xyz2 + (size_t)b * m * 3;
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int jc = j < n ? j : n - 1;
    const float x1 = p1[jc * 3], y1 = p1[jc * 3 + 1], z1 = p1[jc * 3 + 2];
    float best = 0.f;
    int best_i = 0;
    for (int k0 = 0; k0 < m; k0 += TILE) {
        const int cnt = min(TILE, m - k0);
        __syncthreads();
        for (int t = threadIdx.x; t < cnt * 3; t += 256) buf[t] = p2[(size_t)k0 * 3 + t];
        __syncthreads();
        for (int k = 0; k < cnt; ++k) {
            const float dx = buf[k * 3] - x1, dy = buf[k * 3 + 1] - y1, dz = buf[k * 3 + 2] - z1;
            const float d = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
            if ((k0 + k) == 0 || d < best) { best = d; best_i = k0 + k; }       // strict <: earliest index wins
        }
    }
    if (j < n) { dist[(size_t)b * n + j] = best; idx[(size_t)b * n + j] = best_i; }
}

// ---- chamfer backward, deterministic: grad1[i] = 2 g1[i] (x1_i - x2_idx1[i]) - sum_{j: idx2[j] == i} 2 g2[j] (x2_j - x1_i)
__global__ __launch_bounds__(256) void chamfer_grad_kernel(int n, const float* __restrict__ xyz1, int m,
                                                           const float* __restrict__ xyz2,
                                                           const float* __restrict__ g1, const int* __restrict__ idx1,
                                                           const float* __restrict__ g2, const int* __restrict__ idx2,
                                                           float* __restrict__ grad1) {
    __shared__ int ibuf[TILE];
    const int b = blockIdx.y;
    const float* p1 = xyz1 + (size_t)b * n * 3;
    const float* p2 = xyz2 + (size_t)b * m * 3;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int ic = i < n ? i : n - 1;
    const float x1 = p1[ic * 3], y1 = p1[ic * 3 + 1], z1 = p1[ic * 3 + 2];
    const int j2 = idx1[(size_t)b * n + ic];
    const float g = g1[(size_t)b * n + ic] * 2.f;
    float ax = g * (x1 - p2[j2 * 3]), ay = g * (y1 - p2[j2 * 3 + 1]), az = g * (z1 - p2[j2 * 3 + 2]);
    for (int k0 = 0; k0 < m; k0 += TILE) {
        const int cnt = min(TILE, m - k0);
        __syncthreads();
        for (int t = threadIdx.x; t < cnt; t += 256) ibuf[t] = idx2[(size_t)b * m + k0 + t];
        __syncthreads();
        for (int k = 0; k < cnt; ++k)
            if (ibuf[k] == i) {                    // rare: only the points whose nearest neighbour is i
                const int jj = k0 + k;
                const float gg = g2[(size_t)b * m + jj] * 2.f;
                ax -= gg * (p2[jj * 3] - x1);
                ay -= gg * (p2[jj * 3 + 1] - y1);
                az -= gg * (p2[jj * 3 + 2] - z1);
            }
    }
    if (i < n) {
        float* o = grad1 + ((size_t)b * n + i) * 3;
        o[0] = ax; o[1] = ay; o[2] = az;
    }
}

// ---- furthest point sampling: one workgroup per cloud ------------------------------------------------------------------
__global__ __launch_bounds__(512) void fps_kernel(int n, int m, const float* __restrict__ xyz, float* __restrict__ temp,
                                                  int* __restrict__ idxs) {
    __shared__ float wbest[8];
    __shared__ int wbesti[8];
    __shared__ int cur;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* p = xyz + (size_t)b * n * 3;
    float* tmp = temp + (size_t)b * n;
    int* out = idxs + (size_t)b * m;
    for (int k = tid; k < n; k += 512) tmp[k] = 1e10f;                          // sampling.cpp:78-80
    int old = 0;
    if (tid == 0) out[0] = 0;
    __syncthreads();
    for (int j = 1; j < m; ++j) {
        const float x1 = p[old * 3], y1 = p[old * 3 + 1], z1 = p[old * 3 + 2];
        float best = -1.f;
        int besti = 0;
        for (int k = tid; k < n; k += 512) {
            const float x2 = p[k * 3], y2 = p[k * 3 + 1], z2 = p[k * 3 + 2];
            const float mag = x2 * x2 + y2 * y2 + z2 * z2;
            if (mag <= 1e-3f) continue;                                         // sampling_gpu.cu:105-106
            const float dx = x2 - x1, dy = y2 - y1, dz = z2 - z1;
            const float d2 = fminf(dx * dx + dy * dy + dz * dz, tmp[k]);
            tmp[k] = d2;
            if (d2 > best) { best = d2; besti = k; }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ob = __shfl_xor(best, off, 64);
            const int oi = __shfl_xor(besti, off, 64);
            if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
        }
        if (lane == 0) { wbest[wave] = best; wbesti[wave] = besti; }
        __syncthreads();
        if (tid == 0) {
            float bb = wbest[0];
            int bi = wbesti[0];
            for (int w = 1; w < 8; ++w)
                if (wbest[w] > bb || (wbest[w] == bb && wbesti[w] < bi)) { bb = wbest[w]; bi = wbesti[w]; }
            cur = bi;
            out[j] = bi;
        }
        __syncthreads();
        old = cur;
    }
}

// ---- ball query: first nsample indices with d^2 < r^2, padded with the first hit ---------------------------------------
__global__ __launch_bounds__(256) void ball_query_kernel(int n, int m, float radius, int nsample,
                                                         const float* __restrict__ new_xyz,
                                                         const float* __restrict__ xyz, int* __restrict__ idx) {
    __shared__ float buf[TILE * 3];
    const int b = blockIdx.y;
    const float* q = new_xyz + (size_t)b * m * 3;
    const float* p = xyz + (size_t)b * n * 3;
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int jc = j < m ? j : m - 1;
    const float nx = q[jc * 3], ny = q[jc * 3 + 1], nz = q[jc * 3 + 2];
    const float r2 = radius * radius;
    int* o = idx + ((size_t)b * m + jc) * nsample;
    int cnt = j < m ? 0 : nsample;
    for (int k0 = 0; k0 < n; k0 += TILE) {
        const int c = min(TILE, n - k0);
        __syncthreads();
        for (int t = threadIdx.x; t < c * 3; t += 256) buf[t] = p[(size_t)k0 * 3 + t];
        __syncthreads();
        for (int k = 0; k < c && cnt < nsample; ++k) {
            const float dx = nx - buf[k * 3], dy = ny - buf[k * 3 + 1], dz = nz - buf[k * 3 + 2];
            const float d2 = dx * dx + dy * dy + dz * dz;
            if (d2 < r2) {
                if (cnt == 0)
                    for (int l = 0; l < nsample; ++l) o[l] = k0 + k;
                o[cnt] = k0 + k;
                ++cnt;
            }
        }
    }
}

// out[b,c,j,s] = points[b,c,idx[b,j,s]]    (group_points; gather_points is the nsample = 1 case)
__global__ void group_points_kernel(int c, int n, int npoints, int nsample, const float* __restrict__ points,
                                    const int* __restrict__ idx, float* __restrict__ out) {
    const int b = blockIdx.z, ch = blockIdx.y;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= npoints * nsample) return;
    const int ii = idx[(size_t)b * npoints * nsample + e];
    out[((size_t)b * c + ch) * npoints * nsample + e] = points[((size_t)b * c + ch) * n + ii];
}

// three nearest known points of every unknown point (squared distances ascending)
__global__ __launch_bounds__(256) void three_nn_kernel(int n, int m, const float* __restrict__ unknown,
                                                       const float* __restrict__ known, float* __restrict__ dist2,
                                                       int* __restrict__ idx) {
    __shared__ float buf[TILE * 3];
    const int b = blockIdx.y;
    const float* u = unknown + (size_t)b * n * 3;
    const float* kn = known + (size_t)b * m * 3;
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int jc = j < n ? j : n - 1;
    const float ux = u[jc * 3], uy = u[jc * 3 + 1], uz = u[jc * 3 + 2];
    float b1 = 3.0e38f, b2 = 3.0e38f, b3 = 3.0e38f;          // reference: double 1e40 sentinels (interpolate_gpu.cu:33)
    int i1 = 0, i2 = 0, i3 = 0;
    for (int k0 = 0; k0 < m; k0 += TILE) {
        const int c = min(TILE, m - k0);
        __syncthreads();
        for (int t = threadIdx.x; t < c * 3; t += 256) buf[t] = kn[(size_t)k0 * 3 + t];
        __syncthreads();
        for (int k = 0; k < c; ++k) {
            const float dx = ux - buf[k * 3], dy = uy - buf[k * 3 + 1], dz = uz - buf[k * 3 + 2];
            const float d = dx * dx + dy * dy + dz * dz;
            if (d < b1) { b3 = b2; i3 = i2; b2 = b1; i2 = i1; b1 = d; i1 = k0 + k; }
            else if (d < b2) { b3 = b2; i3 = i2; b2 = d; i2 = k0 + k; }
            else if (d < b3) { b3 = d; i3 = k0 + k; }
        }
    }
    if (j < n) {
        float* od = dist2 + ((size_t)b * n + j) * 3;
        int* oi = idx + ((size_t)b * n + j) * 3;
        od[0] = b1; od[1] = b2; od[2] = b3;
        oi[0] = i1; oi[1] = i2; oi[2] = i3;
    }
}

// out[b,c,j] = sum_t points[b,c,idx[b,j,t]] * weight[b,j,t]
__global__ void three_interpolate_kernel(int c, int m, int n, const float* __restrict__ points,
                                         const int* __restrict__ idx, const float* __restrict__ weight,
                                         float* __restrict__ out) {
    const int b = blockIdx.z, ch = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int* ii = idx + ((size_t)b * n + j) * 3;
    const float* w = weight + ((size_t)b * n + j) * 3;
    const float* p = points + ((size_t)b * c + ch) * m;
    out[((size_t)b * c + ch) * n + j] = p[ii[0]] * w[0] + p[ii[1]] * w[1] + p[ii[2]] * w[2];
}

}  // namespace

// dist1/idx1 [B,n]: nearest xyz2 point of every xyz1 point; dist2/idx2 [B,m] the other direction.
// src/chamfer_distance/chamfer_distance.cu:6-155 (ChamferDistanceKernel + launcher)
extern "C" int sed_chamfer_fwd_f32(int B, int n, int m, const float* xyz1, const float* xyz2, float* dist1, int* idx1,
                                   float* dist2, int* idx2, hipStream_t stream) {
    if (B <= 0 || n <= 0 || m <= 0 || !xyz1 || !xyz2 || !dist1 || !idx1 || !dist2 || !idx2) return SED_EINVAL;
    chamfer_nn_kernel<<<dim3((n + 255) / 256, B), 256, 0, stream>>>(n, xyz1, m, xyz2, dist1, idx1);
    chamfer_nn_kernel<<<dim3((m + 255) / 256, B), 256, 0, stream>>>(m, xyz2, n, xyz1, dist2, idx2);
    SED_LAUNCH_CHECK();
    return SED_OK;
}

// src/chamfer_distance/chamfer_distance.cu:158-205 (grad kernel + launcher); outputs fully overwritten, no atomics
extern "C" int sed_chamfer_bwd_f32(int B, int n, int m, const float* xyz1, const float* xyz2, const float* grad_dist1,
                                   const int* idx1, const float* grad_dist2, const int* idx2, float* grad_xyz1,
                                   float* grad_xyz2, hipStream_t stream) {
    if (B <= 0 || n <= 0 || m <= 0 || !xyz1 || !xyz2 || !grad_dist1 || !idx1 || !grad_dist2 || !idx2 || !grad_xyz1 ||
        !grad_xyz2)
        return SED_EINVAL;
    chamfer_grad_kernel<<<dim3((n + 255) / 256, B), 256, 0, stream>>>(n, xyz1, m, xyz2, grad_dist1, idx1, grad_dist2, idx2,
                                                                      grad_xyz1);
    chamfer_grad_kernel<<<dim3((m + 255) / 256, B), 256, 0, stream>>>(m, xyz2, n, xyz1, grad_dist2, idx2, grad_dist1, idx1,
                                                                      grad_xyz2);
    SED_LAUNCH_CHECK();
    return SED_OK;
}

// pointnet2/_ext_src/src/sampling_gpu.cu:74-178; temp_ws [B*n] floats scratch; idx [B,m]
extern "C" int sed_furthest_point_sampling_f32(int B, int n, int m, const float* xyz, float* temp_ws, int* idx,
                                               hipStream_t stream) {
    if (B <= 0 || n <= 0 || m <= 0 || m > n || !xyz || !temp_ws || !idx) return SED_EINVAL;
    fps_kernel<<<B, 512, 0, stream>>>(n, m, xyz, temp_ws, idx);
    SED_LAUNCH_CHECK();
    return SED_OK;
}

// pointnet2/_ext_src/src/ball_query_gpu.cu:14-49; idx [B,m,nsample] must be zero-initialised by the caller (rows with no
// hit keep zeros, like the reference's torch::zeros)
extern "C" int sed_ball_query_f32(int B, int n, int m, float radius, int nsample, const float* new_xyz, const float* xyz,
                                  int* idx, hipStream_t stream) {
    if (B <= 0 || n <= 0 || m <= 0 || nsample <= 0 || !new_xyz || !xyz || !idx) return SED_EINVAL;
    ball_query_kernel<<<dim3((m + 255) / 256, B), 256, 0, stream>>>(n, m, radius, nsample, new_xyz, xyz, idx);
    SED_LAUNCH_CHECK();
    return SED_OK;
}

// pointnet2/_ext_src/src/group_points_gpu.cu:13-42 (nsample >= 1) and sampling_gpu.cu:13-33 gather_points (nsample = 1)
extern "C" int sed_group_points_f32(int B, int c, int n, int npoints, int nsample, const float* points, const int* idx,
                                    float* out, hipStream_t stream) {
    if (B <= 0 || c <= 0 || n <= 0 || npoints <= 0 || nsample <= 0 || !points || !idx || !out) return SED_EINVAL;
    group_points_kernel<<<dim3((npoints * nsample + 255) / 256, c, B), 256, 0, stream>>>(c, n, npoints, nsample, points, idx, out);
    SED_LAUNCH_CHECK();
    return SED_OK;
}

// pointnet2/_ext_src/src/interpolate_gpu.cu:14-64
extern "C" int sed_three_nn_f32(int B, int n, int m, const float* unknown, const float* known, float* dist2, int* idx,
                                hipStream_t stream) {
    if (B <= 0 || n <= 0 || m <= 0 || !unknown || !known || !dist2 || !idx) return SED_EINVAL;
    three_nn_kernel<<<dim3((n + 255) / 256, B), 256, 0, stream>>>(n, m, unknown, known, dist2, idx);
    SED_LAUNCH_CHECK();
    return SED_OK;
}

// pointnet2/_ext_src/src/interpolate_gpu.cu:77-109
extern "C" int sed_three_interpolate_f32(int B, int c, int m, int n, const float* points, const int* idx,
                                         const float* weight, float* out, hipStream_t stream) {
    if (B <= 0 || c <= 0 || m <= 0 || n <= 0 || !points || !idx || !weight || !out) return SED_EINVAL;
    three_interpolate_kernel<<<dim3((n + 255) / 256, c, B), 256, 0, stream>>>(c, m, n, points, idx, weight, out);
    SED_LAUNCH_CHECK();
    return SED_OK;
}
