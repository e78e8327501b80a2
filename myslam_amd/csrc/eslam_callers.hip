// Small caller-side kernels of the tracking / mapping iterations (the glue between get_samples, the renderer and the
// optimiser).  Each replaces a chain of 5-40 tiny PyTorch launches in the reference's loops; inside a captured
// hipGraph every node costs ~8 us regardless of its size, so the chains - not the render kernels - set the
// iteration time of the tracker (0.90 ms with them, of which 0.26 ms is rendering).
//   pose_to_c2w (+bwd)   src/common.py:169-181 cam_pose_to_matrix (pytorch3d quaternion_to_matrix + translation)
//   tracking_mask        src/Tracker.py:192-195: |gt_depth - depth| < 10 * median, over the pre-filtered rays
//   keep_best            src/Tracker.py:304-307: remember the pose of the smallest loss
#include "eslam_common.h"

// rotation part: R = I + s * P(q), s = 2 / |q|^2, q = (r, i, j, k) real first
__global__ void pose_to_c2w_kernel(const float* __restrict__ poses, int b, float* __restrict__ c2ws) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= b) return;
    const float* q = poses + 7 * n;
    const float r = q[0], i = q[1], j = q[2], k = q[3];
    const float s = 2.0f / (r * r + i * i + j * j + k * k);
    float* m = c2ws + 16 * n;
    m[0] = 1.0f - s * (j * j + k * k); m[1] = s * (i * j - k * r);        m[2] = s * (i * k + j * r);         m[3] = q[4];
    m[4] = s * (i * j + k * r);        m[5] = 1.0f - s * (i * i + k * k); m[6] = s * (j * k - i * r);         m[7] = q[5];
    m[8] = s * (i * k - j * r);        m[9] = s * (j * k + i * r);        m[10] = 1.0f - s * (i * i + j * j); m[11] = q[6];
    m[12] = 0.0f; m[13] = 0.0f; m[14] = 0.0f; m[15] = 1.0f;
}

__global__ void pose_to_c2w_bwd_kernel(const float* __restrict__ poses, const float* __restrict__ g_c2ws, int b,
                                       float* __restrict__ g_poses) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= b) return;
    const float* q = poses + 7 * n;
    const float* G = g_c2ws + 16 * n;
    const float r = q[0], i = q[1], j = q[2], k = q[3];
    const float s = 2.0f / (r * r + i * i + j * j + k * k);
    const float G00 = G[0], G01 = G[1], G02 = G[2], G10 = G[4], G11 = G[5], G12 = G[6], G20 = G[8], G21 = G[9], G22 = G[10];
    // sum_ab G_ab P_ab (the part that multiplies ds/dq = -s^2 q)
    const float GP = -G00 * (j * j + k * k) + G01 * (i * j - k * r) + G02 * (i * k + j * r) + G10 * (i * j + k * r) -
                     G11 * (i * i + k * k) + G12 * (j * k - i * r) + G20 * (i * k - j * r) + G21 * (j * k + i * r) -
                     G22 * (i * i + j * j);
    const float dr = -k * G01 + j * G02 + k * G10 - i * G12 - j * G20 + i * G21;
    const float di = j * G01 + k * G02 + j * G10 - 2.0f * i * G11 - r * G12 + k * G20 + r * G21 - 2.0f * i * G22;
    const float dj = -2.0f * j * G00 + i * G01 + r * G02 + i * G10 + k * G12 - r * G20 + k * G21 - 2.0f * j * G22;
    const float dk = -2.0f * k * G00 - r * G01 + i * G02 + r * G10 - 2.0f * k * G11 + j * G12 + i * G20 + j * G21;
    float* g = g_poses + 7 * n;
    const float s2 = s * s * GP;
    g[0] = s * dr - s2 * r;
    g[1] = s * di - s2 * i;
    g[2] = s * dj - s2 * j;
    g[3] = s * dk - s2 * k;
    g[4] = G[3]; g[5] = G[7]; g[6] = G[11];
}

// One workgroup.  The lower median sorted[(n-1)/2] of the kept rays' errors (what torch.median returns) is found by a radix
// select over the errors' bit patterns - non-negative floats order like their unsigned bits - in three passes of 11 + 11 + 10
// bits: histogram of the digit in LDS (integer atomics), scan, the bin that holds rank k becomes the prefix of the next pass.
// ~10 workgroup barriers whatever R is; the bitonic sort this replaces took 66 and 31 us at 2000 rays - as long as the
// tracking iteration's forward kernel, and in its critical path (Tracker.py:192-195 sits between the render and the loss).
#define TM_MAX 8192
#define TM_PER (TM_MAX / 1024)
__global__ __launch_bounds__(1024) void tracking_mask_kernel(const float* __restrict__ depth,
                                                             const float* __restrict__ gt_depth,
                                                             const uint8_t* __restrict__ keep, int R, float factor,
                                                             uint8_t* __restrict__ mask) {
    __shared__ unsigned hist[2048];
    __shared__ unsigned wtot[16];
    __shared__ unsigned sel[2];                        // the chosen bin, and the rank inside it
    __shared__ int cnt[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 2) cnt[tid] = 0;
    // this thread's errors stay in registers: element t = tid + 1024 j
    unsigned key[TM_PER];
    int kept = 0, nans = 0;
#pragma unroll
    for (int j = 0; j < TM_PER; ++j) {
        const int t = tid + 1024 * j;
        key[j] = 0xFFFFFFFFu;                          // not part of the set
        if (t < R && (!keep || keep[t])) {
            float e = fabsf(gt_depth[t] - depth[t]);
            ++kept;
            if (e != e) { ++nans; e = __builtin_inff(); }
            key[j] = __float_as_uint(e);               // e >= 0: sign bit clear, so 0xFFFFFFFF never collides with a member
        }
    }
    __syncthreads();
    if (kept) atomicAdd(&cnt[0], kept);
    if (nans) atomicAdd(&cnt[1], nans);
    __syncthreads();
    const int n = cnt[0];
    unsigned k = n > 0 ? (unsigned)((n - 1) / 2) : 0u; // rank of the lower median among the members
    unsigned prefix = 0u;                              // the median's leading bits found so far
    // pass p: digit = (key >> shift) & (nbins - 1) of the members whose higher bits equal `prefix`
    const int shifts[3] = {21, 10, 0}, nbits[3] = {11, 11, 10};
    for (int p = 0; p < 3 && n > 0; ++p) {
        const int shift = shifts[p];
        const unsigned nbins = 1u << nbits[p];
        for (int i = tid; i < 2048; i += 1024) hist[i] = 0u;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < TM_PER; ++j) {
            const bool in = key[j] != 0xFFFFFFFFu && (p == 0 || (key[j] >> (shift + nbits[p])) == prefix);
            if (in) atomicAdd(&hist[(key[j] >> shift) & (nbins - 1u)], 1u);
        }
        __syncthreads();
        // inclusive scan over the bins: thread t owns bins 2t, 2t + 1
        const unsigned h0 = hist[2 * tid], h1 = hist[2 * tid + 1];
        const unsigned incl = wave_incl_sum_u(h0 + h1);
        if (lane == 63) wtot[wave] = incl;
        __syncthreads();
        unsigned before = incl - (h0 + h1);
        for (int w = 0; w < wave; ++w) before += wtot[w];
        // exactly one bin has before <= k < before + count
        if (k >= before && k < before + h0) { sel[0] = 2u * tid; sel[1] = k - before; }
        else if (k >= before + h0 && k < before + h0 + h1) { sel[0] = 2u * tid + 1u; sel[1] = k - before - h0; }
        __syncthreads();
        prefix = (prefix << nbits[p]) | sel[0];
        k = sel[1];
        __syncthreads();                               // sel and hist are rewritten by the next pass
    }
    // torch.median propagates NaN; over an empty set nothing can pass the test
    const float med = (cnt[1] > 0 || n == 0) ? __builtin_nanf("") : __uint_as_float(prefix);
    const float thr = factor * med;
    for (int t = tid; t < R; t += 1024) {
        const bool kk = !keep || keep[t];
        mask[t] = (kk && fabsf(gt_depth[t] - depth[t]) < thr) ? 1 : 0;
    }
}

__global__ void keep_best_kernel(const float* __restrict__ loss, const float* __restrict__ pose, int n,
                                 float* __restrict__ best, float* __restrict__ best_pose) {
    // single workgroup; every thread reads the old best before anyone overwrites it
    const float l = loss[0], b = best[0];
    __syncthreads();
    if (l < b) {
        for (int t = threadIdx.x; t < n; t += blockDim.x) best_pose[t] = pose[t];
        if (threadIdx.x == 0) best[0] = l;
    }
}

extern "C" int eslam_pose_to_c2w(const float* poses, int b, float* c2ws, eslam_stream_t stream) {
    if (b <= 0) return 0;
    if (!poses || !c2ws) {
        eslam_set_error("eslam_pose_to_c2w: null argument");
        return 1;
    }
    hipLaunchKernelGGL(pose_to_c2w_kernel, dim3((b + 63) / 64), dim3(64), 0, (hipStream_t)stream, poses, b, c2ws);
    return eslam_check_launch("pose_to_c2w_kernel");
}

extern "C" int eslam_pose_to_c2w_bwd(const float* poses, const float* g_c2ws, int b, float* g_poses,
                                     eslam_stream_t stream) {
    if (b <= 0) return 0;
    if (!poses || !g_c2ws || !g_poses) {
        eslam_set_error("eslam_pose_to_c2w_bwd: null argument");
        return 1;
    }
    hipLaunchKernelGGL(pose_to_c2w_bwd_kernel, dim3((b + 63) / 64), dim3(64), 0, (hipStream_t)stream, poses, g_c2ws, b,
                       g_poses);
    return eslam_check_launch("pose_to_c2w_bwd_kernel");
}

extern "C" int eslam_tracking_mask(const float* depth, const float* gt_depth, const uint8_t* keep, int R, float factor,
                                   uint8_t* mask, eslam_stream_t stream) {
    if (R <= 0) return 0;
    if (R > ESLAM_TRACKING_MASK_MAX) {
        eslam_set_error("eslam_tracking_mask: %d rays exceed the single-workgroup limit of %d", R, ESLAM_TRACKING_MASK_MAX);
        return 1;
    }
    if (!depth || !gt_depth || !mask) {
        eslam_set_error("eslam_tracking_mask: null argument");
        return 1;
    }
    hipLaunchKernelGGL(tracking_mask_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, depth, gt_depth, keep, R,
                       factor, mask);
    return eslam_check_launch("tracking_mask_kernel");
}

extern "C" int eslam_keep_best(const float* loss, const float* pose, int n, float* best, float* best_pose,
                               eslam_stream_t stream) {
    if (n <= 0) return 0;
    if (!loss || !pose || !best || !best_pose) {
        eslam_set_error("eslam_keep_best: null argument");
        return 1;
    }
    hipLaunchKernelGGL(keep_best_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, loss, pose, n, best, best_pose);
    return eslam_check_launch("keep_best_kernel");
}
