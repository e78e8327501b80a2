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

// One workgroup; the errors of the kept rays are sorted in LDS (bitonic, +inf padding), the lower median
// sorted[(n-1)/2] is what torch.median returns.
#define TM_MAX 8192
__global__ __launch_bounds__(1024) void tracking_mask_kernel(const float* __restrict__ depth,
                                                             const float* __restrict__ gt_depth,
                                                             const uint8_t* __restrict__ keep, int R, float factor,
                                                             uint8_t* __restrict__ mask) {
    __shared__ float v[TM_MAX];
    __shared__ int cnt[2];
    if (threadIdx.x < 2) cnt[threadIdx.x] = 0;
    __syncthreads();
    int npow = 1;
    while (npow < R) npow <<= 1;
    int kept = 0, nans = 0;
    for (int t = threadIdx.x; t < npow; t += blockDim.x) {
        float e = __builtin_inff();
        if (t < R && (!keep || keep[t])) {
            e = fabsf(gt_depth[t] - depth[t]);
            ++kept;
            if (e != e) { ++nans; e = __builtin_inff(); }
        }
        v[t] = e;
    }
    if (kept) atomicAdd(&cnt[0], kept);
    if (nans) atomicAdd(&cnt[1], nans);
    __syncthreads();
    for (int size = 2; size <= npow; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = threadIdx.x; t < npow / 2; t += blockDim.x) {
                const int lo = ((t / stride) * stride * 2) + (t % stride);
                const int hi = lo + stride;
                const bool up = ((lo & size) == 0);
                const float a = v[lo], c = v[hi];
                if ((a > c) == up) { v[lo] = c; v[hi] = a; }
            }
            __syncthreads();
        }
    }
    const int n = cnt[0];
    // torch.median propagates NaN; over an empty set nothing can pass the test
    const float med = (cnt[1] > 0 || n == 0) ? __builtin_nanf("") : v[(n - 1) / 2];
    const float thr = factor * med;
    for (int t = threadIdx.x; t < R; t += blockDim.x) {
        const bool k = !keep || keep[t];
        mask[t] = (k && fabsf(gt_depth[t] - depth[t]) < thr) ? 1 : 0;
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
