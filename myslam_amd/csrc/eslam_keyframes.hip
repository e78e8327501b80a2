// Keyframe selection by view overlap: the projection test of reference
// src/Mapper.py:170-201 for all keyframes in one launch.  The reference forms [K, N, 4, 1] homogeneous points and
// runs ~30 small ops plus a batched 4x4 inverse per mapped frame; here one workgroup per keyframe inverts its
// camera-to-world matrix, walks the N = n_rays x num_samples points and counts those that land inside the image.
#include "eslam_common.h"

// general 4x4 inverse by cofactors (what torch.inverse returns for these well-conditioned rigid matrices, to rounding)
__device__ void invert4(const float* m, float* inv) {
    float a[16];
    a[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    a[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    a[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    a[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    a[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    a[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    a[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    a[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    a[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    a[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    a[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    a[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    a[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    a[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    a[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    a[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    const float det = m[0] * a[0] + m[1] * a[4] + m[2] * a[8] + m[3] * a[12];
    const float r = 1.0f / det;
    for (int i = 0; i < 16; ++i) inv[i] = a[i] * r;
}

__global__ __launch_bounds__(256) void keyframe_overlap_kernel(const float* __restrict__ rays_o,
                                                               const float* __restrict__ rays_d,
                                                               const float* __restrict__ gt_depth, int n_rays,
                                                               int num_samples, const float* __restrict__ c2ws, int n_keyframes,
                                                               int H, int W, float fx, float fy, float cx, float cy, int edge,
                                                               int32_t* __restrict__ counts) {
    // counts[k] = points inside keyframe k's image, counts[K] = rays with depth > 0 (written by workgroup 0)
    __shared__ float w2c[16];
    __shared__ int red[4][2];
    const int k = blockIdx.x;
    const bool have_kf = k < n_keyframes;           // with no keyframes one workgroup still counts the rays with depth
    if (threadIdx.x == 0) {
        if (have_kf) invert4(c2ws + 16 * k, w2c);
        else for (int i = 0; i < 16; ++i) w2c[i] = 0.0f;
    }
    __syncthreads();
    const float step = 1.0f / (float)(num_samples - 1);
    const int total = n_rays * num_samples;
    int inside = 0, valid = 0;
    for (int i = threadIdx.x; i < total; i += blockDim.x) {
        const int ray = i / num_samples, j = i - ray * num_samples;
        const float dep = gt_depth[ray];
        if (!(dep > 0.0f)) continue;                                       // Mapper.py:171-174
        valid += (j == 0);
        // torch.linspace(0, 1, steps): start + j*step in the lower half, end - (steps-1-j)*step in the upper
        const float t = num_samples == 1 ? 0.0f : (j < num_samples / 2 ? (float)j * step : 1.0f - (float)(num_samples - 1 - j) * step);
        const float z = (dep * 0.8f) * (1.0f - t) + (dep + 0.5f) * t;      // Mapper.py:177-179
        const float px = rays_o[3 * ray + 0] + rays_d[3 * ray + 0] * z;
        const float py = rays_o[3 * ray + 1] + rays_d[3 * ray + 1] * z;
        const float pz = rays_o[3 * ray + 2] + rays_d[3 * ray + 2] * z;
        float X = w2c[0] * px + w2c[1] * py + w2c[2] * pz + w2c[3];
        const float Y = w2c[4] * px + w2c[5] * py + w2c[6] * pz + w2c[7];
        const float Z = w2c[8] * px + w2c[9] * py + w2c[10] * pz + w2c[11];
        X = -X;                                                            // Mapper.py:192
        const float zc = Z + 1e-5f;
        const float u = (fx * X + cx * Z) / zc;                            // K @ cam, Mapper.py:190-195
        const float v = (fy * Y + cy * Z) / zc;
        const bool in = (u < (float)(W - edge)) && (u > (float)edge) && (v < (float)(H - edge)) && (v > (float)edge) && (zc < 0.0f);
        inside += in ? 1 : 0;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int o = 32; o > 0; o >>= 1) {
        inside += __shfl_down(inside, o, WAVE);
        valid += __shfl_down(valid, o, WAVE);
    }
    if (lane == 0) { red[wave][0] = inside; red[wave][1] = valid; }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (have_kf) counts[k] = red[0][0] + red[1][0] + red[2][0] + red[3][0];
        if (k == 0) counts[n_keyframes] = red[0][1] + red[1][1] + red[2][1] + red[3][1];
    }
}

extern "C" int eslam_keyframe_overlap(const float* rays_o, const float* rays_d, const float* gt_depth, int n_rays,
                                      int num_samples, const float* c2ws, int n_keyframes, int H, int W, float fx,
                                      float fy, float cx, float cy, int edge, int32_t* counts, eslam_stream_t stream) {
    if (n_rays < 0 || num_samples < 1 || n_keyframes < 0 || H <= 0 || W <= 0) {
        eslam_set_error("eslam_keyframe_overlap: bad sizes (n_rays=%d, num_samples=%d, n_keyframes=%d)", n_rays,
                        num_samples, n_keyframes);
        return 1;
    }
    if (!counts || (n_keyframes > 0 && !c2ws) || (n_rays > 0 && (!rays_o || !rays_d || !gt_depth))) {
        eslam_set_error("eslam_keyframe_overlap: null pointer");
        return 1;
    }
    if ((int64_t)n_rays * num_samples > 0x7fffffff) {
        eslam_set_error("eslam_keyframe_overlap: too many points");
        return 1;
    }
    hipStream_t st = (hipStream_t)stream;
    eslam_prof_begin(PROF_KF_OVERLAP, st);
    hipLaunchKernelGGL(keyframe_overlap_kernel, dim3(n_keyframes > 0 ? n_keyframes : 1), dim3(256), 0, st, rays_o, rays_d,
                       gt_depth, n_rays, num_samples, c2ws, n_keyframes, H, W, fx, fy, cx, cy, edge, counts);
    eslam_prof_end(PROF_KF_OVERLAP, st);
    return eslam_check_launch("keyframe_overlap_kernel");
}
