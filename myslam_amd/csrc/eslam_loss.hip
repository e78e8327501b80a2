// Fused caller-side loss (value + upstream gradients).  Restates reference src/Mapper.py:110-144,337-346 and
// src/Tracker.py:114-148,197-204 without boolean-mask indexing (each of which costs the PyTorch formulation a
// nonzero() + host sync).  Two launches: (1) per-ray partial sums and set sizes -> 9 global accumulators,
// (2) gradients scaled by the now-known set sizes, and the loss value.
#include "eslam_loss_final.h"

// FINAL = false: the sums are added to acc with one atomic per workgroup and accumulator (acc pre-zeroed by the caller; the
//   data-parallel path all-reduces acc before anything is derived from it).
// FINAL = true: single-GPU path, one launch and no buffer to pre-zero per call: every workgroup adds its sums to a
//   persistent scratch and takes a ticket; the workgroup that draws the last ticket moves the totals to acc [16], writes
//   the loss value and leaves the scratch zeroed for the next call.
template <bool FINAL>
__global__ __launch_bounds__(256) void loss_reduce_kernel(const float* __restrict__ depth, const float* __restrict__ rgb,
                                                          const float* __restrict__ sdf, const float* __restrict__ z_vals,
                                                          const float* __restrict__ gt_depth,
                                                          const float* __restrict__ gt_color,
                                                          const uint8_t* __restrict__ ray_mask, int R, int S, const Trunc tr,
                                                          float* __restrict__ acc, float* __restrict__ scratch,
                                                          const LossW w, float* __restrict__ loss, const int det) {
    // grid-stride over rays: the 10 accumulators share one cache line, and same-line float atomics serialise at the
    // memory side (~12 ns each), so the number of workgroups - not of rays - sets the cost of the final adds
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float v[A_COUNT];
#pragma unroll
    for (int k = 0; k < A_COUNT; ++k) v[k] = 0.0f;
    for (int ray = blockIdx.x * 4 + wave; ray < R; ray += gridDim.x * 4) {
        const float d = gt_depth[ray];
        const bool mc = ray_mask ? (ray_mask[ray] != 0) : true;      // rays of the batch (colour term)
        const bool m = mc && d > 0.0f;                                // ... that have depth (SDF and depth terms)
        if (m) {
            for (int s = lane; s < S; s += WAVE) {
                const float z = z_vals[(int64_t)ray * S + s];
                const float sd = sdf[(int64_t)ray * S + s];
                const int reg = sdf_region(z, d, tr);
                if (reg == 0) { v[A_N_FRONT] += 1.0f; const float e = sd - 1.0f; v[A_S_FRONT] += e * e; }
                else if (reg == 1) { v[A_N_CENTER] += 1.0f; const float e = (z + sd * tr.t) - d; v[A_S_CENTER] += e * e; }
                else if (reg == 2) { v[A_N_TAIL] += 1.0f; const float e = (z + sd * tr.t) - d; v[A_S_TAIL] += e * e; }
            }
            if (lane == 0) { const float e = d - depth[ray]; v[A_N_DEPTH] += 1.0f; v[A_S_DEPTH] += e * e; }
        }
        if (mc && lane < 3) { const float e = gt_color[3 * ray + lane] - rgb[3 * ray + lane]; v[A_S_COLOR] += e * e; v[A_N_COLOR] += 1.0f; }
    }
    __shared__ float red[4][A_COUNT];
#pragma unroll
    for (int k = 0; k < A_COUNT; ++k) {
        const float s = wave_sum(v[k]);
        if (lane == 0) red[wave][k] = s;
    }
    __syncthreads();
    float tot = 0.0f;
    if (threadIdx.x < A_COUNT)
        tot = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    if (!FINAL) {
        if (threadIdx.x < A_COUNT && tot != 0.0f) atomicAdd(acc + threadIdx.x, tot);
        return;
    }
    if (det) loss_finalize_det(tot, scratch, acc, w, loss);
    else loss_finalize(tot, scratch, acc, w, loss);
}

__global__ __launch_bounds__(256) void loss_grad_kernel(const float* __restrict__ depth, const float* __restrict__ rgb,
                                                        const float* __restrict__ sdf, const float* __restrict__ z_vals,
                                                        const float* __restrict__ gt_depth,
                                                        const float* __restrict__ gt_color,
                                                        const uint8_t* __restrict__ ray_mask, int R, int S, const Trunc tr,
                                                        const LossW w_in, const float* __restrict__ acc,
                                                        float* __restrict__ loss, float* __restrict__ g_depth,
                                                        float* __restrict__ g_rgb, float* __restrict__ g_sdf,
                                                        const float* __restrict__ upstream) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ray = blockIdx.x * 4 + wave;
    const float nd = acc[A_N_DEPTH], ncol = acc[A_N_COLOR];
    if (blockIdx.x == 0 && threadIdx.x == 0 && loss) loss[0] = loss_value_from_acc(w_in, acc);
    if (ray >= R || !g_sdf) return;
    const LossW w = loss_scaled_weights(w_in, upstream);
    const float d = gt_depth[ray];
    const bool mc = ray_mask ? (ray_mask[ray] != 0) : true;
    const bool m = mc && d > 0.0f;
    const LossK k = loss_sdf_factors(w, tr, acc);
    for (int s = lane; s < S; s += WAVE) {
        float z = 0.0f, sd = 0.0f;
        if (m) { z = z_vals[(int64_t)ray * S + s]; sd = sdf[(int64_t)ray * S + s]; }
        g_sdf[(int64_t)ray * S + s] = loss_g_sdf(m, z, sd, d, tr, k);
    }
    if (lane == 0) g_depth[ray] = loss_g_depth(m, d, depth[ray], w, nd);
    if (lane < 3) g_rgb[3 * ray + lane] = loss_g_color(mc, gt_color[3 * ray + lane], rgb[3 * ray + lane], w, ncol);
}

static int loss_args_ok(const char* who, const float* depth, const float* rgb, const float* sdf, const float* z_vals,
                        const float* gt_depth, const float* gt_color, int R, int S) {
    if (R <= 0 || S <= 0) {
        eslam_set_error("%s: empty batch", who);
        return 0;
    }
    if (!depth || !rgb || !sdf || !z_vals || !gt_depth || !gt_color) {
        eslam_set_error("%s: null argument", who);
        return 0;
    }
    return 1;
}

extern "C" int eslam_loss_reduce(const float* depth, const float* rgb, const float* sdf, const float* z_vals,
                                 const float* gt_depth, const float* gt_color, int R, int S, double truncation,
                                 const uint8_t* ray_mask, float* acc, eslam_stream_t stream) {
    if (!loss_args_ok("eslam_loss_reduce", depth, rgb, sdf, z_vals, gt_depth, gt_color, R, S)) return 1;
    if (!acc) {
        eslam_set_error("eslam_loss_reduce: null accumulator");
        return 1;
    }
    const int nwg = (R + 3) / 4 < 256 ? (R + 3) / 4 : 256;
    hipLaunchKernelGGL(loss_reduce_kernel<false>, dim3(nwg), dim3(256), 0, (hipStream_t)stream, depth, rgb, sdf, z_vals,
                       gt_depth, gt_color, ray_mask, R, S, make_trunc(truncation), acc, (float*)nullptr, LossW{}, (float*)nullptr, 0);
    return eslam_check_launch("loss_reduce_kernel");
}

extern "C" int eslam_loss_value(const float* depth, const float* rgb, const float* sdf, const float* z_vals,
                                const float* gt_depth, const float* gt_color, int R, int S, double truncation,
                                const float* weights5_host, const uint8_t* ray_mask, float* scratch, float* acc,
                                float* loss, eslam_stream_t stream) {
    if (!loss_args_ok("eslam_loss_value", depth, rgb, sdf, z_vals, gt_depth, gt_color, R, S)) return 1;
    if (!weights5_host || !scratch || !acc) {
        eslam_set_error("eslam_loss_value: null argument");
        return 1;
    }
    const LossW w = {weights5_host[0], weights5_host[1], weights5_host[2], weights5_host[3], weights5_host[4]};
    const int nwg = (R + 3) / 4 < 256 ? (R + 3) / 4 : 256;
    eslam_prof_begin(PROF_LOSS, (hipStream_t)stream);
    hipLaunchKernelGGL(loss_reduce_kernel<true>, dim3(nwg), dim3(256), 0, (hipStream_t)stream, depth, rgb, sdf, z_vals,
                       gt_depth, gt_color, ray_mask, R, S, make_trunc(truncation), acc, scratch, w, loss, eslam_deterministic());
    eslam_prof_end(PROF_LOSS, (hipStream_t)stream);
    return eslam_check_launch("loss_reduce_kernel<final>");
}

extern "C" int eslam_loss_grad(const float* depth, const float* rgb, const float* sdf, const float* z_vals,
                               const float* gt_depth, const float* gt_color, int R, int S, double truncation,
                               const float* weights5_host, const uint8_t* ray_mask, const float* acc, float* loss,
                               float* g_depth, float* g_rgb, float* g_sdf, const float* upstream,
                               eslam_stream_t stream) {
    if (!loss_args_ok("eslam_loss_grad", depth, rgb, sdf, z_vals, gt_depth, gt_color, R, S)) return 1;
    if (!weights5_host || !acc || (!loss && !g_sdf) || ((g_depth == nullptr) != (g_sdf == nullptr)) ||
        ((g_rgb == nullptr) != (g_sdf == nullptr))) {
        eslam_set_error("eslam_loss_grad: null argument (give loss and/or all three gradient buffers)");
        return 1;
    }
    const LossW w = {weights5_host[0], weights5_host[1], weights5_host[2], weights5_host[3], weights5_host[4]};
    hipLaunchKernelGGL(loss_grad_kernel, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, depth, rgb, sdf, z_vals,
                       gt_depth, gt_color, ray_mask, R, S, make_trunc(truncation), w, acc, loss, g_depth, g_rgb, g_sdf,
                       upstream);
    return eslam_check_launch("loss_grad_kernel");
}

extern "C" int eslam_mapping_loss(const float* depth, const float* rgb, const float* sdf, const float* z_vals,
                                  const float* gt_depth, const float* gt_color, int R, int S, double truncation,
                                  const float* weights5_host, const uint8_t* ray_mask, float* loss,
                                  float* g_depth, float* g_rgb, float* g_sdf, void* scratch, eslam_stream_t stream) {
    if (!scratch) {
        eslam_set_error("eslam_mapping_loss: null scratch");
        return 1;
    }
    if (hipMemsetAsync(scratch, 0, 64, (hipStream_t)stream) != hipSuccess) {
        eslam_set_error("eslam_mapping_loss: memset failed");
        return 2;
    }
    eslam_prof_begin(PROF_LOSS, (hipStream_t)stream);
    if (int rc = eslam_loss_reduce(depth, rgb, sdf, z_vals, gt_depth, gt_color, R, S, truncation, ray_mask,
                                   (float*)scratch, stream))
        return rc;
    const int rc = eslam_loss_grad(depth, rgb, sdf, z_vals, gt_depth, gt_color, R, S, truncation, weights5_host,
                                   ray_mask, (const float*)scratch, loss, g_depth, g_rgb, g_sdf, nullptr, stream);
    eslam_prof_end(PROF_LOSS, (hipStream_t)stream);
    return rc;
}
