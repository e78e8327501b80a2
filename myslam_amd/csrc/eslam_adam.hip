// Multi-tensor Adam step for the mapper's / tracker's parameter groups.
// Restates torch.optim.Adam as the reference constructs it (src/Mapper.py:291-299, src/Tracker.py:262-266: default
// betas/eps, no weight decay, no amsgrad) and steps it (src/Mapper.py:348-350, src/Tracker.py:206-208):
//     exp_avg    = exp_avg + (1-b1) * (grad - exp_avg)                    (lerp_)
//     exp_avg_sq = exp_avg_sq * b2 + ((1-b2) * grad) * grad               (mul_, addcmul_)
//     denom      = sqrt(exp_avg_sq) / sqrt(1 - b2^t) + eps
//     param      = param + ((-lr / (1 - b1^t)) * exp_avg) / denom         (addcdiv_)
// with the scalars formed in double and rounded to float32 once, as torch does for Python-float operands.
//
// One launch covers all tensors of a step (12 planes + 12 decoder tensors + beta + poses = 26): the table of
// pointers travels in the kernel arguments, each workgroup owns one 4096-element chunk of one tensor.
// The kernel is a pure stream over 6.8 M elements (room0): 16 B read + 12 B written per element when dense.
// Elements with grad == exp_avg == exp_avg_sq == 0 are left untouched - for those the update is exactly
// param + (-step*0)/eps = param, so skipping the three stores is bit-identical to the dense step; the mapper
// builds a fresh optimiser for every mapped frame (Mapper.py:291), so every texel its rays have not visited since
// then is in that state (4-25 % of the texels are touched per frame).
// With zero_grad the consumed gradient is cleared in the same pass (optimizer.zero_grad(), Mapper.py:348), which
// replaces the 27-70 MB fill the next backward would otherwise need.
#include <math.h>
#include "eslam_common.h"

#define ADAM_CHUNK 4096          // elements per workgroup: 256 threads x 4 float4

struct AdamTable {
    eslam_adam_tensor_t t[ESLAM_ADAM_MAX_TENSORS];
    int32_t first_block[ESLAM_ADAM_MAX_TENSORS + 1];
};

struct AdamScalars { float w1, b2, omb2, bc2s, eps; double bc1; };

__device__ __forceinline__ AdamScalars adam_scalars(int step, double beta1, double beta2, double eps) {
    AdamScalars s;
    s.w1 = (float)(1.0 - beta1);
    s.b2 = (float)beta2;
    s.omb2 = (float)(1.0 - beta2);
    s.bc1 = 1.0 - pow(beta1, (double)step);
    s.bc2s = (float)sqrt(1.0 - pow(beta2, (double)step));
    s.eps = (float)eps;
    return s;
}

__device__ __forceinline__ void adam_elem(float& p, float g, float& m, float& v, const AdamScalars& s, float nss) {
    m = m + s.w1 * (g - m);
    v = v * s.b2 + (s.omb2 * g) * g;
    const float denom = sqrtf(v) / s.bc2s + s.eps;
    p = p + (nss * m) / denom;
}

__global__ void adam_tick_kernel(int32_t* step_dev) { step_dev[0] += 1; }

__global__ __launch_bounds__(256) void adam_step_kernel(const AdamTable tab, int n_tensors,
                                                        const int32_t* __restrict__ step_dev, int step_host,
                                                        double beta1, double beta2, double eps, int zero_grad) {
    int k = 0;
    for (int i = 1; i < n_tensors; ++i)
        if ((int)blockIdx.x >= tab.first_block[i]) k = i;          // uniform: block -> tensor
    const eslam_adam_tensor_t T = tab.t[k];
    const int64_t e0 = (int64_t)(blockIdx.x - tab.first_block[k]) * ADAM_CHUNK;
    const int64_t e1 = e0 + ADAM_CHUNK < T.n ? e0 + ADAM_CHUNK : T.n;
    const int step = step_dev ? step_dev[0] : step_host;
    const AdamScalars s = adam_scalars(step, beta1, beta2, eps);
    const float nss = (float)(-(T.lr / s.bc1));

    // float4 path when the tensor's four arrays are 16-byte aligned (uniform per workgroup); slices of a flat buffer
    // that start at odd offsets (the colour decoder inside the 2692-float decoder gradient) take the scalar path
    const uintptr_t bits = (uintptr_t)T.param | (uintptr_t)T.grad | (uintptr_t)T.exp_avg | (uintptr_t)T.exp_avg_sq;
    if ((bits & 15) == 0) {
        const int64_t v1 = e0 + ((e1 - e0) & ~(int64_t)3);
        for (int64_t i = e0 + 4 * (int64_t)threadIdx.x; i < v1; i += 4 * 256) {
            const float4_t g = *(const float4_t*)(T.grad + i);
            float4_t m = *(const float4_t*)(T.exp_avg + i);
            float4_t v = *(const float4_t*)(T.exp_avg_sq + i);
            bool idle = true;
#pragma unroll
            for (int c = 0; c < 4; ++c) idle = idle && g[c] == 0.0f && m[c] == 0.0f && v[c] == 0.0f;
            if (idle) continue;
            float4_t p = *(const float4_t*)(T.param + i);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float pc = p[c], mc = m[c], vc = v[c];
                adam_elem(pc, g[c], mc, vc, s, nss);
                p[c] = pc; m[c] = mc; v[c] = vc;
            }
            *(float4_t*)(T.param + i) = p;
            *(float4_t*)(T.exp_avg + i) = m;
            *(float4_t*)(T.exp_avg_sq + i) = v;
            if (zero_grad) *(float4_t*)(T.grad + i) = (float4_t){0.f, 0.f, 0.f, 0.f};
        }
        // ragged tail of the tensor (n % 4 elements, only in its last chunk)
        const int64_t i = v1 + threadIdx.x;
        if (i < e1) {
            const float g = T.grad[i];
            float m = T.exp_avg[i], v = T.exp_avg_sq[i];
            if (!(g == 0.0f && m == 0.0f && v == 0.0f)) {
                float p = T.param[i];
                adam_elem(p, g, m, v, s, nss);
                T.param[i] = p; T.exp_avg[i] = m; T.exp_avg_sq[i] = v;
                if (zero_grad) T.grad[i] = 0.0f;
            }
        }
    } else {
        for (int64_t i = e0 + threadIdx.x; i < e1; i += 256) {
            const float g = T.grad[i];
            float m = T.exp_avg[i], v = T.exp_avg_sq[i];
            if (g == 0.0f && m == 0.0f && v == 0.0f) continue;
            float p = T.param[i];
            adam_elem(p, g, m, v, s, nss);
            T.param[i] = p; T.exp_avg[i] = m; T.exp_avg_sq[i] = v;
            if (zero_grad) T.grad[i] = 0.0f;
        }
    }
}

extern "C" int eslam_adam_step(const eslam_adam_tensor_t* tensors_host, int n_tensors, int step, int32_t* step_dev,
                               double beta1, double beta2, double eps, int zero_grad, eslam_stream_t stream) {
    if (!tensors_host || n_tensors < 0) {
        eslam_set_error("eslam_adam_step: null tensor table");
        return 1;
    }
    if (!step_dev && step < 1) {
        eslam_set_error("eslam_adam_step: step must be >= 1 (got %d)", step);
        return 1;
    }
    if (!(beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0 && eps >= 0.0)) {
        eslam_set_error("eslam_adam_step: betas must lie in [0,1) and eps be >= 0");
        return 1;
    }
    for (int i = 0; i < n_tensors; ++i) {
        const eslam_adam_tensor_t& t = tensors_host[i];
        if (t.n < 0 || (t.n > 0 && (!t.param || !t.grad || !t.exp_avg || !t.exp_avg_sq))) {
            eslam_set_error("eslam_adam_step: tensor %d has a null pointer or a negative size", i);
            return 1;
        }
    }
    hipStream_t st = (hipStream_t)stream;
    if (step_dev) hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, st, step_dev);
    eslam_prof_begin(PROF_ADAM, st);
    for (int base = 0; base < n_tensors; base += ESLAM_ADAM_MAX_TENSORS) {
        const int cnt = n_tensors - base < ESLAM_ADAM_MAX_TENSORS ? n_tensors - base : ESLAM_ADAM_MAX_TENSORS;
        AdamTable tab;
        int64_t blocks = 0;
        for (int i = 0; i < cnt; ++i) {
            tab.t[i] = tensors_host[base + i];
            tab.first_block[i] = (int32_t)blocks;
            blocks += (tab.t[i].n + ADAM_CHUNK - 1) / ADAM_CHUNK;
        }
        tab.first_block[cnt] = (int32_t)blocks;
        if (blocks > 0x7fffffff) {
            eslam_set_error("eslam_adam_step: too many elements for one launch");
            return 1;
        }
        if (blocks == 0) continue;
        hipLaunchKernelGGL(adam_step_kernel, dim3((unsigned)blocks), dim3(256), 0, st, tab, cnt, step_dev, step, beta1,
                           beta2, eps, zero_grad);
    }
    eslam_prof_end(PROF_ADAM, st);
    return eslam_check_launch("adam_step_kernel");
}
