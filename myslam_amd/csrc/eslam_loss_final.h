// Shared pieces of the fused loss: accumulator slots, region rule, and the ticket finalisation that turns per-workgroup
// sums into acc [16] + the loss value without a pre-zeroed buffer (used by loss_reduce_kernel<true> and by the forward
// kernel's loss epilogue).
#pragma once
#include "eslam_common.h"

// accumulator slots (floats)
enum { A_N_FRONT = 0, A_N_CENTER, A_N_TAIL, A_S_FRONT, A_S_CENTER, A_S_TAIL, A_N_DEPTH, A_S_DEPTH, A_S_COLOR, A_N_COLOR, A_COUNT };

struct LossW { float fs, center, tail, depth, color; };

// truncation constants as the reference forms them: Python-float products cast to float32 by torch
struct Trunc { float t, t04; };
static inline Trunc make_trunc(double truncation) { return Trunc{(float)truncation, (float)(0.4 * truncation)}; }

__device__ __forceinline__ int sdf_region(float z, float d, Trunc tr) {
    // Mapper.py:124-134: 0 front, 1 center, 2 tail, 3 back (no loss)
    const bool front = z < (d - tr.t);
    const bool back = z > (d + tr.t);
    const bool center = (z > (d - tr.t04)) && (z < (d + tr.t04));
    if (front) return 0;
    if (back) return 3;
    if (center) return 1;
    return 2;
}

// Upstream gradients of the loss for one ray / one sample, shared by loss_grad_kernel and by the backward kernel that forms
// them inline (eslam_render_bwd_loss): same expressions, so both paths give the same bits.
struct LossGradIn {
    const float* gt_depth;       // [R]
    const float* gt_color;       // [R,3]
    const uint8_t* ray_mask;     // [R] or NULL
    const float* depth;          // [R]   forward outputs
    const float* rgb;            // [R,3]
    const float* acc;            // [ESLAM_LOSS_ACC] set sizes and sums (global, i.e. after any all-reduce)
    const float* upstream;       // [1] d L / d loss, or NULL (= 1)
    float* loss_out;             // [1] or NULL: the loss value formed from acc
    Trunc tr;
    LossW w;
};

__device__ __forceinline__ LossW loss_scaled_weights(const LossW w_in, const float* __restrict__ upstream) {
    LossW w = w_in;
    if (upstream) {
        const float u = upstream[0];
        w.fs *= u; w.center *= u; w.tail *= u; w.depth *= u; w.color *= u;
    }
    return w;
}

struct LossK { float kf, kc, kt; };
__device__ __forceinline__ LossK loss_sdf_factors(const LossW w, const Trunc tr, const float* __restrict__ acc) {
    return LossK{2.0f * w.fs / acc[A_N_FRONT], 2.0f * w.center * tr.t / acc[A_N_CENTER], 2.0f * w.tail * tr.t / acc[A_N_TAIL]};
}

// d loss / d sdf of one sample of a ray with depth d (m: the ray takes part in the SDF terms)
__device__ __forceinline__ float loss_g_sdf(bool m, float z, float sd, float d, const Trunc tr, const LossK k) {
    float g = 0.0f;
    if (m) {
        const int reg = sdf_region(z, d, tr);
        if (reg == 0) g = k.kf * (sd - 1.0f);
        else if (reg == 1) g = k.kc * ((z + sd * tr.t) - d);
        else if (reg == 2) g = k.kt * ((z + sd * tr.t) - d);
    }
    return g;
}
__device__ __forceinline__ float loss_g_depth(bool m, float d, float depth, const LossW w, float nd) {
    return m ? -2.0f * w.depth * (d - depth) / nd : 0.0f;
}
__device__ __forceinline__ float loss_g_color(bool mc, float gt, float c, const LossW w, float ncol) {
    return mc ? -2.0f * w.color * (gt - c) / ncol : 0.0f;
}
__device__ __forceinline__ float loss_value_from_acc(const LossW w, const float* __restrict__ acc) {
    // torch.mean over an empty set is NaN (0/0); keep that behaviour
    return w.fs * (acc[A_S_FRONT] / acc[A_N_FRONT]) + w.center * (acc[A_S_CENTER] / acc[A_N_CENTER]) +
           w.tail * (acc[A_S_TAIL] / acc[A_N_TAIL]) + w.color * (acc[A_S_COLOR] / acc[A_N_COLOR]) +
           w.depth * (acc[A_S_DEPTH] / acc[A_N_DEPTH]);
}

// Deterministic variant (ESLAM_DETERMINISTIC=1): the float atomics above add the workgroups' sums in arrival order, so the
// last bits of acc - and of everything scaled by it - change from run to run.  Here every workgroup stores its sums into a
// slot of its own (write-through stores, drained before the ticket is drawn: MI355X_MICROARCH.md, "Valid forms"), and the
// workgroup that draws the last ticket adds the slots in a fixed order.  scratch: [0] ticket, slots from float 32 on, 16
// floats per workgroup (eslam_loss_scratch_floats).
__device__ __forceinline__ void loss_finalize_det(float tot, float* __restrict__ scratch, float* __restrict__ acc,
                                                  const LossW w, float* __restrict__ loss) {
    __shared__ unsigned ticket_d;
    __shared__ float part[4][16];
    __shared__ float fin_d[16];
    float* slots = scratch + 32;
    if (threadIdx.x < 16)
        __hip_atomic_store(slots + (size_t)blockIdx.x * 16 + threadIdx.x, threadIdx.x < A_COUNT ? tot : 0.0f, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) ticket_d = atomicAdd((unsigned*)scratch, 1u);
    __syncthreads();
    if (ticket_d != gridDim.x - 1) return;
    // thread t adds slots t, t + 256, ... of every accumulator; then a fixed shuffle tree and the four waves in order
    float p[A_COUNT];
#pragma unroll
    for (int k = 0; k < A_COUNT; ++k) p[k] = 0.0f;
    for (unsigned s = threadIdx.x; s < gridDim.x; s += blockDim.x) {
#pragma unroll
        for (int k = 0; k < A_COUNT; ++k)
            p[k] += __hip_atomic_load(slots + (size_t)s * 16 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < A_COUNT; ++k) {
        const float t = wave_sum(p[k]);
        if (lane == 0 && wave < 4) part[wave][k] = t;
    }
    __syncthreads();
    if (threadIdx.x < 16) {
        float v = 0.0f;
        if (threadIdx.x < A_COUNT) {
            const int nw = (blockDim.x + 63) >> 6;
            for (int q = 0; q < nw && q < 4; ++q) v += part[q][threadIdx.x];
        }
        fin_d[threadIdx.x] = v;
        acc[threadIdx.x] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (loss)
            loss[0] = w.fs * (fin_d[A_S_FRONT] / fin_d[A_N_FRONT]) + w.center * (fin_d[A_S_CENTER] / fin_d[A_N_CENTER]) +
                      w.tail * (fin_d[A_S_TAIL] / fin_d[A_N_TAIL]) + w.color * (fin_d[A_S_COLOR] / fin_d[A_N_COLOR]) +
                      w.depth * (fin_d[A_S_DEPTH] / fin_d[A_N_DEPTH]);
        atomicExch((unsigned*)scratch, 0u);
    }
}

// Called by ALL threads of a workgroup (>= 64 threads); `tot` = this workgroup's sum of accumulator threadIdx.x (threads
// 0 .. A_COUNT-1).  scratch: [0] ticket counter (unsigned); accumulator k at scratch[32 * (k + 1)] - one 128-B line each,
// so the float atomics of different accumulators go to different memory channels.  Everything is exchanged through
// device-scope atomics (performed at the memory side, the one point all XCDs agree on): a release fence instead would
// write back each XCD's L2 - full of freshly written features at this point - and cost 20 us.  Every adder waits for its
// atomic's return value before the workgroup takes its ticket, so the workgroup that draws the last ticket sees all sums;
// it swaps them out for zeros, which leaves the scratch ready for the next call.
__device__ __forceinline__ void loss_finalize(float tot, float* __restrict__ scratch, float* __restrict__ acc,
                                              const LossW w, float* __restrict__ loss) {
    __shared__ unsigned ticket;
    __shared__ float fin[16];
    if (threadIdx.x < A_COUNT && tot != 0.0f) {
        const float old = atomicAdd(scratch + 32 * (threadIdx.x + 1), tot);
        asm volatile("" ::"v"(old));                    // the add has been performed once its result is back
    }
    __syncthreads();
    if (threadIdx.x == 0) ticket = atomicAdd((unsigned*)scratch, 1u);
    __syncthreads();
    if (ticket != gridDim.x - 1) return;
    if (threadIdx.x < 16) {
        const float v = threadIdx.x < A_COUNT ? atomicExch(scratch + 32 * (threadIdx.x + 1), 0.0f) : 0.0f;
        fin[threadIdx.x] = v;
        acc[threadIdx.x] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (loss)      // torch.mean over an empty set is NaN (0/0); keep that behaviour
            loss[0] = w.fs * (fin[A_S_FRONT] / fin[A_N_FRONT]) + w.center * (fin[A_S_CENTER] / fin[A_N_CENTER]) +
                      w.tail * (fin[A_S_TAIL] / fin[A_N_TAIL]) + w.color * (fin[A_S_COLOR] / fin[A_N_COLOR]) +
                      w.depth * (fin[A_S_DEPTH] / fin[A_N_DEPTH]);
        atomicExch((unsigned*)scratch, 0u);
    }
}
