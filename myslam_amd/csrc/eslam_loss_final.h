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
