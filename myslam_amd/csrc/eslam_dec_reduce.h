// Reduction of the decoder-gradient slabs the decoder backward leaves behind (one row per workgroup) into the flat decoder
// gradient, as a device function: run by dec_grad_reduce_kernel, or by the first workgroups of the scatter's grid.
#pragma once
#include "eslam_common.h"

#define SLAB 1364          // floats per decoder per wave slab (rgb decoder needs 1363)
// offsets inside a per-decoder slab
#define SL_W1 0
#define SL_B1 1024
#define SL_W2 1040
#define SL_B2 1296
#define SL_W3 1312         // [nout][16], nout <= 3
#define SL_B3 1360         // [nout]


struct DecReduceArgs {
    const float* slabs;          // [nrows][2][SLAB]
    int nrows;
    float* g_dec;                // [ESLAM_N_DEC_PARAMS], order of eslam_decoders_t
    const float* beta_parts;     // [n_beta_parts] partial sums of g_beta, or NULL
    int n_beta_parts;
    float* g_beta;               // [1] or NULL
};
#define DEC_RED_COLBLOCKS ((SLAB + 63) / 64)         // 22 blocks of 64 columns per decoder

// One block of 64 slab columns of decoder d, summed over all rows by NT threads = 64 columns x NT/64 row groups; every
// thread keeps 8 independent loads in flight (a first version walked 512 rows with one load outstanding: 0.19 ms for 22 MB).
// red: NT floats of LDS.  The last column block of decoder 0 has only 20 live columns: it also sums the g_beta partials.
template <int NT>
__device__ __forceinline__ void dec_grad_reduce_block(const DecReduceArgs a, int colblock, int d, float* red) {
    constexpr int PARTS = NT / 64;
    const int tid = threadIdx.x;
    if (colblock == DEC_RED_COLBLOCKS - 1 && d == 0 && a.beta_parts && a.g_beta) {
        float s = 0.f;
        for (int i = tid; i < a.n_beta_parts; i += NT) s += a.beta_parts[i];
        s = wave_sum(s);
        if ((tid & 63) == 0) red[tid >> 6] = s;
        __syncthreads();
        if (tid == 0) {
            float t = 0.f;
            for (int k = 0; k < PARTS; ++k) t += red[k];
            a.g_beta[0] = t;
        }
        __syncthreads();
    }
    const int cl = tid & 63;
    const int col = colblock * 64 + cl;
    const int part = tid >> 6;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (col < SLAB) {
        const float* src = a.slabs + (int64_t)d * SLAB + col;
        int row = part;
        for (; row + 7 * PARTS < a.nrows; row += 8 * PARTS) {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u] += src[(int64_t)(row + u * PARTS) * 2 * SLAB];
        }
        for (; row < a.nrows; row += PARTS) acc[0] += src[(int64_t)row * 2 * SLAB];
    }
    red[part * 64 + cl] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    __syncthreads();
    if (part == 0 && col < SLAB) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < PARTS; ++k) v += red[k * 64 + cl];
        const int nout = d ? 3 : 1;
        // slab offset -> flat offset inside the decoder's parameter block
        int dst = -1;
        if (col < SL_W3) dst = col;                                   // W1,b1,W2,b2 are laid out identically
        else if (col < SL_W3 + nout * 16) dst = 1312 + (col - SL_W3);
        else if (col >= SL_B3 && col < SL_B3 + nout) dst = 1312 + nout * 16 + (col - SL_B3);
        if (dst >= 0) a.g_dec[(d ? 1329 : 0) + dst] = v;
    }
}
