// Device pieces of the ray-sharded iteration's prologue (eslam_shard.hip; the fused prologue kernel lives in eslam_sample.hip,
// beside the sampler arithmetic it replays): conservative texel marking from ray geometry, and the block list's clear.
#pragma once
#include "eslam_common.h"

struct BlockBase32 { int64_t b[NPL]; };

__device__ __forceinline__ float aabb_exit_plain(const float o[3], const float d[3], const Bound& bnd) {
    // min over axes of max over the two slabs (Renderer.py:114-115); only its magnitude matters here
    float t = 3.4e38f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float t0 = (bnd.lo[k] - o[k]) / d[k], t1 = (bnd.hi[k] - o[k]) / d[k];
        const float m = fmaxf(t0, t1);
        t = (m != m) ? m : fminf(t, m);
        if (t0 != t0 || t1 != t1) t = __builtin_nanf("");
    }
    return t;
}

// One wave per ray.  The samples of a ray lie at parameters z in [t0, t1] of p(z) = o + z d:
//   depth d > 0 : z_free in [0, 1.2 d], z_surf in [d - 1.5 tau, d + 1.5 tau], jitter stays between the first and last sample
//                 (Renderer.py:55-61,96-100)                          -> [min(0, d - 1.5 tau), max(1.2 d, d + 1.5 tau)]
//   depth-less  : uniform samples in [0, far], far = AABB exit + 0.01, importance samples inside their bins (Renderer.py:114-134)
// In a plane the texel coordinate is affine in z, x(z) = ax + bx z (normalisation + align_corners scaling), clamped at the
// border.  The segment is cut into steps of at most one cell along either axis; a step's samples fall into the cells between
// its end points' cells, whose bilinear corners are the box [i_lo, i_hi + 1] x [j_lo, j_hi + 1] (at most 3 x 3 texels), widened
// by MARK_EPS cells against the float32 rounding of the kernels' own coordinate arithmetic.
#define MARK_EPS 0.02f
struct MarkArgs {
    PlaneSet planes;
    Bound bnd;
    const float* rays_o;
    const float* rays_d;
    const float* gt_depth;
    int R;
    float c15;
    BlockBase32 base;
    uint8_t* touched;
};
// block `bid` of `nblocks` blocks of 256 threads
__device__ __forceinline__ void mark_rays_block(const MarkArgs& m, int bid, int nblocks) {
    const PlaneSet& planes = m.planes;
    const Bound& bnd = m.bnd;
    const float* __restrict__ rays_o = m.rays_o;
    const float* __restrict__ rays_d = m.rays_d;
    const float* __restrict__ gt_depth = m.gt_depth;
    const int R = m.R;
    const float c15 = m.c15;
    const BlockBase32& base = m.base;
    uint8_t* __restrict__ touched = m.touched;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int ray = bid * 4 + wave; ray < R; ray += nblocks * 4) {
        const float o[3] = {rays_o[3 * ray], rays_o[3 * ray + 1], rays_o[3 * ray + 2]};
        const float d[3] = {rays_d[3 * ray], rays_d[3 * ray + 1], rays_d[3 * ray + 2]};
        const float gd = gt_depth[ray];
        float t0, t1;
        if (gd > 0.0f) {
            t0 = fminf(0.0f, gd - c15);
            t1 = fmaxf(1.2f * gd, gd + c15);
        } else {
            t0 = 0.0f;
            t1 = aabb_exit_plain(o, d, bnd) + 0.01f;
        }
        const float pad = 1e-5f * (fabsf(t0) + fabsf(t1)) + 1e-6f;
        t0 -= pad; t1 += pad;
        const float len = t1 - t0;
#pragma unroll 1
        for (int pi = 0; pi < NPL; ++pi) {
            const int orient = (pi % 6) >> 1;                   // 0 xy, 1 xz, 2 yz: first coordinate -> width, second -> height
            const int au = orient == 2 ? 1 : 0, av = orient == 0 ? 1 : 2;
            const int pw = planes.p[pi].w, ph = planes.p[pi].h;
            const float su = (float)(pw - 1) / (bnd.hi[au] - bnd.lo[au]), sv = (float)(ph - 1) / (bnd.hi[av] - bnd.lo[av]);
            const float ax = (o[au] - bnd.lo[au]) * su, bx = d[au] * su;
            const float ay = (o[av] - bnd.lo[av]) * sv, by = d[av] * sv;
            const float span = fmaxf(fabsf(bx), fabsf(by)) * len;
            // a ray that is not finite (or absurdly long) is marked in one step: its clamped box, at worst the whole plane
            const int n = (span == span && span < 4096.0f) ? (int)ceilf(span) + 1 : 1;
            const float dt = len / (float)n;
            uint8_t* __restrict__ tp = touched + base.b[pi];
            const float wm1 = (float)(pw - 1), hm1 = (float)(ph - 1);
            for (int k = lane; k < n; k += WAVE) {
                const float ta = t0 + dt * (float)k, tb = (k + 1 == n) ? t1 : ta + dt;
                const float xa = ax + bx * ta, xb = ax + bx * tb, ya = ay + by * ta, yb = ay + by * tb;
                // fmaxf / fminf drop a NaN operand: a NaN coordinate clamps to 0, the cell the kernels' axis_coord gives it too
                const int i_lo = (int)floorf(fminf(fmaxf(fminf(xa, xb) - MARK_EPS, 0.0f), wm1));
                const int i_hi = min((int)floorf(fminf(fmaxf(fmaxf(xa, xb) + MARK_EPS, 0.0f), wm1)) + 1, pw - 1);
                const int j_lo = (int)floorf(fminf(fmaxf(fminf(ya, yb) - MARK_EPS, 0.0f), hm1));
                const int j_hi = min((int)floorf(fminf(fmaxf(fmaxf(ya, yb) + MARK_EPS, 0.0f), hm1)) + 1, ph - 1);
                for (int j = j_lo; j <= j_hi; ++j)
                    for (int i = i_lo; i <= i_hi; ++i) tp[(int64_t)j * pw + i] = 1;
            }
        }
    }
}

// MODE 2 of blocks_move_dev_kernel as a device function: zero the listed 128-byte blocks of flat and the dense tail
struct ClearArgs {
    float* flat;
    const int32_t* idx;
    const int32_t* meta;
    float* tail;
    int n_tail;
};
__device__ __forceinline__ void clear_blocks_block(const ClearArgs& c, int bid, int nblocks) {
    const int n_idx = c.meta[0];
    const int sub = threadIdx.x & 7;
    for (int i = bid * 32 + (threadIdx.x >> 3); i < n_idx; i += nblocks * 32)
        *(float4_t*)(c.flat + (int64_t)c.idx[i] * 32 + sub * 4) = (float4_t){0.f, 0.f, 0.f, 0.f};
    for (int i = bid * 256 + threadIdx.x; i < c.n_tail; i += nblocks * 256) c.tail[i] = 0.0f;
}
