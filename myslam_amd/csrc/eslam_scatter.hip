// Plane-gradient scatter (autograd of the 12 grid_sample calls of reference src/networks/decoders.py:79-81):
// spatially ordered ray bundles + a per-workgroup sort of the bundle's samples by texel cell + register run-merge.
//
// Why: on the bench workload (room0, 4096 x 64) the 262144 samples make 12.6 M texel contributions but touch only
// ~36 k distinct texels (4.6 MB of the 27 MB of planes), because all rays of a frame leave from one camera centre.
// Global float atomics run at ~1.3 TB/s of added bytes chip-wide (MI355X_MICROARCH.md, "Global float atomics"), so
// the cost of the scatter is the number of 256-B atomic wave-instructions that reach memory (tools/sim_scatter.py):
//      unmerged                                             6.29 M  -> 1.24 ms at the atomic rate
//      v1: run-merge along each ray (scatter_kernel)         2.19 M  -> 0.43 ms   (measured 0.81 ms: hot texels contend)
//      bundles of ~16 rays of similar direction, merged      ~0.3 M  -> 0.06 ms
// An LDS hash table of texel accumulators fed with ds_add_f32 reached that atomic count but ran 2.2 ms: one dependent
// global load + two LDS float atomics per sample and plane, at the 1-2 workgroups per CU the tables leave room for.
// What is built instead needs no LDS atomics at all:
//
// ray_order_kernel    sorts rays by a Morton key of their direction (and origin cell): neighbours in the order are rays
//                     through neighbouring pixels of one camera, which cross the same texels for most of their length.
// scatter_sort_kernel one workgroup = one bundle of consecutive rays in that order x one plane.  It (1) computes the
//                     bilinear cell of each of the bundle's <= 1024 samples, (2) sorts the samples by cell in LDS
//                     (bitonic network), (3) gives each wave a contiguous quarter of the sorted list to walk with
//                     lane = (x-corner, channel): consecutive entries of one cell are summed in two registers (rows
//                     y0, y1) and each cell is written once with two 256-B-shaped float atomics.
#include <stdlib.h>
#include "eslam_decode_tile.h"

#define SORT_MAX 8192

// ---------------------------------------------------------------------------------------------------------
// ray ordering
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned spread3(unsigned v) {      // 10 bits -> every third bit
    v &= 0x3FFu;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

// One workgroup orders one chunk of up to SORT_MAX rays with a single-pass counting sort in LDS:
// key = 15-bit Morton code of the point one metre along the ray, quantised inside the chunk's bounding box of such
// points (rays of one camera: a patch of the unit sphere round its centre; several cameras: several patches, and
// rays of nearby cameras with similar directions end up adjacent, which is what shares texels).  Order inside a cell
// is arbitrary (atomic tickets).  perm[chunk*SORT_MAX + i] = ray id.  ~5 us for 4096 rays (a 78-stage bitonic
// network in one workgroup took 62 us).
#define ORD_BITS 5
#define ORD_CELLS (1 << (3 * ORD_BITS))      // 32768 counters = 128 KB of LDS
#define ORD_PER_THREAD (SORT_MAX / 1024)
__global__ __launch_bounds__(1024) void ray_order_kernel(const float* __restrict__ rays_o,
                                                         const float* __restrict__ rays_d, int R,
                                                         int* __restrict__ perm) {
    extern __shared__ __attribute__((aligned(16))) unsigned hist[];       // [ORD_CELLS]
    __shared__ float red[16][6];
    __shared__ unsigned wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int base = blockIdx.x * SORT_MAX;
    const int n = min(SORT_MAX, R - base);

    float px[ORD_PER_THREAD], py[ORD_PER_THREAD], pz[ORD_PER_THREAD];
    float lo[3] = {3.4e38f, 3.4e38f, 3.4e38f}, hi[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
#pragma unroll
    for (int k = 0; k < ORD_PER_THREAD; ++k) {
        const int i = tid + k * 1024;
        px[k] = py[k] = pz[k] = 0.f;
        if (i < n) {
            const int ray = base + i;
            const float dx = rays_d[3 * ray], dy = rays_d[3 * ray + 1], dz = rays_d[3 * ray + 2];
            const float inv = rsqrtf(fmaxf(dx * dx + dy * dy + dz * dz, 1e-20f));
            px[k] = rays_o[3 * ray] + dx * inv;
            py[k] = rays_o[3 * ray + 1] + dy * inv;
            pz[k] = rays_o[3 * ray + 2] + dz * inv;
            lo[0] = fminf(lo[0], px[k]); hi[0] = fmaxf(hi[0], px[k]);
            lo[1] = fminf(lo[1], py[k]); hi[1] = fmaxf(hi[1], py[k]);
            lo[2] = fminf(lo[2], pz[k]); hi[2] = fmaxf(hi[2], pz[k]);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], m, WAVE));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], m, WAVE));
        }
        if (lane == 0) { red[wave][a] = lo[a]; red[wave][3 + a] = hi[a]; }
    }
    for (int i = tid; i < ORD_CELLS; i += 1024) hist[i] = 0u;
    __syncthreads();
    float scale[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float l = red[0][a], h = red[0][3 + a];
        for (int w = 1; w < 16; ++w) { l = fminf(l, red[w][a]); h = fmaxf(h, red[w][3 + a]); }
        lo[a] = l;
        scale[a] = (float)(1 << ORD_BITS) / fmaxf(h - l, 1e-6f);
    }
    unsigned key[ORD_PER_THREAD], ticket[ORD_PER_THREAD];
#pragma unroll
    for (int k = 0; k < ORD_PER_THREAD; ++k) {
        const int i = tid + k * 1024;
        key[k] = 0; ticket[k] = 0;
        if (i < n) {
            const unsigned qmax = (1u << ORD_BITS) - 1;
            const unsigned qx = min((unsigned)fmaxf((px[k] - lo[0]) * scale[0], 0.f), qmax);
            const unsigned qy = min((unsigned)fmaxf((py[k] - lo[1]) * scale[1], 0.f), qmax);
            const unsigned qz = min((unsigned)fmaxf((pz[k] - lo[2]) * scale[2], 0.f), qmax);
            key[k] = spread3(qx) | (spread3(qy) << 1) | (spread3(qz) << 2);
            ticket[k] = atomicAdd(&hist[key[k]], 1u);
        }
    }
    __syncthreads();
    // exclusive scan of the counters: thread t owns counters [32t, 32t+32)
    const int per = ORD_CELLS / 1024;
    unsigned local = 0;
    for (int j = 0; j < per; ++j) local += hist[tid * per + j];
    unsigned incl = local;
#pragma unroll
    for (int dlt = 1; dlt < WAVE; dlt <<= 1) {
        const unsigned o = __shfl_up(incl, dlt, WAVE);
        if (lane >= dlt) incl += o;
    }
    if (lane == WAVE - 1) wsum[wave] = incl;
    __syncthreads();
    unsigned wbase = 0;
    for (int w = 0; w < wave; ++w) wbase += wsum[w];
    unsigned run = wbase + incl - local;
    for (int j = 0; j < per; ++j) {
        const unsigned cnt = hist[tid * per + j];
        hist[tid * per + j] = run;
        run += cnt;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < ORD_PER_THREAD; ++k) {
        const int i = tid + k * 1024;
        if (i < n) perm[base + hist[key[k]] + ticket[k]] = base + i;
    }
}

// ---------------------------------------------------------------------------------------------------------
// bundle scatter
// ---------------------------------------------------------------------------------------------------------
#define BUNDLE_MAX 1024            // samples per workgroup (power of two, 4 per thread)

template <bool RENDER>
__global__ __launch_bounds__(256) void scatter_sort_kernel(const PlaneSet planes, const Bound bnd,
                                                           const float* __restrict__ rays_o,
                                                           const float* __restrict__ rays_d,
                                                           const float* __restrict__ z_vals,     // RENDER ? [R,S] : pts [N,3]
                                                           const int* __restrict__ perm, int R, int S,
                                                           const float* __restrict__ g_feat, int bundle, int dbg_mode) {
    __shared__ unsigned skey[BUNDLE_MAX];          // (cell << 10) | local sample slot, sorted
    __shared__ unsigned sxy[BUNDLE_MAX];           // per slot: x0 | y0 << 12 | (x1 > x0) << 24 | (y1 > y0) << 25
    __shared__ float swx[BUNDLE_MAX], swy[BUNDLE_MAX];   // per slot: bilinear fractions
    __shared__ int sgrow[BUNDLE_MAX];              // per slot: row of g_feat (global point index)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hx = lane >> 5, c = lane & 31;

    const int pi = blockIdx.y;                               // plane index in all_planes order
    const int d = pi / 6, o = (pi % 6) >> 1, lvl = pi & 1;
    const eslam_plane_t& P = planes.p[pi];
    const int pw = P.w, ph = P.h;
    const int psy = (int)P.stride_y, psx = (int)P.stride_x, psc = (int)P.stride_c;
    float* __restrict__ grad = P.grad;
    const int64_t npts = RENDER ? (int64_t)R * S : (int64_t)R;       // decode mode: R = N points, unit = 64 points
    const int nunits = RENDER ? R : (int)((npts + 63) / 64);
    const int per = RENDER ? S : 64;                                  // samples per unit
    const int u0 = blockIdx.x * bundle;
    const int nu = min(bundle, nunits - u0);
    const int n = nu * per;                                           // <= BUNDLE_MAX by construction of `bundle`

    // (1) cells
    for (int slot = threadIdx.x; slot < BUNDLE_MAX; slot += 256) {
        unsigned key = 0xFFFFFFFFu;
        if (slot < n) {
            const int ui = u0 + slot / per, s = slot % per;
            const int unit = (RENDER && perm) ? perm[ui] : ui;
            const int64_t pt = RENDER ? (int64_t)unit * S + s : (int64_t)unit * 64 + s;
            if (pt < npts) {
                float x, y, z;
                if (RENDER) {
                    const float zz = z_vals[pt];
                    x = rays_o[unit * 3 + 0] + rays_d[unit * 3 + 0] * zz;
                    y = rays_o[unit * 3 + 1] + rays_d[unit * 3 + 1] * zz;
                    z = rays_o[unit * 3 + 2] + rays_d[unit * 3 + 2] * zz;
                } else {
                    x = z_vals[pt * 3 + 0]; y = z_vals[pt * 3 + 1]; z = z_vals[pt * 3 + 2];
                }
                x = norm_coord(x, bnd.lo[0], bnd.hi[0]);
                y = norm_coord(y, bnd.lo[1], bnd.hi[1]);
                z = norm_coord(z, bnd.lo[2], bnd.hi[2]);
                const AxisCoord ax = axis_coord((o == 2) ? y : x, pw);
                const AxisCoord ay = axis_coord((o == 0) ? y : z, ph);
                key = ((unsigned)(ay.i0 * pw + ax.i0) << 10) | (unsigned)slot;
                sxy[slot] = (unsigned)ax.i0 | ((unsigned)ay.i0 << 12) | ((unsigned)(ax.i1 > ax.i0) << 24) |
                            ((unsigned)(ay.i1 > ay.i0) << 25);
                swx[slot] = ax.t;
                swy[slot] = ay.t;
                sgrow[slot] = (int)pt;
            }
        }
        skey[slot] = key;
    }
    __syncthreads();
    // (2) bitonic sort of the keys (invalid slots carry the maximum key and end up last).  Wave w owns elements
    // [256w, 256w+256): every compare-exchange distance j < 256 stays inside one wave's chunk and needs no workgroup
    // barrier (DS operations of a wave execute in order); only the 3 stages with j >= 256 synchronise the workgroup.
    {
        const int wbase = wave * (BUNDLE_MAX / 4);
        for (int k = 2; k <= BUNDLE_MAX; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                if (j >= BUNDLE_MAX / 4) {
                    __syncthreads();
#pragma unroll
                    for (int t = 0; t < BUNDLE_MAX / 512; ++t) {
                        const int p = threadIdx.x + t * 256;                      // pair index
                        const int i = ((p & ~(j - 1)) << 1) | (p & (j - 1));      // lower element of the pair
                        const int l = i | j;
                        const unsigned a = skey[i], b2 = skey[l];
                        if ((a > b2) == ((i & k) == 0)) { skey[i] = b2; skey[l] = a; }
                    }
                    __syncthreads();
                } else {
#pragma unroll
                    for (int t = 0; t < BUNDLE_MAX / 512; ++t) {
                        const int p = lane + t * WAVE;                            // pair index inside the wave's chunk
                        const int i = wbase + (((p & ~(j - 1)) << 1) | (p & (j - 1)));
                        const int l = i | j;
                        const unsigned a = skey[i], b2 = skey[l];
                        if ((a > b2) == ((i & k) == 0)) { skey[i] = b2; skey[l] = a; }
                    }
                    WAVE_SYNC();
                }
            }
        }
    }
    if (dbg_mode == 2) return;      // profiling only (tools/dbg_scatter.py): cost of phases 1+2

    // (3) walk: wave w owns sorted entries [256w, 256w+256), 64 at a time.  Every lane fetches ONE entry's record from
    // LDS (cell, g_feat row, fractions); the walk reads records with v_readlane.  The g_feat values of 32 entries are
    // loaded ahead of the walk of the previous 32: a wave's loads, stores and atomics retire in order on one vmcnt
    // counter, so a load issued BEHIND an atomic would wait for it (~3000 cycles under load); issued ahead of them, the
    // loads only ever wait for other loads.
    const float* __restrict__ gcol = g_feat + d * 64 + lvl * 32 + c;
    const int DUMMY = 0x3FFFFF;                    // cell of padding entries: never flushed
    int cur_cell = -1;
    unsigned cur_xy = 0;
    float acc0 = 0.f, acc1 = 0.f;
    const int e0 = wave * (BUNDLE_MAX / 4);

    auto flush = [&]() {
        if (cur_cell >= 0 && cur_cell != DUMMY) {
            const int x0 = cur_xy & 0xFFF, y0 = (cur_xy >> 12) & 0xFFF;
            const int dxs = ((cur_xy >> 24) & 1) * psx, dys = ((cur_xy >> 25) & 1) * psy;
            float* gp = grad + y0 * psy + x0 * psx + hx * dxs + c * psc;
            if (dbg_mode != 1) {
                atomicAdd(gp, acc0);
                atomicAdd(gp + dys, acc1);
            } else if (acc0 == 1.2345e30f) gp[0] = acc1;      // profiling only: walk without atomics
        }
    };

    struct Rec { int cell, row; unsigned xy; float tx, ty; };
    auto fetch = [&](int blk) {
        Rec r;
        const unsigned k = skey[e0 + blk * WAVE + lane];
        const bool valid = k != 0xFFFFFFFFu;
        const int slot = k & 1023u;
        r.cell = valid ? (int)(k >> 10) : DUMMY;
        r.row = valid ? sgrow[slot] : 0;
        r.xy = valid ? sxy[slot] : 0u;
        r.tx = valid ? swx[slot] : 0.f;
        r.ty = valid ? swy[slot] : 0.f;
        return r;
    };
#ifndef WALK_N
#define WALK_N 8                   // entries per load-ahead group (2 groups in flight: 2*WALK_N VGPRs)
#endif
#define LOAD_HALF(buf, rec, half)                                                             \
    _Pragma("unroll") for (int t = 0; t < WALK_N; ++t) {                                      \
        const int row = __builtin_amdgcn_readlane((rec).row, (half) * WALK_N + t);            \
        buf[t] = gcol[(int64_t)row * 128];                                                    \
    }
#define WALK_HALF(buf, rec, half)                                                             \
    _Pragma("unroll") for (int t = 0; t < WALK_N; ++t) {                                      \
        const int idx = (half) * WALK_N + t;                                                  \
        const int cell = __builtin_amdgcn_readlane((rec).cell, idx);                          \
        if (cell != cur_cell) {                                                               \
            flush();                                                                          \
            cur_cell = cell;                                                                  \
            cur_xy = (unsigned)__builtin_amdgcn_readlane((int)(rec).xy, idx);                 \
            acc0 = 0.f;                                                                       \
            acc1 = 0.f;                                                                       \
        }                                                                                     \
        const float tx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, (rec).tx), idx)); \
        const float ty = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, (rec).ty), idx)); \
        const float wx = hx ? tx : 1.0f - tx;                                                 \
        const float g = (cell != DUMMY) ? buf[t] : 0.0f;                                      \
        acc0 += g * (wx * (1.0f - ty));                                                       \
        acc1 += g * (wx * ty);                                                                \
    }

    const int nblk = BUNDLE_MAX / 4 / WAVE;
    const int ngrp = WAVE / WALK_N;                 // groups per 64-entry record block (even)
    float ga[WALK_N], gb[WALK_N];
    Rec rec = fetch(0);
    LOAD_HALF(ga, rec, 0)
#pragma unroll 1
    for (int blk = 0; blk < nblk; ++blk) {
        if (__builtin_amdgcn_readfirstlane(rec.cell) == DUMMY) break;          // sorted: everything from here is padding
        Rec nxt = rec;
        if (blk + 1 < nblk) nxt = fetch(blk + 1);
#pragma unroll
        for (int g2 = 0; g2 < ngrp; g2 += 2) {
            LOAD_HALF(gb, rec, g2 + 1)
            WALK_HALF(ga, rec, g2)
            if (g2 + 2 < ngrp) {
                LOAD_HALF(ga, rec, g2 + 2)
            } else if (blk + 1 < nblk) {
                LOAD_HALF(ga, nxt, 0)
            }
            WALK_HALF(gb, rec, g2 + 1)
        }
        rec = nxt;
    }
#undef LOAD_HALF
#undef WALK_HALF
    flush();
}

// ---------------------------------------------------------------------------------------------------------
// host side (called from eslam_render_bwd.hip)
// ---------------------------------------------------------------------------------------------------------
static int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}

int eslam_scatter_v2_init();

// perm [R] <- rays ordered by direction; chunks of SORT_MAX rays are ordered independently
extern "C" int eslam_ray_order(const float* rays_o, const float* rays_d, int R, int32_t* perm, eslam_stream_t stream) {
    if (R <= 0) return 0;
    if (!rays_o || !rays_d || !perm) {
        eslam_set_error("eslam_ray_order: null argument");
        return 1;
    }
    hipStream_t st = (hipStream_t)stream;
    if (int rc = eslam_scatter_v2_init()) return rc;
    const int chunks = (R + SORT_MAX - 1) / SORT_MAX;
    hipLaunchKernelGGL(ray_order_kernel, dim3(chunks), dim3(1024), ORD_CELLS * sizeof(unsigned), st, rays_o, rays_d, R,
                       perm);
    return eslam_check_launch("ray_order_kernel");
}

// perm: ray order to bundle by (render mode), or NULL for the given order
int eslam_scatter_v2(const eslam_plane_t* planes, const Bound& bnd, const float* rays_o, const float* rays_d,
                     const float* z_or_pts, int64_t R, int S, bool render, const float* g_feat, const int* perm,
                     hipStream_t st) {
    PlaneSet ps;
    for (int i = 0; i < NPL; ++i) {
        ps.p[i] = planes[i];
        if (planes[i].w > 4096 || planes[i].h > 4096 || (int64_t)planes[i].w * planes[i].h >= (1 << 22)) {
            eslam_set_error("scatter: plane %d is %d x %d, the cell key supports up to 4096 per side and 2^22 cells", i,
                            planes[i].h, planes[i].w);
            return 1;
        }
    }
    const int64_t N = render ? R * S : R;
    const int nunits = render ? (int)R : (int)((N + 63) / 64);
    static const int nosort = env_int("ESLAM_SC_NOSORT", 0), dbg_mode = env_int("ESLAM_SC_MODE", 0);       // A/B switch for profiling only
    if (nosort) perm = nullptr;
    const int per = render ? S : 64;
    const int bundle = BUNDLE_MAX / per;          // S <= ESLAM_MAX_SAMPLES = 256 -> at least 4 rays
    dim3 grid((nunits + bundle - 1) / bundle, NPL), block(256);
    if (render)
        hipLaunchKernelGGL((scatter_sort_kernel<true>), grid, block, 0, st, ps, bnd, rays_o, rays_d, z_or_pts,
                           perm, (int)R, S, g_feat, bundle, dbg_mode);
    else
        hipLaunchKernelGGL((scatter_sort_kernel<false>), grid, block, 0, st, ps, bnd, rays_o, rays_d, z_or_pts,
                           (const int*)nullptr, (int)R, 64, g_feat, bundle, dbg_mode);
    return eslam_check_launch("scatter_sort_kernel");
}

int eslam_scatter_v2_init() {
    static bool done = false;
    if (done) return 0;
    if (hipFuncSetAttribute((const void*)ray_order_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            ORD_CELLS * (int)sizeof(unsigned)) != hipSuccess) {
        eslam_set_error("scatter: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
        return 2;
    }
    done = true;
    return 0;
}
