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

#define SORT_MAX 16384

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

// One workgroup sorts one chunk of up to SORT_MAX rays (bitonic network in LDS).  perm[chunk*SORT_MAX + i] = ray id.
__global__ __launch_bounds__(1024) void ray_order_kernel(const float* __restrict__ rays_o,
                                                         const float* __restrict__ rays_d, int R, const Bound bnd,
                                                         int* __restrict__ perm) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long skey[];      // (key << 32) | local index
    const int base = blockIdx.x * SORT_MAX;
    const int n = min(SORT_MAX, R - base);
    int npow = 1;
    while (npow < n) npow <<= 1;
    for (int i = threadIdx.x; i < npow; i += blockDim.x) {
        unsigned long long kv = ~0ull;                                                // padding sorts to the end
        if (i < n) {
            const int ray = base + i;
            const float dx = rays_d[3 * ray], dy = rays_d[3 * ray + 1], dz = rays_d[3 * ray + 2];
            const float inv = rsqrtf(fmaxf(dx * dx + dy * dy + dz * dz, 1e-20f));
            const unsigned qx = (unsigned)fminf(fmaxf((dx * inv + 1.0f) * 32.0f, 0.0f), 63.0f);
            const unsigned qy = (unsigned)fminf(fmaxf((dy * inv + 1.0f) * 32.0f, 0.0f), 63.0f);
            const unsigned qz = (unsigned)fminf(fmaxf((dz * inv + 1.0f) * 32.0f, 0.0f), 63.0f);
            const unsigned dkey = spread3(qx) | (spread3(qy) << 1) | (spread3(qz) << 2);       // 18 bits
            // origin cell (16 per axis over the scene bound): rays of different cameras never share a bundle prefix
            const unsigned ox = (unsigned)fminf(fmaxf((rays_o[3 * ray] - bnd.lo[0]) / (bnd.hi[0] - bnd.lo[0]) * 16.0f, 0.0f), 15.0f);
            const unsigned oy = (unsigned)fminf(fmaxf((rays_o[3 * ray + 1] - bnd.lo[1]) / (bnd.hi[1] - bnd.lo[1]) * 16.0f, 0.0f), 15.0f);
            const unsigned oz = (unsigned)fminf(fmaxf((rays_o[3 * ray + 2] - bnd.lo[2]) / (bnd.hi[2] - bnd.lo[2]) * 16.0f, 0.0f), 15.0f);
            const unsigned okey = spread3(ox) | (spread3(oy) << 1) | (spread3(oz) << 2);       // 12 bits
            kv = ((unsigned long long)((okey << 18) | dkey) << 32) | (unsigned)i;
        }
        skey[i] = kv;
    }
    __syncthreads();
    for (int k = 2; k <= npow; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < npow; i += blockDim.x) {
                const int l = i ^ j;
                if (l > i) {
                    const unsigned long long a = skey[i], b = skey[l];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { skey[i] = b; skey[l] = a; }
                }
            }
            __syncthreads();
        }
    }
    for (int i = threadIdx.x; i < n; i += blockDim.x) perm[base + i] = base + (int)(skey[i] & 0xFFFFFFFFu);
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
                                                           const float* __restrict__ g_feat, int bundle) {
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
    // (2) bitonic sort of the keys (invalid slots carry the maximum key and end up last)
    for (int k = 2; k <= BUNDLE_MAX; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
#pragma unroll
            for (int t = 0; t < BUNDLE_MAX / 512; ++t) {
                const int p = threadIdx.x + t * 256;                      // pair index
                const int i = ((p & ~(j - 1)) << 1) | (p & (j - 1));      // lower element of the pair
                const int l = i | j;
                const unsigned a = skey[i], b = skey[l];
                const bool up = (i & k) == 0;
                if ((a > b) == up) { skey[i] = b; skey[l] = a; }
            }
            __syncthreads();
        }
    }
    // (3) walk: wave w owns sorted entries [w*256, w*256+256)
    const float* __restrict__ gcol = g_feat + d * 64 + lvl * 32 + c;
    int cur_cell = -1;
    unsigned cur_xy = 0;
    float acc0 = 0.f, acc1 = 0.f;
    const int e0 = wave * (BUNDLE_MAX / 4);
#pragma unroll 1
    for (int e = e0; e < e0 + BUNDLE_MAX / 4; e += 4) {
        unsigned key[4];
        float g[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            key[t] = skey[e + t];                                         // LDS broadcast reads (wave-uniform)
            const int slot = key[t] & 1023u;
            g[t] = (key[t] != 0xFFFFFFFFu) ? gcol[(int64_t)sgrow[slot] * 128] : 0.0f;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (key[t] == 0xFFFFFFFFu) break;                             // wave-uniform: padding from here on
            const int slot = key[t] & 1023u;
            const int cell = (int)(key[t] >> 10);
            if (cell != cur_cell) {                                        // wave-uniform
                if (cur_cell >= 0) {
                    const int x0 = cur_xy & 0xFFF, y0 = (cur_xy >> 12) & 0xFFF;
                    const int dxs = ((cur_xy >> 24) & 1) * psx, dys = ((cur_xy >> 25) & 1) * psy;
                    float* gp = grad + y0 * psy + x0 * psx + hx * dxs + c * psc;
                    atomicAdd(gp, acc0);
                    atomicAdd(gp + dys, acc1);
                }
                cur_cell = cell;
                cur_xy = sxy[slot];
                acc0 = 0.f;
                acc1 = 0.f;
            }
            const float tx = swx[slot], ty = swy[slot];
            const float wx = hx ? tx : 1.0f - tx;
            acc0 += g[t] * (wx * (1.0f - ty));
            acc1 += g[t] * (wx * ty);
        }
    }
    if (cur_cell >= 0) {
        const int x0 = cur_xy & 0xFFF, y0 = (cur_xy >> 12) & 0xFFF;
        const int dxs = ((cur_xy >> 24) & 1) * psx, dys = ((cur_xy >> 25) & 1) * psy;
        float* gp = grad + y0 * psy + x0 * psx + hx * dxs + c * psc;
        atomicAdd(gp, acc0);
        atomicAdd(gp + dys, acc1);
    }
}

// ---------------------------------------------------------------------------------------------------------
// host side (called from eslam_render_bwd.hip)
// ---------------------------------------------------------------------------------------------------------
static int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}

int eslam_scatter_v2(const eslam_plane_t* planes, const Bound& bnd, const float* rays_o, const float* rays_d,
                     const float* z_or_pts, int64_t R, int S, bool render, const float* g_feat, int* perm,
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
    static const int nosort = env_int("ESLAM_SC_NOSORT", 0);       // A/B switch for profiling only
    if (nosort) perm = nullptr;
    if (render && perm) {
        const int chunks = (int)((R + SORT_MAX - 1) / SORT_MAX);
        int npow = 1;
        while (npow < (R < SORT_MAX ? (int)R : SORT_MAX)) npow <<= 1;
        hipLaunchKernelGGL(ray_order_kernel, dim3(chunks), dim3(1024), npow * sizeof(unsigned long long), st, rays_o,
                           rays_d, (int)R, bnd, perm);
        if (int rc = eslam_check_launch("ray_order_kernel")) return rc;
    }
    const int per = render ? S : 64;
    const int bundle = BUNDLE_MAX / per;          // S <= ESLAM_MAX_SAMPLES = 256 -> at least 4 rays
    dim3 grid((nunits + bundle - 1) / bundle, NPL), block(256);
    if (render)
        hipLaunchKernelGGL((scatter_sort_kernel<true>), grid, block, 0, st, ps, bnd, rays_o, rays_d, z_or_pts,
                           (const int*)perm, (int)R, S, g_feat, bundle);
    else
        hipLaunchKernelGGL((scatter_sort_kernel<false>), grid, block, 0, st, ps, bnd, rays_o, rays_d, z_or_pts,
                           (const int*)nullptr, (int)R, 64, g_feat, bundle);
    return eslam_check_launch("scatter_sort_kernel");
}

int eslam_scatter_v2_init() {
    static bool done = false;
    if (done) return 0;
    if (hipFuncSetAttribute((const void*)ray_order_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            SORT_MAX * (int)sizeof(unsigned long long)) != hipSuccess) {
        eslam_set_error("scatter: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
        return 2;
    }
    done = true;
    return 0;
}
