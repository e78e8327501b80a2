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
#include <mutex>
#include <type_traits>
#include "eslam_decode_tile.h"
#include "eslam_dec_reduce.h"

#define SORT_MAX 8192
typedef float float2_t __attribute__((ext_vector_type(2)));

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
// When all rays of the chunk leave from ONE point (a tracking batch; a mapping batch of a single frame) their
// directions form a 2-D patch, and a 3-D Morton code wastes a third of its bits on a coordinate the other two
// determine: the key is then the 16-bit Hilbert index of the direction's gnomonic projection about the mean
// direction - neighbours along the curve are always neighbours in the image.  Measured on the bench workload:
// 14 % fewer cell flushes, scatter 124 -> 116 us (tools/sim_order.py, tools/exp_order.py).
#ifndef RAY_ORDER_AZIMUTH
#define RAY_ORDER_AZIMUTH 1       // A/B switch: 0 = the three orders are the same 2-D Hilbert order (round 2)
#endif
#define ORD_BITS 5
#define ORD_CELLS (1 << (3 * ORD_BITS))      // 32768 words = 65536 packed 16-bit counters = 128 KB of LDS
#define ORD_PER_THREAD (SORT_MAX / 1024)
// key_bits (12, 14 or 16): keys of the counting sort.  The histogram's zero fill and scan are what this single-workgroup
// kernel spends its time on, so small batches get short keys (12 bits for <= 1024 rays: 2048 words instead of 32768).
__global__ __launch_bounds__(1024) void ray_order_kernel(const float* __restrict__ rays_o,
                                                         const float* __restrict__ rays_d, int R,
                                                         int* __restrict__ perm, int key_bits, int max_cams) {
    extern __shared__ __attribute__((aligned(16))) unsigned hist[];       // [(1 << key_bits) / 2] packed 16-bit counters
    const int nwords = (1 << key_bits) >> 1;
    const int hbits = key_bits >> 1;                                      // Hilbert grid: 2^hbits x 2^hbits
    const int mbits = key_bits / 3;                                       // Morton grid: 2^mbits per axis
    __shared__ float red[16][6];
    __shared__ float red2[16][13];
    __shared__ unsigned wsum[16];
    // several origins (a keyframe window, src/Mapper.py:308-319: the batch is camera-major): which rays start a new camera
    // (bit i of the map: ray i leaves from another point than ray i - 1), the running count per 64 rays, and per camera the sum
    // of its directions and the extent of its fan in this plane
    constexpr int NCAM_MAX = 32;
    __shared__ unsigned long long cam_bits[SORT_MAX / 64];
    __shared__ unsigned cam_before[SORT_MAX / 64 + 1];
    __shared__ float cam_dir[NCAM_MAX][3];
    __shared__ unsigned cam_ext[NCAM_MAX][2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < SORT_MAX / 64) cam_bits[tid] = 0ull;
    if (tid < NCAM_MAX) { cam_dir[tid][0] = cam_dir[tid][1] = cam_dir[tid][2] = 0.f; cam_ext[tid][0] = 0xFFFFFFFFu; cam_ext[tid][1] = 0u; }
    __syncthreads();
    const int base = blockIdx.x * SORT_MAX;
    const int n = min(SORT_MAX, R - base);
    // blockIdx.y = plane orientation (0: xy, 1: xz, 2: yz): order number o is written to perm + o * R.  When the rays share one
    // origin, order o sorts them by the AZIMUTH of their direction projected into plane o: the rays of a bundle then lie on top
    // of each other in that plane's projection whatever their angle out of it, and that is what shares cells - a plane collapses
    // one axis, so a square patch of the image (the Hilbert order: one order for all planes) spreads over many more cells of
    // each plane than a thin wedge does.  tools/sim_order.py, bench rays: 58.8 k cell flushes against 135 k.
    const int orient = blockIdx.y;
    float* const fan = (float*)(perm + (size_t)ESLAM_RAY_ORDERS * R);      // [ESLAM_RAY_ORDERS] angular extent of the fan in each plane (0: unknown)
    perm += (size_t)orient * R;
    const int pa = orient == 2 ? 1 : 0, pb = orient == 0 ? 1 : 2;       // the plane's two axes

    // Rays are re-read from memory (98 KB, cache resident) in every pass instead of being held in 24 registers per
    // thread: at 1024 threads per workgroup the budget is 128 VGPRs, and the Hilbert path spilled.
    auto unit_dir = [&](int ray, float o[3], float d[3]) {
        const float dx = rays_d[3 * ray], dy = rays_d[3 * ray + 1], dz = rays_d[3 * ray + 2];
        const float inv = rsqrtf(fmaxf(dx * dx + dy * dy + dz * dz, 1e-20f));
        o[0] = rays_o[3 * ray]; o[1] = rays_o[3 * ray + 1]; o[2] = rays_o[3 * ray + 2];
        d[0] = dx * inv; d[1] = dy * inv; d[2] = dz * inv;
    };
    float lo[3] = {3.4e38f, 3.4e38f, 3.4e38f}, hi[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
    float olo[3] = {3.4e38f, 3.4e38f, 3.4e38f}, ohi[3] = {-3.4e38f, -3.4e38f, -3.4e38f}, dsum[3] = {0.f, 0.f, 0.f};
    for (int i = tid; i < n; i += 1024) {
        float o[3], d[3];
        unit_dir(base + i, o, d);
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            olo[a] = fminf(olo[a], o[a]); ohi[a] = fmaxf(ohi[a], o[a]);
            dsum[a] += d[a];
            const float p = o[a] + d[a];
            lo[a] = fminf(lo[a], p); hi[a] = fmaxf(hi[a], p);
        }
        bool first = i == 0;
        if (!first) {
            const float* po = rays_o + 3 * (size_t)(base + i - 1);
            first = po[0] != o[0] || po[1] != o[1] || po[2] != o[2];
        }
        const unsigned long long fb = __ballot(first);          // (a wave's rays of one trip are 64 consecutive ones)
        if (lane == 0) cam_bits[i >> 6] = fb;
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], m, WAVE));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], m, WAVE));
            olo[a] = fminf(olo[a], __shfl_xor(olo[a], m, WAVE));
            ohi[a] = fmaxf(ohi[a], __shfl_xor(ohi[a], m, WAVE));
            dsum[a] += __shfl_xor(dsum[a], m, WAVE);
        }
        if (lane == 0) {
            red[wave][a] = lo[a]; red[wave][3 + a] = hi[a];
            red2[wave][a] = olo[a]; red2[wave][3 + a] = ohi[a]; red2[wave][6 + a] = dsum[a];
        }
    }
    for (int i = tid; i < nwords; i += 1024) hist[i] = 0u;
    __syncthreads();
    // cameras of the chunk: running count of "first ray of a camera" bits in front of every 64-ray word
    if (wave == 0) {
        constexpr int NW = SORT_MAX / 64;                        // 128 words: two per lane
        const unsigned c0 = __popcll(cam_bits[2 * lane]), c1 = __popcll(cam_bits[2 * lane + 1]);
        const unsigned incl = wave_incl_sum_u(c0 + c1);
        cam_before[2 * lane] = incl - c0 - c1;
        cam_before[2 * lane + 1] = incl - c1;
        if (lane == 63) cam_before[NW] = incl;
    }
    __syncthreads();
    const int ncam = (int)cam_before[SORT_MAX / 64];
    auto camera_of = [&](int i) {                                // 0-based camera of ray i of the chunk
        const unsigned long long below = cam_bits[i >> 6] & (~0ull >> (63 - (i & 63)));
        return (int)(cam_before[i >> 6] + (unsigned)__popcll(below)) - 1;
    };
    auto sortable = [](float x) { const unsigned u = __float_as_uint(x); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); };
    auto unsortable = [](unsigned u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u); };
    const bool per_camera = RAY_ORDER_AZIMUTH && ncam >= 2 && ncam <= min(NCAM_MAX, max_cams);
    // one origin?  then order by the Hilbert index of the direction in a 2-D chart about the mean direction
    float mdir[3];
    bool single = true;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float l = red2[0][a], h = red2[0][3 + a], sm = red2[0][6 + a];
        for (int w = 1; w < 16; ++w) { l = fminf(l, red2[w][a]); h = fmaxf(h, red2[w][3 + a]); sm += red2[w][6 + a]; }
        mdir[a] = sm;
        single = single && (h - l) <= 1e-6f * fmaxf(1.0f, fabsf(h));
    }
    const float mlen = sqrtf(mdir[0] * mdir[0] + mdir[1] * mdir[1] + mdir[2] * mdir[2]);
    single = single && mlen > 0.5f * (float)n;          // a usable mean direction (field of view well below 180 degrees)
    unsigned key[ORD_PER_THREAD], ticket[ORD_PER_THREAD];
#pragma unroll
    for (int k = 0; k < ORD_PER_THREAD; ++k) { key[k] = 0; ticket[k] = 0; }
    if (per_camera) {                                   // uniform over the workgroup
        // A keyframe window: key = (camera, azimuth of the direction projected into this plane, measured against the camera's own
        // projected mean direction) - each camera's rays form thin wedges of the plane, as the single camera's do below.
        // tools/sim_order_window.py (10 cameras x 400 rays x 40): 211 k cell flushes against 338 k for the Morton order.
        for (int i = tid; i < n; i += 1024) {
            float o[3], d[3];
            unit_dir(base + i, o, d);
            const int c = camera_of(i);
            atomicAdd(&cam_dir[c][0], d[0]); atomicAdd(&cam_dir[c][1], d[1]); atomicAdd(&cam_dir[c][2], d[2]);
        }
        __syncthreads();
        auto angle = [&](int i, int& c) {
            float o[3], d[3];
            unit_dir(base + i, o, d);
            c = camera_of(i);
            const float ma = cam_dir[c][pa], mb = cam_dir[c][pb];
            return atan2f(ma * d[pb] - mb * d[pa], ma * d[pa] + mb * d[pb]);
        };
        for (int i = tid; i < n; i += 1024) {
            int c;
            const unsigned u = sortable(angle(i, c));
            atomicMin(&cam_ext[c][0], u);
            atomicMax(&cam_ext[c][1], u);
        }
        __syncthreads();
        const int cbits = 32 - __clz(ncam - 1);          // bits for the camera (ncam >= 2)
        const int abits = key_bits - cbits;              // >= 7: key_bits >= 12, cbits <= 5
        if (tid == 0 && blockIdx.x == 0) fan[orient] = 0.0f;      // several origins: no single fan
#pragma unroll
        for (int k = 0; k < ORD_PER_THREAD; ++k) {
            const int i = tid + k * 1024;
            if (i < n) {
                int c;
                const float ang = angle(i, c);
                const float l = unsortable(cam_ext[c][0]), h = unsortable(cam_ext[c][1]);
                const unsigned amax = (1u << abits) - 1u;
                const unsigned qa = min((unsigned)fmaxf((ang - l) * ((float)(1u << abits) / fmaxf(h - l, 1e-6f)), 0.f), amax);
                const unsigned kk = ((unsigned)c << abits) | qa;
                key[k] = kk;
                ticket[k] = (atomicAdd(&hist[kk >> 1], 1u << ((kk & 1u) * 16u)) >> ((kk & 1u) * 16u)) & 0xFFFFu;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    } else if (single) {                                // uniform over the workgroup
        const float m0 = mdir[0] / mlen, m1 = mdir[1] / mlen, m2 = mdir[2] / mlen;
        // e1 = normalize(m x axis least aligned with m), e2 = m x e1
        float ax0 = 1.f, ax1 = 0.f, ax2 = 0.f;
        if (fabsf(m1) <= fabsf(m0) && fabsf(m1) <= fabsf(m2)) { ax0 = 0.f; ax1 = 1.f; }
        else if (fabsf(m2) <= fabsf(m0) && fabsf(m2) <= fabsf(m1)) { ax0 = 0.f; ax2 = 1.f; }
        float e10 = m1 * ax2 - m2 * ax1, e11 = m2 * ax0 - m0 * ax2, e12 = m0 * ax1 - m1 * ax0;
        const float el = rsqrtf(e10 * e10 + e11 * e11 + e12 * e12);
        e10 *= el; e11 *= el; e12 *= el;
        const float e20 = m1 * e12 - m2 * e11, e21 = m2 * e10 - m0 * e12, e22 = m0 * e11 - m1 * e10;
        const float mm[3] = {m0, m1, m2};
        const float mpa = mm[pa], mpb = mm[pb];           // the mean direction projected into the plane
        auto chart = [&](int ray, float& u, float& v) {
            float o[3], d[3];
            unit_dir(ray, o, d);
            if (RAY_ORDER_AZIMUTH) {                     // u = signed angle between the projected direction and the projected mean
                u = atan2f(mpa * d[pb] - mpb * d[pa], mpa * d[pa] + mpb * d[pb]);
                v = 0.0f;
                return;
            }
            // gnomonic chart about the mean direction, clamped at ~87 degrees
            const float t = fmaxf(d[0] * m0 + d[1] * m1 + d[2] * m2, 0.05f);
            u = (d[0] * e10 + d[1] * e11 + d[2] * e12) / t;
            v = (d[0] * e20 + d[1] * e21 + d[2] * e22) / t;
        };
        float blo[2] = {3.4e38f, 3.4e38f}, bhi[2] = {-3.4e38f, -3.4e38f};
        for (int i = tid; i < n; i += 1024) {
            float u, v;
            chart(base + i, u, v);
            blo[0] = fminf(blo[0], u); bhi[0] = fmaxf(bhi[0], u);
            blo[1] = fminf(blo[1], v); bhi[1] = fmaxf(bhi[1], v);
        }
#pragma unroll
        for (int a = 0; a < 2; ++a) {
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) {
                blo[a] = fminf(blo[a], __shfl_xor(blo[a], m, WAVE));
                bhi[a] = fmaxf(bhi[a], __shfl_xor(bhi[a], m, WAVE));
            }
            if (lane == 0) { red2[wave][9 + a] = blo[a]; red2[wave][11 + a] = bhi[a]; }
        }
        __syncthreads();
        float sc2[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            float l = red2[0][9 + a], h = red2[0][11 + a];
            for (int w = 1; w < 16; ++w) { l = fminf(l, red2[w][9 + a]); h = fmaxf(h, red2[w][11 + a]); }
            blo[a] = l;
            sc2[a] = (float)(1 << (RAY_ORDER_AZIMUTH ? key_bits : hbits)) / fmaxf(h - l, 1e-6f);
            // the fan's angular extent in this plane (radians), for the scatter's choice of grid order; the first chunk speaks for the batch
            if (a == 0 && tid == 0 && blockIdx.x == 0) fan[orient] = RAY_ORDER_AZIMUTH ? h - l : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < ORD_PER_THREAD; ++k) {
            const int i = tid + k * 1024;
            if (i < n) {
                float u, v;
                chart(base + i, u, v);
                const unsigned hmax = (1u << (RAY_ORDER_AZIMUTH ? key_bits : hbits)) - 1u;
                unsigned x = min((unsigned)fmaxf((u - blo[0]) * sc2[0], 0.f), hmax);
                unsigned y = min((unsigned)fmaxf((v - blo[1]) * sc2[1], 0.f), hmax);
                unsigned d = RAY_ORDER_AZIMUTH ? x : 0u; // azimuth: the quantised angle is the key; else the Hilbert index of (x, y)
                for (unsigned sft = RAY_ORDER_AZIMUTH ? 0u : 1u << (hbits - 1); sft > 0; sft >>= 1) {
                    const unsigned rx = (x & sft) ? 1u : 0u, ry = (y & sft) ? 1u : 0u;
                    d += sft * sft * ((3u * rx) ^ ry);
                    if (ry == 0) {
                        if (rx == 1) { x = hmax - x; y = hmax - y; }
                        const unsigned tswap = x; x = y; y = tswap;
                    }
                }
                key[k] = d;
                ticket[k] = (atomicAdd(&hist[d >> 1], 1u << ((d & 1u) * 16u)) >> ((d & 1u) * 16u)) & 0xFFFFu;
            }
            __builtin_amdgcn_sched_barrier(0);           // one ray at a time keeps the register pressure down
        }
    } else {
        if (tid == 0 && blockIdx.x == 0) fan[orient] = 0.0f;      // several origins: no fan
        float scale[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            float l = red[0][a], h = red[0][3 + a];
            for (int w = 1; w < 16; ++w) { l = fminf(l, red[w][a]); h = fmaxf(h, red[w][3 + a]); }
            lo[a] = l;
            scale[a] = (float)(1 << mbits) / fmaxf(h - l, 1e-6f);
        }
#pragma unroll
        for (int k = 0; k < ORD_PER_THREAD; ++k) {
            const int i = tid + k * 1024;
            if (i < n) {
                float o[3], d[3];
                unit_dir(base + i, o, d);
                const unsigned qmax = (1u << mbits) - 1;
                const unsigned qx = min((unsigned)fmaxf((o[0] + d[0] - lo[0]) * scale[0], 0.f), qmax);
                const unsigned qy = min((unsigned)fmaxf((o[1] + d[1] - lo[1]) * scale[1], 0.f), qmax);
                const unsigned qz = min((unsigned)fmaxf((o[2] + d[2] - lo[2]) * scale[2], 0.f), qmax);
                const unsigned kk = spread3(qx) | (spread3(qy) << 1) | (spread3(qz) << 2);
                key[k] = kk;
                ticket[k] = (atomicAdd(&hist[kk >> 1], 1u << ((kk & 1u) * 16u)) >> ((kk & 1u) * 16u)) & 0xFFFFu;
            }
        }
    }
    __syncthreads();
    // exclusive scan of the counters (two 16-bit counters per word, at most SORT_MAX = 8192 rays: no overflow):
    // thread t owns words [32t, 32t+32)
    const int per = nwords / 1024;                     // >= 2 (key_bits >= 12)
    unsigned local = 0;
    for (int j = 0; j < per; ++j) {
        const unsigned w = hist[tid * per + j];
        local += (w & 0xFFFFu) + (w >> 16);
    }
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
        const unsigned w = hist[tid * per + j];
        const unsigned c0 = w & 0xFFFFu, c1 = w >> 16;
        hist[tid * per + j] = run | ((run + c0) << 16);
        run += c0 + c1;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < ORD_PER_THREAD; ++k) {
        const int i = tid + k * 1024;
        if (i < n) {
            const unsigned start = (hist[key[k] >> 1] >> ((key[k] & 1u) * 16u)) & 0xFFFFu;
            perm[base + start + ticket[k]] = base + i;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// bundle scatter
// ---------------------------------------------------------------------------------------------------------
// NT threads per workgroup, BM = 4*NT samples per workgroup (power of two).
//   DBG: 0 production; 1 walk without atomics; 2 stop after the sort; 3 = 1 without the g_feat row loads; 4 = 3 without the
//   LDS weight reads; 5 = 3 WITH the atomics (atomics, but no loads queued behind them); 6 = 0 with plain stores in place of
//   the atomics (profiling only, tools/scatter_anatomy.sh; 5 and 6 write garbage into the gradients)
//   (Round 2 had the kernel's two halves - cells + sort / walk - as separately launchable phases, the first one beside the
//   forward kernel on a side stream: 0.369 vs 0.323 ms per step, DESIGN.md section 10.  Removed in round 3.)
#ifndef SC_STAMPS
#define SC_STAMPS 0                                   // profiling only: per-phase cycles of one workgroup's first thread
#endif
//   DET (ESLAM_DETERMINISTIC=1): the sums of a cell are formed in 64-bit fixed point (2^-44 units: integer adds commute, so
//   neither the arbitrary order of the counting sort's tickets inside a cell, nor the order in which workgroups' atomics
//   reach a texel, nor the ray order itself can change a bit of the result) and added to an int64 shadow of the gradient
//   planes; scatter_fixed_to_float_kernel then adds the shadow to the float gradients and clears it.
#define FIX_SCALE 17592186044416.0f                  // 2^44: resolution 5.7e-14
#define FIX_LIMIT 262144.0f                          // |one contribution| < 2^18: 2^62 in fixed point; a texel's SUM wraps beyond
                                                     // 2^63 / 2^44 = 5.2e5 - contributions past the limit poison the gradient (NaN)
struct ShadowOff { int64_t o[NPL]; };
template <bool RENDER, int DBG, int NT, bool DET, int SPT = 4>
__global__ __launch_bounds__(NT) void scatter_sort_kernel(const PlaneSet planes, const Bound bnd,
                                                          const float* __restrict__ rays_o,
                                                          const float* __restrict__ rays_d,
                                                          const float* __restrict__ z_vals,     // RENDER ? [R,S] : pts [N,3]
                                                          const int* __restrict__ perm, int R, int S,
                                                          const float* __restrict__ g_feat, int bundle,
                                                          int allow_counting, int nbundles, int xcd_map,
                                                          long long* __restrict__ shadow,
                                                          const ShadowOff shoff, const DecReduceArgs red, const int red_blocks) {
    constexpr int dbg_mode = DBG;
    constexpr int BM = SPT * NT;                       // SPT samples per thread in the cell / sort phases
    constexpr int CH = WAVE * SPT;                     // sorted entries a wave walks
    static_assert(SPT == 2 || SPT == 4, "samples per thread");
    constexpr int SLOT_BITS = (BM == 1024) ? 10 : 11;
    constexpr unsigned SLOT_MASK = BM - 1;
    static_assert(BM == 1024 || BM == 2048, "bundle size");
    // 6*BM words (48 KB at BM = 2048; 3 workgroups per CU).  Layout the walk reads, records in SORTED order:
    //   sw    [BM] float4  bilinear weights of the entry for lane halves hx = 0 / 1 and rows 0 / 1:
    //                      ((1-tm)(1-tM), (1-tm)tM, tm(1-tM), tm tM) - a lane reads its (row 0, row 1) pair with one
    //                      ds_read_b64, so the per-entry VALU work of the walk is ONE packed FMA
    //   sxy   [BM] byte offset of texel (x0,y0) << 2 | minor-axis step flag | major flag << 1; 0xFFFFFFFF = padding
    //   sgrow [BM] row of g_feat (global point index)
    // The sort phases use the same memory differently (see below).
    __shared__ __attribute__((aligned(16))) unsigned lds_raw[6 * BM];
    float4_t* const sw = (float4_t*)lds_raw;
    unsigned* const sxy = lds_raw + 4 * BM;
    int* const sgrow = (int*)(lds_raw + 5 * BM);
    unsigned* const cnt = lds_raw;                     // counting sort: 4*BM packed 16-bit counters (dead before sw is written)
    unsigned* const swsum = lds_raw + 2 * BM;          // counting sort: per-wave totals of the scan
    int* const sbox = (int*)(lds_raw + 2 * BM + 64);   // bounding box of the bundle's cells (registers before sw is written)
    // bitonic fallback (boxes too large for the counters): keys + slot-indexed temporaries, permuted into the layout above
    unsigned* const skey = lds_raw;                    // (cell << SLOT_BITS) | local sample slot
    unsigned* const txy = lds_raw + BM;
    float* const ttm = (float*)(lds_raw + 2 * BM + 128);   // after sbox / swsum
    float* const ttM = (float*)(lds_raw + 3 * BM + 128);
    int* const trow = (int*)(lds_raw + 4 * BM + 128);
    constexpr unsigned PAD_XY = 0xFFFFFFFFu;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hx = lane >> 5, c = lane & 31;   // hx: corner along the MINOR axis

    // The first red_blocks workgroups (a multiple of 8) are not scatter workgroups:
    // they sum the decoder-gradient slabs of the decoder backward that ran just before (eslam_dec_reduce.h), beside the
    // scatter's first workgroups instead of as a launch of their own in front of them.
    if (red_blocks > 0 && (int)blockIdx.x < red_blocks) {
        if ((int)blockIdx.x < 2 * DEC_RED_COLBLOCKS)
            dec_grad_reduce_block<NT>(red, blockIdx.x % DEC_RED_COLBLOCKS, blockIdx.x / DEC_RED_COLBLOCKS, (float*)lds_raw);
        return;
    }
    const int bid = (int)blockIdx.x - red_blocks;
    int bidx, pi;
    // Workgroup -> (bundle, plane), 1-D grid (ESLAM_SC_XCDMAP; measured in profiles/r03/l_*, one ray order per orientation):
    //   4 (default) bundle-major: the 12 planes of bundle 0, then of bundle 1, ...  The four planes of an orientation bundle the
    //     same rays and read the four 128-byte segments of the same feature-gradient rows close together in time.
    //   1 round 2's dealing: the three orientations of one (bundle, decoder, level) as neighbours on ONE XCD - they read the SAME
    //     segment of the same rows while all planes shared one ray order (81 % of the row reads missed L2 without it); with an
    //     order per orientation they no longer share rays, and the dealing is 0-10 us slower than 4.
    //   3 plane-major: every bundle of plane 0, then of plane 1, ...  12-14 us FASTER than 4 on room0's batches (4096 x 64: 86 us)
    //     and 10-35 us slower on freiburg1_desk's and scene0000's, with identical FETCH / atomic counters: not understood, not the default.
    //   0 the 2-D (bundle, plane) grid of round 1 (= 3 without the slab reduction in the grid).
    if (xcd_map == 1) {
        const int xcd = bid & 7, j = bid >> 3;
        const int q = (j / 3) * 8 + xcd;                     // (bundle, segment) pair handled by this XCD slot
        if (q >= nbundles * 4) return;
        const int seg = q & 3;                               // decoder * 2 + level
        bidx = q >> 2;
        pi = (seg >> 1) * 6 + (j % 3) * 2 + (seg & 1);
    } else if (xcd_map == 3) {
        bidx = bid % nbundles;
        pi = bid / nbundles;
        if (pi >= NPL) return;
    } else if (xcd_map == 4) {
        // auto: plane-major for a WIDE fan of rays from one origin, bundle-major otherwise.  Measured (profiles/r03/n_*): with the
        // same pixels through lenses of different focal length, bundle-major costs 110 / 97 / 83 / 81 / 80 us at a horizontal field
        // of view of 118 / 90 / 67 / 53 / 37 degrees where plane-major stays at 99 / 87 / 87 / 88 / 90 - the orders cross near 80
        // degrees.  The fan's extent in each plane comes from the ray ordering kernel, through device memory (no host round trip);
        // the SECOND largest of the three is the criterion (the largest is ~360 degrees in the plane the camera looks down on).
        bool plane_major = false;
        if (RENDER && perm) {
            const float* fan = (const float*)(perm + (size_t)ESLAM_RAY_ORDERS * R);
            const float f0 = fan[0], f1 = fan[1], f2 = fan[2];
            const float second = fmaxf(fminf(f0, f1), fminf(fmaxf(f0, f1), f2));
            plane_major = second >= 1.4f;
        }
        if (plane_major) { bidx = bid % nbundles; pi = bid / nbundles; }
        else { pi = bid % NPL; bidx = bid / NPL; }
        if (bidx >= nbundles || pi >= NPL) return;
    } else {
        bidx = bid;
        pi = blockIdx.y;                                     // plane index in all_planes order
        if (bidx >= nbundles) return;
    }
    const int d = pi / 6, o = (pi % 6) >> 1, lvl = pi & 1;
    const eslam_plane_t& P = planes.p[pi];
    const int pw = P.w, ph = P.h;
    const int psy = (int)P.stride_y, psx = (int)P.stride_x, psc = (int)P.stride_c;
    float* __restrict__ grad = P.grad;
    const int64_t npts = RENDER ? (int64_t)R * S : (int64_t)R;       // decode mode: R = N points, unit = 64 points
    const int nunits = RENDER ? R : (int)((npts + 63) / 64);
    const int per = RENDER ? S : 64;                                  // samples per unit
    const int u0 = bidx * bundle;
    const int nu = min(bundle, nunits - u0);
    const int n = nu * per;                                           // <= BM by construction of `bundle`

#if SC_STAMPS
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last = clock64();
#define SSTAMP(i) { const unsigned long long now_ = clock64(); st_acc[i] += now_ - st_last; st_last = now_; }
#else
#define SSTAMP(i)
#endif
    bool swap = false;
    if (threadIdx.x == 0) { sbox[0] = 0x7FFFFFFF; sbox[1] = -1; sbox[2] = 0x7FFFFFFF; sbox[3] = -1; }
    __syncthreads();

    // (1) bilinear cell of every sample of the bundle (4 per thread), and the bundle's bounding box in the plane
    AxisCoord cax[SPT], cay[SPT];
    int cpt[SPT];
    int bx0 = 0x7FFFFFFF, bx1 = -1, by0 = 0x7FFFFFFF, by1 = -1;
#pragma unroll
    for (int k = 0; k < SPT; ++k) {
        const int slot = threadIdx.x + k * NT;
        cpt[k] = -1;
        if (slot < n) {
            const int ui = u0 + slot / per, s = slot % per;
            const int unit = (RENDER && perm) ? perm[(size_t)o * R + ui] : ui;      // the order made for this plane's orientation
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
                cax[k] = axis_coord((o == 2) ? y : x, pw);
                cay[k] = axis_coord((o == 0) ? y : z, ph);
                cpt[k] = (int)pt;
                bx0 = min(bx0, cax[k].i0); bx1 = max(bx1, cax[k].i0);
                by0 = min(by0, cay[k].i0); by1 = max(by1, cay[k].i0);
            }
        }
    }
    SSTAMP(0)
    bx0 = wave_min_i(bx0); bx1 = wave_max_i(bx1);
    by0 = wave_min_i(by0); by1 = wave_max_i(by1);
    if (lane == 0) {
        atomicMin(&sbox[0], bx0); atomicMax(&sbox[1], bx1);
        atomicMin(&sbox[2], by0); atomicMax(&sbox[3], by1);
    }
    __syncthreads();
    // The sort key runs along the axis the bundle travels along ("minor" = fastest-varying), so that consecutive cells
    // of the sorted list are neighbours along it and share a texel column that is carried instead of flushed twice.
    const int bxmin = sbox[0], bxmax = sbox[1], bymin = sbox[2], bymax = sbox[3];
    __syncthreads();                                                  // sbox's memory is reused from here on
    if (bxmax < 0) return;                                            // no valid sample in this bundle
    swap = (bymax - bymin) > (bxmax - bxmin);                         // travels along y: column-major keys
    // Bundles whose cells fit a small box (the normal case: 32 neighbouring rays) are ordered by a counting sort over
    // the box - one LDS integer atomic per sample, one scan, one placement pass - instead of the 66-stage bitonic
    // network (50 us of this kernel's 165).  Rows of the box get one padding column so that "next cell along the minor
    // axis" never wraps into the next row.  Order inside a cell is arbitrary; the cell's sum does not depend on it
    // beyond float rounding.
    const int mmin = swap ? bymin : bxmin, mext = (swap ? bymax : bxmax) - mmin + 2;
    const int Mmin = swap ? bxmin : bymin, Mext = (swap ? bxmax : bymax) - Mmin + 1;
    const bool counting = allow_counting && (int64_t)mext * Mext <= 4 * BM;
    SSTAMP(1)
    if (counting) {
        constexpr int WPT = 2 * BM / NT;                       // counter words per thread in the scan
        for (int i = threadIdx.x; i < 2 * BM; i += NT) cnt[i] = 0u;
        __syncthreads();
        unsigned loc[SPT], tick[SPT];
#pragma unroll
        for (int k = 0; k < SPT; ++k) {
            loc[k] = 0; tick[k] = 0;
            if (cpt[k] >= 0) {
                const int im = swap ? cay[k].i0 : cax[k].i0, iM = swap ? cax[k].i0 : cay[k].i0;
                loc[k] = (unsigned)((iM - Mmin) * mext + (im - mmin));
                const unsigned sh = (loc[k] & 1u) * 16u;
                tick[k] = (atomicAdd(&cnt[loc[k] >> 1], 1u << sh) >> sh) & 0xFFFFu;
            }
        }
        __syncthreads();
        SSTAMP(2)
        // exclusive scan of the counters: thread t owns words [WPT t, WPT t + WPT)
        unsigned w[WPT], local = 0;
#pragma unroll
        for (int j = 0; j < WPT; ++j) {
            w[j] = cnt[threadIdx.x * WPT + j];
            local += (w[j] & 0xFFFFu) + (w[j] >> 16);
        }
        const unsigned incl = wave_incl_sum_u(local);
        if (lane == WAVE - 1) swsum[wave] = incl;
        __syncthreads();
        unsigned run = incl - local, nvalid = 0;
        for (int wv = 0; wv < NT / WAVE; ++wv) {
            if (wv < wave) run += swsum[wv];
            nvalid += swsum[wv];
        }
#pragma unroll
        for (int j = 0; j < WPT; ++j) {
            const unsigned lo16 = w[j] & 0xFFFFu, hi16 = w[j] >> 16;
            cnt[threadIdx.x * WPT + j] = run | ((run + lo16) << 16);
            run += lo16 + hi16;
        }
        __syncthreads();
        SSTAMP(3)
        unsigned pos[SPT];
#pragma unroll
        for (int k = 0; k < SPT; ++k) pos[k] = ((cnt[loc[k] >> 1] >> ((loc[k] & 1u) * 16u)) & 0xFFFFu) + tick[k];
        __syncthreads();                                       // counters are dead: their memory becomes stm / stM
#pragma unroll
        for (int k = 0; k < SPT; ++k) {
            if (cpt[k] >= 0) {
                const AxisCoord& am = swap ? cay[k] : cax[k];
                const AxisCoord& aM = swap ? cax[k] : cay[k];
                const unsigned q = pos[k];                  // records are stored in sorted order
                sxy[q] = ((unsigned)(cay[k].i0 * psy + cax[k].i0 * psx) << 2) | (unsigned)(am.i1 > am.i0) |
                         ((unsigned)(aM.i1 > aM.i0) << 1);
                sw[q] = (float4_t){(1.0f - am.t) * (1.0f - aM.t), (1.0f - am.t) * aM.t, am.t * (1.0f - aM.t), am.t * aM.t};
                sgrow[q] = cpt[k];
            }
        }
        for (int i = (int)nvalid + threadIdx.x; i < BM; i += NT) {
            sxy[i] = PAD_XY;
            sw[i] = (float4_t){0.f, 0.f, 0.f, 0.f};
            sgrow[i] = 0;
        }
        __syncthreads();
    } else {
#pragma unroll
    for (int k = 0; k < SPT; ++k) {
        const int slot = threadIdx.x + k * NT;
        unsigned key = 0xFFFFFFFFu;
        if (cpt[k] >= 0) {
            const AxisCoord& am = swap ? cay[k] : cax[k];       // minor axis
            const AxisCoord& aM = swap ? cax[k] : cay[k];       // major axis
            const int cell = aM.i0 * (swap ? ph : pw) + am.i0;
            key = ((unsigned)cell << SLOT_BITS) | (unsigned)slot;
            txy[slot] = ((unsigned)(cay[k].i0 * psy + cax[k].i0 * psx) << 2) | (unsigned)(am.i1 > am.i0) |
                        ((unsigned)(aM.i1 > aM.i0) << 1);
            ttm[slot] = am.t;
            ttM[slot] = aM.t;
            trow[slot] = cpt[k];
        }
        skey[slot] = key;
    }
    __syncthreads();

    // (2) bitonic sort of the keys (invalid slots carry the maximum key and end up last).  Wave w owns elements
    // [256w, 256w+256): every compare-exchange distance j < 256 stays inside one wave's chunk and needs no workgroup
    // barrier (DS operations of a wave execute in order); only the stages with j >= 256 synchronise the workgroup.
    {
        const int wbase = wave * CH;
        for (int k = 2; k <= BM; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                if (j >= CH) {
                    __syncthreads();
#pragma unroll
                    for (int t = 0; t < SPT / 2; ++t) {
                        const int p = threadIdx.x + t * NT;                       // pair index
                        const int i = ((p & ~(j - 1)) << 1) | (p & (j - 1));      // lower element of the pair
                        const int l = i | j;
                        const unsigned a = skey[i], b2 = skey[l];
                        if ((a > b2) == ((i & k) == 0)) { skey[i] = b2; skey[l] = a; }
                    }
                    __syncthreads();
                } else {
#pragma unroll
                    for (int t = 0; t < SPT / 2; ++t) {
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
    // records into sorted order, through registers (the temporaries and the final layout overlap)
    __syncthreads();
    unsigned pxy[SPT];
    float ptm[SPT], ptM[SPT];
    int prow[SPT];
#pragma unroll
    for (int k = 0; k < SPT; ++k) {
        const unsigned key = skey[threadIdx.x + k * NT];
        const bool valid = key != 0xFFFFFFFFu;
        const int slot = key & SLOT_MASK;
        pxy[k] = valid ? txy[slot] : PAD_XY;
        ptm[k] = valid ? ttm[slot] : 0.f;
        ptM[k] = valid ? ttM[slot] : 0.f;
        prow[k] = valid ? trow[slot] : 0;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < SPT; ++k) {
        const int q = threadIdx.x + k * NT;
        const bool valid = pxy[k] != PAD_XY;
        sxy[q] = pxy[k];
        sw[q] = valid ? (float4_t){(1.0f - ptm[k]) * (1.0f - ptM[k]), (1.0f - ptm[k]) * ptM[k], ptm[k] * (1.0f - ptM[k]),
                                   ptm[k] * ptM[k]}
                      : (float4_t){0.f, 0.f, 0.f, 0.f};
        sgrow[q] = prow[k];
    }
    __syncthreads();
    }
    SSTAMP(4)
    if (dbg_mode == 2) return;

    // (3) walk: wave w owns sorted entries [256w, 256w+256), 64 at a time.  Per 64-entry block every lane fetches ONE
    // entry's (xy, g_feat row) and the block's cell boundaries become two 64-bit scalar masks (ballots): "entry starts a
    // new cell" and "... which is the next cell along the minor axis and shares a texel column".  The serial part per
    // entry is then a scalar bit test, one LDS read of the lane's two weights and ONE packed FMA - the version that
    // carried (cell, tm, tM) in registers and rebuilt the weights per entry spent ~11 VALU instructions there and was
    // issue-bound.  The g_feat values of WALK_N entries are loaded ahead of the walk of the
    // previous WALK_N: a wave's loads, stores and atomics retire in order on one vmcnt counter, so a load issued BEHIND an
    // atomic would wait for it (~3000 cycles under load).
    const char* __restrict__ gcol = (const char*)(g_feat + d * 64 + lvl * 32);     // + row * 512 + c * 4 bytes
    unsigned cur_xy = PAD_XY;                  // cell being accumulated (PAD_XY: none / padding, never flushed)
    unsigned last_xy = PAD_XY;                 // xy of the last entry of the previous block
    typedef typename std::conditional<DET, long long, float>::type acc_t;
    acc_t acc0 = 0, acc1 = 0;
    bool det_bad = false;                      // DET: a contribution of the current cell was not representable
    const int e0 = wave * CH;

    // Flush of a finished cell: lane (hx, c) adds its two sums (major-axis corners 0 and 1) for its minor-axis corner.
    // All addressing is 32-bit VALU arithmetic on a byte offset from the wave-uniform plane base (SGPR base + VGPR
    // offset form): a first version did 64-bit address arithmetic on the scalar unit, and with 32 waves per CU sharing
    // ONE scalar ALU the walk was SALU-bound (11 scalar instructions per entry).
    const unsigned lane_off = (unsigned)(c * psc) << 2;
    const unsigned dm_bytes = (unsigned)(swap ? psy : psx) << 2, dM_bytes = (unsigned)(swap ? psx : psy) << 2;
    char* __restrict__ gbytes = DET ? (char*)(shadow + shoff.o[pi]) : (char*)grad;
    auto flush = [&](bool lower_half_only) {
        if (cur_xy != PAD_XY) {
            const unsigned o0 = (cur_xy & ~3u) + lane_off + ((cur_xy & 1u) & (unsigned)hx) * dm_bytes;
            const unsigned o1 = o0 + ((cur_xy >> 1) & 1u) * dM_bytes;
            if (DET) {
                if (!lower_half_only || hx == 0) {           // the shadow mirrors the plane element for element: 8 bytes each
                    atomicAdd((unsigned long long*)(gbytes + 2 * (size_t)o0), (unsigned long long)acc0);
                    atomicAdd((unsigned long long*)(gbytes + 2 * (size_t)o1), (unsigned long long)acc1);
                    // a contribution that was NaN / Inf or beyond the fixed-point range (|g w| >= 2.6e5; __float2ll_rn would
                    // have saturated or wrapped it silently): poison the float gradient so that divergence stays visible
                    if (det_bad) {
                        atomicAdd((float*)((char*)grad + o0), __builtin_nanf(""));
                        atomicAdd((float*)((char*)grad + o1), __builtin_nanf(""));
                    }
                }
            } else if (dbg_mode == 0 || dbg_mode == 5) {
                if (!lower_half_only || hx == 0) {
                    atomicAdd((float*)(gbytes + o0), (float)acc0);
                    atomicAdd((float*)(gbytes + o1), (float)acc1);
                }
            } else if (dbg_mode == 6) {
                if (!lower_half_only || hx == 0) {
                    *(float*)(gbytes + o0) = (float)acc0;
                    *(float*)(gbytes + o1) = (float)acc1;
                }
            } else if ((float)acc0 == 1.2345e30f) *(float*)(gbytes + o0) = (float)acc1;      // profiling only: walk without atomics
        }
    };

    struct Rec { unsigned xy; int row; unsigned long long fresh, adjacent; };     // row: BYTE offset of the g_feat row (row * 512)
    auto fetch = [&](int blk, unsigned prev_last) {
        Rec r;
        const int e = e0 + blk * WAVE + lane;
        r.xy = sxy[e];
        r.row = sgrow[e] * 512;
        unsigned prev = __shfl_up(r.xy, 1, WAVE);
        if (lane == 0) prev = prev_last;
        const bool fresh = r.xy != prev;
        const bool adj = fresh && r.xy != PAD_XY && prev != PAD_XY && (prev & 1u) && (r.xy & ~3u) == (prev & ~3u) + dm_bytes;
        r.fresh = __ballot(fresh);
        r.adjacent = __ballot(adj);
        return r;
    };
#ifndef WALK_N
#define WALK_N 8                   // entries per load-ahead group (2 groups in flight: 2*WALK_N VGPRs)
#endif
    const float2_t* const wlane = (const float2_t*)sw + hx;                    // + 2 * entry
    // A row load is TWO instructions: v_readlane of the row's byte offset into an SGPR, and a buffer load that adds that SGPR
    // (soffset) and the lane's channel offset (voffset) to the descriptor's base - the global_load form needed a VALU add
    // per entry in between, and the walk is bound by instruction issue.
    const __amdgpu_buffer_rsrc_t grsrc = __builtin_amdgcn_make_buffer_rsrc((void*)gcol, 0, (int)((unsigned)npts * 512u - (unsigned)(d * 64 + lvl * 32) * 4u), 0x00020000);
    const int cvoff = c * 4;
#define LOAD_HALF(buf, rec, half)                                                             \
    _Pragma("unroll") for (int t = 0; t < WALK_N; ++t) {                                      \
        const int rowb = __builtin_amdgcn_readlane((rec).row, (half) * WALK_N + t);           \
        buf[t] = (dbg_mode >= 3 && dbg_mode <= 5) ? __int_as_float(rowb)                                       \
                                 : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(grsrc, cvoff, rowb, 0)); \
    }
#define WALK_HALF(buf, rec, half, ebase)                                                      \
    _Pragma("unroll") for (int t = 0; t < WALK_N; ++t) {                                      \
        const int idx = (half) * WALK_N + t;                                                  \
        const float2_t w2 = (dbg_mode == 4) ? (float2_t){1.f, 2.f} : wlane[2 * ((ebase) + idx)]; \
        if (((idx < 32 ? fresh_lo : fresh_hi) >> (idx & 31)) & 1u) {     /* one s_bitcmp on a 32-bit scalar */ \
            if (((idx < 32 ? adj_lo : adj_hi) >> (idx & 31)) & 1u) {                          \
                /* next cell along the minor axis: its first texel column is our second one - keep those sums */ \
                flush(true);          /* (det_bad stays: the carried column holds part of the cell's sums) */ \
                const acc_t s0 = __shfl_xor(acc0, 32, WAVE), s1 = __shfl_xor(acc1, 32, WAVE); \
                acc0 = hx ? (acc_t)0 : s0;                                                    \
                acc1 = hx ? (acc_t)0 : s1;                                                    \
            } else {                                                                          \
                flush(false);                                                                 \
                det_bad = false;                                                              \
                acc0 = 0;                                                                     \
                acc1 = 0;                                                                     \
            }                                                                                 \
            cur_xy = (unsigned)__builtin_amdgcn_readlane((int)(rec).xy, idx);                 \
        }                                                                                     \
        const float g = buf[t];        /* padding entries have zero weights and belong to no cell */ \
        if (DET) {                                                                            \
            const float t0_ = g * w2[0], t1_ = g * w2[1];                                     \
            det_bad = det_bad || !(fabsf(t0_) < FIX_LIMIT) || !(fabsf(t1_) < FIX_LIMIT);      \
            acc0 += (acc_t)__float2ll_rn(t0_ * FIX_SCALE);                                    \
            acc1 += (acc_t)__float2ll_rn(t1_ * FIX_SCALE);                                    \
        } else {                                                                              \
            acc0 += (acc_t)(g * w2[0]);                                                       \
            acc1 += (acc_t)(g * w2[1]);                                                       \
        }                                                                                     \
    }

    const int nblk = CH / WAVE;
    const int ngrp = WAVE / WALK_N;                 // groups per 64-entry record block (even)
    float ga[WALK_N], gb[WALK_N];
    Rec rec = fetch(0, PAD_XY);
    LOAD_HALF(ga, rec, 0)
#pragma unroll 1
    for (int blk = 0; blk < nblk; ++blk) {
        if ((unsigned)__builtin_amdgcn_readfirstlane((int)rec.xy) == PAD_XY) break;   // sorted: everything from here is padding
        last_xy = (unsigned)__builtin_amdgcn_readlane((int)rec.xy, WAVE - 1);
        Rec nxt = rec;
        if (blk + 1 < nblk) nxt = fetch(blk + 1, last_xy);
        const int ebase = e0 + blk * WAVE;
        const unsigned fresh_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)rec.fresh);
        const unsigned fresh_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(rec.fresh >> 32));
        const unsigned adj_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)rec.adjacent);
        const unsigned adj_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(rec.adjacent >> 32));
#pragma unroll
        for (int g2 = 0; g2 < ngrp; g2 += 2) {
            LOAD_HALF(gb, rec, g2 + 1)
            WALK_HALF(ga, rec, g2, ebase)
            if (g2 + 2 < ngrp) {
                LOAD_HALF(ga, rec, g2 + 2)
            } else if (blk + 1 < nblk) {
                LOAD_HALF(ga, nxt, 0)
            }
            WALK_HALF(gb, rec, g2 + 1, ebase)
        }
        rec = nxt;
    }
#undef LOAD_HALF
#undef WALK_HALF
    flush(false);
    SSTAMP(5)
#if SC_STAMPS
    if (bid == 400 && threadIdx.x == 0)
        printf("scatter stamps (cycles): cells %llu | box %llu | zero+tickets %llu | scan %llu | placement %llu | walk %llu\n", st_acc[0], st_acc[1],
               st_acc[2], st_acc[3], st_acc[4], st_acc[5]);
#endif
}

// deterministic mode: float gradient += shadow * 2^-44, shadow cleared (it is all zero again for the next call)
__global__ __launch_bounds__(256) void scatter_fixed_to_float_kernel(float* __restrict__ grad, long long* __restrict__ shadow,
                                                                     int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const long long v = shadow[i];
        if (v != 0) {
            grad[i] += (float)((double)v * (1.0 / 17592186044416.0));
            shadow[i] = 0;
        }
    }
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
    const int n = R < SORT_MAX ? R : SORT_MAX;         // rays per chunk
    // (16-bit keys were for the 2-D Hilbert order of round 2: 8 bits per axis.  The azimuth keys are one-dimensional: 14 bits = two
    // bins per ray at 8192 rays, and the histogram's zero fill and scan - 63 us of this kernel at 5000 rays with 16 bits - shrink 4x)
    const int key_bits = n <= 1024 ? 12 : 14;
    static const int max_cams = env_int("ESLAM_RAY_ORDER_CAMERAS", 32);      // A/B switch: 0 = several origins always get the Morton order
    hipLaunchKernelGGL(ray_order_kernel, dim3(chunks, ESLAM_RAY_ORDERS), dim3(1024), ((size_t)1 << key_bits) / 2 * sizeof(unsigned), st,
                       rays_o, rays_d, R, perm, key_bits, max_cams);
    return eslam_check_launch("ray_order_kernel");
}

// perm: ray order to bundle by (render mode), or NULL for the given order
// Samples per workgroup (*bm_out).  2048 (512 threads x 4) halve the cell flushes of 1024; a batch whose 2048-sample bundles
// make fewer than SC_SMALL_WGS workgroups (most CUs would idle while a few walk 256 entries per wave) is cut into
// 1024-sample bundles walked by the same 512 threads (2 samples per thread, 128 sorted entries per wave): twice the
// workgroups, half the walk.  Measured (profiles/r03/c_*): 200 x 32 (84 workgroups) 50.9 -> 30.6 us; 1024 x 64 (384) 47.2 ->
// 50.6, 1024 x 96 (588) 55.1 -> 68.5, 2048 x 64 75.5 -> 89.7: from a few hundred workgroups on the kernel is throughput-bound
// and the extra cell flushes of the smaller bundles cost more than the shorter walk gains.  ESLAM_SC_BUNDLE = 2048 / 1024 forces one of the two; 256 = the old 256-thread form of 1024
// (profiling only).
#ifndef SC_SMALL_WGS
#define SC_SMALL_WGS 192
#endif
static int scatter_bundle_size(int per, int nunits, int* bm_out) {
    static const int forced = env_int("ESLAM_SC_BUNDLE", 0);
    int bm = forced == 1024 || forced == 2048 || forced == 256 ? forced : 0;
    if (eslam_deterministic()) bm = 2048;              // (the fixed-point kernel is instantiated for 2048 only)
    if (bm == 0) {
        const int big = 2048 / per;
        const int nb = big > 0 ? (nunits + big - 1) / big : nunits;
        bm = (nb * NPL <= SC_SMALL_WGS && 1024 / per >= 1) ? 1024 : 2048;
    }
    *bm_out = bm;
    return (bm == 2048 ? 2048 : 1024) / per;
}

// whether the scatter launch of this mode can also run the decoder-gradient slab reduction (the production render path: one
// kernel, XCD-mapped 1-D grid, 512 threads); the stand-alone dec_grad_reduce_kernel covers the rest
bool eslam_scatter_can_reduce(bool render) {
    static const int xcd_map = env_int("ESLAM_SC_XCDMAP", 4), dbg_mode = env_int("ESLAM_SC_MODE", 0), bm = env_int("ESLAM_SC_BUNDLE", 0);
    static const int off = env_int("ESLAM_SC_NO_REDUCE", 0);
    return render && !eslam_deterministic() && xcd_map && dbg_mode == 0 && bm != 256 && !off;
}

int eslam_scatter_v2(const eslam_plane_t* planes, const Bound& bnd, const float* rays_o, const float* rays_d,
                     const float* z_or_pts, int64_t R, int S, bool render, const float* g_feat, const int* perm,
                     hipStream_t st, const DecReduceArgs* red) {
    PlaneSet ps;
    for (int i = 0; i < NPL; ++i) {
        ps.p[i] = planes[i];

    }
    const int64_t N = render ? R * S : R;
    const int nunits = render ? (int)R : (int)((N + 63) / 64);
    // the walk addresses g_feat rows and plane texels with 32-bit byte offsets from a uniform base
    if (N * 512 >= ((int64_t)1 << 32)) {
        eslam_set_error("scatter: %lld points exceed the 32-bit offset range of the feature-gradient buffer (8.3 M): split "
                        "the batch", (long long)N);
        return 1;
    }
    for (int i = 0; i < NPL; ++i) {
        const int64_t extent = (ESLAM_C_DIM - 1) * planes[i].stride_c + (int64_t)(planes[i].h - 1) * planes[i].stride_y +
                               (int64_t)(planes[i].w - 1) * planes[i].stride_x + 1;
        if (extent >= ((int64_t)1 << 30)) {
            eslam_set_error("scatter: plane %d spans %lld elements, limit 2^30", i, (long long)extent);
            return 1;
        }
    }
    static const int nosort = env_int("ESLAM_SC_NOSORT", 0), dbg_mode = env_int("ESLAM_SC_MODE", 0);       // A/B switch for profiling only
    if (nosort) perm = nullptr;
    const int per = render ? S : 64;
    // 2048 samples per workgroup (512 threads) halve the number of cell flushes of 1024; S up to 256 -> >= 8 rays
    int bm;
    const int bundle = scatter_bundle_size(per, nunits, &bm);
    for (int i = 0; i < NPL; ++i)
        if ((int64_t)planes[i].w * planes[i].h >= ((1 << 21) - 2)) {
            eslam_set_error("scatter: plane %d has %d x %d cells, limit 2^21", i, planes[i].h, planes[i].w);
            return 1;
        }
    static const int counting = env_int("ESLAM_SC_COUNTING", 1);   // A/B switch: 0 = always the bitonic network
    static const int xcd_map = env_int("ESLAM_SC_XCDMAP", 4);      // A/B switch: 0 = plain (bundle, plane) grid
    const int nbundles = (nunits + bundle - 1) / bundle;
    dim3 grid(nbundles, NPL);
    if (xcd_map == 3 || xcd_map == 4) grid = dim3(nbundles * NPL, 1);
    else if (xcd_map == 1) grid = dim3(((nbundles * 4 + 7) / 8) * 8 * 3, 1);
    DecReduceArgs red_args = {};
    int red_blocks = 0;
    if (red) {
        if (!eslam_scatter_can_reduce(render)) {
            eslam_set_error("scatter: cannot carry the decoder-gradient reduction in this mode");
            return 1;
        }
        red_args = *red;
        red_blocks = (2 * DEC_RED_COLBLOCKS + 7) / 8 * 8;
        grid.x += red_blocks;
    }
#define LAUNCH_SC(RD, DB, NTv, PERM, SS, ...)                                                                              \
    hipLaunchKernelGGL((scatter_sort_kernel<RD, DB, NTv, false, ##__VA_ARGS__>), grid, dim3(NTv), 0, st, ps, bnd, rays_o, rays_d, z_or_pts, \
                       PERM, (int)R, SS, g_feat, bundle, counting, nbundles, xcd_map, (long long*)nullptr, ShadowOff{}, \
                       red_args, red_blocks)
    if (eslam_deterministic()) {
        // fixed-point scatter into the int64 shadow, then shadow -> float gradients (DET in the kernel's header comment)
        ShadowOff so;
        int64_t total = 0;
        for (int i = 0; i < NPL; ++i) {
            const int64_t numel = (int64_t)ESLAM_C_DIM * planes[i].h * planes[i].w;
            const int64_t extent = (ESLAM_C_DIM - 1) * planes[i].stride_c + (int64_t)(planes[i].h - 1) * planes[i].stride_y +
                                   (int64_t)(planes[i].w - 1) * planes[i].stride_x + 1;
            if (extent != numel) {
                eslam_set_error("scatter (deterministic mode): plane %d is not dense", i);
                return 1;
            }
            so.o[i] = total;
            total += numel;
        }
        // One shadow per DEVICE (the pointer is a device allocation of the device that is current now).  A shadow that has
        // been handed to a launch is never freed: a hipGraph captured earlier may still hold its address, so a larger scene
        // gets a new allocation and the old one stays (a few scenes per process at most).
        constexpr int MAXDEV = 64;
        static long long* shadows[MAXDEV];
        static int64_t shadow_ns[MAXDEV];
        static std::mutex shadow_mu;
        int devi = 0;
        if (hipGetDevice(&devi) != hipSuccess || devi < 0 || devi >= MAXDEV) {
            eslam_set_error("scatter (deterministic mode): no usable current device");
            return 2;
        }
        long long* shadow;
        {
            std::lock_guard<std::mutex> lk(shadow_mu);
            if (shadow_ns[devi] < total) {       // first use (an eager warm-up call; hipMalloc cannot be captured into a graph)
                hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
                (void)hipStreamIsCapturing(st, &cs);
                if (cs != hipStreamCaptureStatusNone) {
                    eslam_set_error("scatter (deterministic mode): run one eager iteration before capturing a graph");
                    return 1;
                }
                long long* fresh = nullptr;
                if (hipMalloc(&fresh, (size_t)total * 8) != hipSuccess || hipMemset(fresh, 0, (size_t)total * 8) != hipSuccess) {
                    eslam_set_error("scatter (deterministic mode): cannot allocate the %lld-element shadow", (long long)total);
                    return 2;
                }
                shadows[devi] = fresh;           // (the previous, smaller one is deliberately kept alive: see above)
                shadow_ns[devi] = total;
            }
            shadow = shadows[devi];
        }
        if (render)
            hipLaunchKernelGGL((scatter_sort_kernel<true, 0, 512, true>), grid, dim3(512), 0, st, ps, bnd, rays_o, rays_d,
                               z_or_pts, perm, (int)R, S, g_feat, bundle, counting, nbundles, xcd_map, shadow, so,
                               DecReduceArgs{}, 0);
        else
            hipLaunchKernelGGL((scatter_sort_kernel<false, 0, 512, true>), grid, dim3(512), 0, st, ps, bnd, rays_o, rays_d,
                               z_or_pts, (const int*)nullptr, (int)R, 64, g_feat, bundle, counting, nbundles, xcd_map,
                               shadow, so, DecReduceArgs{}, 0);
        if (int rc = eslam_check_launch("scatter_sort_kernel<det>")) return rc;
        for (int i = 0; i < NPL; ++i) {
            const int64_t numel = (int64_t)ESLAM_C_DIM * planes[i].h * planes[i].w;
            hipLaunchKernelGGL(scatter_fixed_to_float_kernel, dim3((unsigned)((numel + 255) / 256 < 2048 ? (numel + 255) / 256 : 2048)),
                               dim3(256), 0, st, planes[i].grad, shadow + so.o[i], numel);
        }
        return eslam_check_launch("scatter_fixed_to_float_kernel");
    }
    if (render) {
        if (dbg_mode == 1) LAUNCH_SC(true, 1, 512, perm, S);
        else if (dbg_mode == 2) LAUNCH_SC(true, 2, 512, perm, S);
        else if (dbg_mode == 3) LAUNCH_SC(true, 3, 512, perm, S);
        else if (dbg_mode == 4) LAUNCH_SC(true, 4, 512, perm, S);
        else if (dbg_mode == 5) LAUNCH_SC(true, 5, 512, perm, S);
        else if (dbg_mode == 6) LAUNCH_SC(true, 6, 512, perm, S);
        else if (bm == 256) LAUNCH_SC(true, 0, 256, perm, S);
        else if (bm == 1024) LAUNCH_SC(true, 0, 512, perm, S, 2);
        else LAUNCH_SC(true, 0, 512, perm, S);
    } else {
        if (bm == 1024) LAUNCH_SC(false, 0, 512, (const int*)nullptr, 64, 2);
        else LAUNCH_SC(false, 0, 512, (const int*)nullptr, 64);
    }
#undef LAUNCH_SC
    return eslam_check_launch("scatter_sort_kernel");
}

bool eslam_planes_channels_last(const eslam_plane_t* planes, int first, int count);
int eslam_validate_planes(const eslam_plane_t* planes, int first, int count);

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
