// Micro-benchmark (dev tool, not product): how the lane -> (point, 16-byte piece) assignment of the tri-plane gather
// changes the rate at which the texture addresser / L1 serve it.  Same points, same planes, same bytes, same FMAs:
//   map A  lane = 16 q + r   (the MFMA B-operand layout the forward kernel gathers in: 4 consecutive lanes = 4 POINTS)
//   map B  lane = 4 p + q    (4 consecutive lanes = the 4 pieces of ONE point's half line: 64 contiguous bytes per quad)
//   map D  lane = 8 p + o    (8 consecutive lanes = ONE texel's whole 128-B line; 8 points per instruction)
//   map E  (round 3) map B for the six FINE planes; the six COARSE planes (0.24 m cells: the 16 consecutive samples of a
//          ray that make a block cross 1-4 cells per axis) staged per block through LDS: the block's cell box - from the
//          cells of its first and last point, the samples of a ray being sorted along it - is fetched ONCE, 8 lanes per
//          texel line, written to a wave-private LDS image (144-byte texel pitch) and the 16 points interpolate from there
//          with ds_read_b128.  Boxes of more than ECAP texels fall back to the direct gather.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/ubench_gather tools/ubench_gather.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float float4_t __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct Plane { const float* data; int h, w; };
struct Planes { Plane p[12]; };

__device__ __forceinline__ void axis(float u, int n, int& i0, int& i1, float& t) {
    const float nm1 = (float)(n - 1);
    float x = ((u + 1.0f) * 0.5f) * nm1;
    x = fminf(fmaxf(x, 0.0f), nm1);
    const float f = floorf(x);
    i0 = (int)f; i1 = min(i0 + 1, n - 1); t = x - f;
}

// pts [N,3] normalised coordinates; out [N,128] features (d*64 + lvl*32 + channel)
#ifndef UB_WAVES
#define UB_WAVES 2                 // waves per SIMD the plain gather kernels are compiled for (-DUB_WAVES=3 / 4: occupancy study)
#endif
template <int MAP, int PIPE>
__global__ __launch_bounds__(256, UB_WAVES) void gather_kernel(const Planes planes, const float* __restrict__ pts, int N,
                                                        float* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tile = blockIdx.x * 4 + wave;           // 64 points per wave
    if (tile * 64 >= N) return;
    constexpr int PPI = (MAP == 2) ? 8 : 16;           // points per instruction
    constexpr int NCH = (MAP == 2) ? 4 : 8;            // channels per lane and level-plane
    int pl, piece;                                     // point-in-step, piece of the line
    if (MAP == 0) { pl = lane & 15; piece = lane >> 4; }
    else if (MAP == 1) { pl = lane >> 2; piece = lane & 3; }
    else { pl = lane >> 3; piece = lane & 7; }
    for (int b = 0; b < 64 / PPI; ++b) {
        const int pt = tile * 64 + b * PPI + pl;
        const float x = pts[pt * 3], y = pts[pt * 3 + 1], z = pts[pt * 3 + 2];
#pragma unroll 1
        for (int d = 0; d < 2; ++d) {
            float acc[2][NCH];
#pragma unroll
            for (int l = 0; l < 2; ++l)
#pragma unroll
                for (int i = 0; i < NCH; ++i) acc[l][i] = 0.f;
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const int o = k % 3, lvl = k / 3;
                const Plane& P = planes.p[2 * (3 * d + o) + lvl];
                const float u = (o == 2) ? y : x, v = (o == 0) ? y : z;
                int x0, x1, y0, y1; float tx, ty;
                axis(u, P.w, x0, x1, tx); axis(v, P.h, y0, y1, ty);
                const float w00 = (1 - tx) * (1 - ty), w01 = tx * (1 - ty), w10 = (1 - tx) * ty, w11 = tx * ty;
                const unsigned t00 = (y0 * P.w + x0) * 32u, t01 = (y0 * P.w + x1) * 32u, t10 = (y1 * P.w + x0) * 32u, t11 = (y1 * P.w + x1) * 32u;
                if (MAP == 2) {
                    const unsigned c = 4u * piece;
                    const float4_t a00 = *(const float4_t*)(P.data + t00 + c), a01 = *(const float4_t*)(P.data + t01 + c);
                    const float4_t a10 = *(const float4_t*)(P.data + t10 + c), a11 = *(const float4_t*)(P.data + t11 + c);
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[lvl][i] += a00[i] * w00 + a01[i] * w01 + a10[i] * w10 + a11[i] * w11;
                } else {
                    // MAP 0: the kernel's own split (channels 8q..8q+7: bytes 32q and 32q+16); MAP 1: contiguous halves
                    const unsigned ca = (MAP == 0) ? 8u * piece : 4u * piece, cb = (MAP == 0) ? 8u * piece + 4u : 16u + 4u * piece;
                    const float4_t a00 = *(const float4_t*)(P.data + t00 + ca), b00 = *(const float4_t*)(P.data + t00 + cb);
                    const float4_t a01 = *(const float4_t*)(P.data + t01 + ca), b01 = *(const float4_t*)(P.data + t01 + cb);
                    const float4_t a10 = *(const float4_t*)(P.data + t10 + ca), b10 = *(const float4_t*)(P.data + t10 + cb);
                    const float4_t a11 = *(const float4_t*)(P.data + t11 + ca), b11 = *(const float4_t*)(P.data + t11 + cb);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        acc[lvl][i] += a00[i] * w00 + a01[i] * w01 + a10[i] * w10 + a11[i] * w11;
                        acc[lvl][4 + i] += b00[i] * w00 + b01[i] * w01 + b10[i] * w10 + b11[i] * w11;
                    }
                }
                if (PIPE == 0) __builtin_amdgcn_sched_barrier(0);
            }
            float* dst = out + (size_t)pt * 128 + d * 64;
#pragma unroll
            for (int l = 0; l < 2; ++l) {
                if (MAP == 2) {
                    *(float4_t*)(dst + l * 32 + 4 * piece) = (float4_t){acc[l][0], acc[l][1], acc[l][2], acc[l][3]};
                } else {
                    const int ca = (MAP == 0) ? 8 * piece : 4 * piece, cb = (MAP == 0) ? 8 * piece + 4 : 16 + 4 * piece;
                    *(float4_t*)(dst + l * 32 + ca) = (float4_t){acc[l][0], acc[l][1], acc[l][2], acc[l][3]};
                    *(float4_t*)(dst + l * 32 + cb) = (float4_t){acc[l][4], acc[l][5], acc[l][6], acc[l][7]};
                }
            }
        }
    }
}


#ifndef ECAP
#define ECAP 16                    // texels of a staged box (2 load instructions of 8 texels)
#endif
#define EPITCH 36                  // floats per staged texel: 144 bytes, conflict-free for texels at the same piece

template <int STAGE>               // 0: never stage (= map B, same code path otherwise); 1: stage coarse planes
__global__ __launch_bounds__(256, 2) void gather_staged_kernel(const Planes planes, const float* __restrict__ pts, int N,
                                                               float* __restrict__ out, int* __restrict__ stats) {
    __shared__ __attribute__((aligned(16))) float stage[4][3][ECAP * EPITCH];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tile = blockIdx.x * 4 + wave;
    if (tile * 64 >= N) return;
    const int pl = lane >> 2, piece = lane & 3;
    const int tl = lane >> 3, tp = lane & 7;          // staging role: texel-in-instruction, 16-byte piece of its line
    int n_staged = 0, n_direct = 0, n_tex = 0;
    for (int b = 0; b < 4; ++b) {
        const int pt = tile * 64 + b * 16 + pl;
        const float x = pts[pt * 3], y = pts[pt * 3 + 1], z = pts[pt * 3 + 2];
        // the block's first and last point (lanes 0 and 60): cells along each axis are monotone between them
        const float xf = __shfl(x, 0, 64), yf = __shfl(y, 0, 64), zf = __shfl(z, 0, 64);
        const float xl = __shfl(x, 60, 64), yl = __shfl(y, 60, 64), zl = __shfl(z, 60, 64);
#pragma unroll 1
        for (int d = 0; d < 2; ++d) {
            float acc[2][8];
#pragma unroll
            for (int l = 0; l < 2; ++l)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[l][i] = 0.f;
            // ---- coarse planes (lvl 0) ----
            bool staged[3];
            int blo_u[3], blo_v[3], bw[3];
#pragma unroll
            for (int o = 0; o < 3; ++o) {
                const Plane& P = planes.p[2 * (3 * d + o)];
                const float uf = (o == 2) ? yf : xf, vf = (o == 0) ? yf : zf, ul = (o == 2) ? yl : xl, vl = (o == 0) ? yl : zl;
                int a0, a1, b0, b1, c0, c1, d0, d1; float t_;
                axis(uf, P.w, a0, a1, t_); axis(ul, P.w, b0, b1, t_);
                axis(vf, P.h, c0, c1, t_); axis(vl, P.h, d0, d1, t_);
                const int lo_u = min(a0, b0), hi_u = max(a1, b1), lo_v = min(c0, d0), hi_v = max(c1, d1);
                const int w_ = hi_u - lo_u + 1, h_ = hi_v - lo_v + 1, T = w_ * h_;
                blo_u[o] = lo_u; blo_v[o] = lo_v; bw[o] = w_;
                staged[o] = STAGE && T <= ECAP;
                if (staged[o]) {
                    const float rw = 1.0f / (float)w_;
                    float* L = stage[wave][o];
#pragma unroll
                    for (int i = 0; i < ECAP / 8; ++i) {
                        const int t = tl + 8 * i;
                        if (t < T) {
                            const int ty = (int)(((float)t + 0.5f) * rw), tx = t - ty * w_;
                            const float4_t v = *(const float4_t*)(P.data + ((lo_v + ty) * P.w + lo_u + tx) * 32u + 4u * tp);
                            *(float4_t*)(L + t * EPITCH + 4 * tp) = v;
                        }
                    }
                    n_staged++; n_tex += T;
                } else n_direct++;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int o = 0; o < 3; ++o) {
                const Plane& P = planes.p[2 * (3 * d + o)];
                const float u = (o == 2) ? y : x, v = (o == 0) ? y : z;
                int x0, x1, y0, y1; float tx, ty;
                axis(u, P.w, x0, x1, tx); axis(v, P.h, y0, y1, ty);
                const float w00 = (1 - tx) * (1 - ty), w01 = tx * (1 - ty), w10 = (1 - tx) * ty, w11 = tx * ty;
                float4_t a00, b00, a01, b01, a10, b10, a11, b11;
                if (staged[o]) {
                    const float* L = stage[wave][o] + 4 * piece;
                    const int l00 = ((y0 - blo_v[o]) * bw[o] + (x0 - blo_u[o])) * EPITCH, l01 = l00 + (x1 - x0) * EPITCH;
                    const int l10 = l00 + (y1 - y0) * bw[o] * EPITCH, l11 = l10 + (x1 - x0) * EPITCH;
                    a00 = *(const float4_t*)(L + l00); b00 = *(const float4_t*)(L + l00 + 16);
                    a01 = *(const float4_t*)(L + l01); b01 = *(const float4_t*)(L + l01 + 16);
                    a10 = *(const float4_t*)(L + l10); b10 = *(const float4_t*)(L + l10 + 16);
                    a11 = *(const float4_t*)(L + l11); b11 = *(const float4_t*)(L + l11 + 16);
                } else {
                    const unsigned t00 = (y0 * P.w + x0) * 32u, t01 = (y0 * P.w + x1) * 32u, t10 = (y1 * P.w + x0) * 32u, t11 = (y1 * P.w + x1) * 32u;
                    const unsigned ca = 4u * piece, cb = 16u + 4u * piece;
                    a00 = *(const float4_t*)(P.data + t00 + ca); b00 = *(const float4_t*)(P.data + t00 + cb);
                    a01 = *(const float4_t*)(P.data + t01 + ca); b01 = *(const float4_t*)(P.data + t01 + cb);
                    a10 = *(const float4_t*)(P.data + t10 + ca); b10 = *(const float4_t*)(P.data + t10 + cb);
                    a11 = *(const float4_t*)(P.data + t11 + ca); b11 = *(const float4_t*)(P.data + t11 + cb);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[0][i] += a00[i] * w00 + a01[i] * w01 + a10[i] * w10 + a11[i] * w11;
                    acc[0][4 + i] += b00[i] * w00 + b01[i] * w01 + b10[i] * w10 + b11[i] * w11;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // ---- fine planes (lvl 1): direct gather, map B ----
#pragma unroll
            for (int o = 0; o < 3; ++o) {
                const Plane& P = planes.p[2 * (3 * d + o) + 1];
                const float u = (o == 2) ? y : x, v = (o == 0) ? y : z;
                int x0, x1, y0, y1; float tx, ty;
                axis(u, P.w, x0, x1, tx); axis(v, P.h, y0, y1, ty);
                const float w00 = (1 - tx) * (1 - ty), w01 = tx * (1 - ty), w10 = (1 - tx) * ty, w11 = tx * ty;
                const unsigned t00 = (y0 * P.w + x0) * 32u, t01 = (y0 * P.w + x1) * 32u, t10 = (y1 * P.w + x0) * 32u, t11 = (y1 * P.w + x1) * 32u;
                const unsigned ca = 4u * piece, cb = 16u + 4u * piece;
                const float4_t a00 = *(const float4_t*)(P.data + t00 + ca), b00 = *(const float4_t*)(P.data + t00 + cb);
                const float4_t a01 = *(const float4_t*)(P.data + t01 + ca), b01 = *(const float4_t*)(P.data + t01 + cb);
                const float4_t a10 = *(const float4_t*)(P.data + t10 + ca), b10 = *(const float4_t*)(P.data + t10 + cb);
                const float4_t a11 = *(const float4_t*)(P.data + t11 + ca), b11 = *(const float4_t*)(P.data + t11 + cb);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[1][i] += a00[i] * w00 + a01[i] * w01 + a10[i] * w10 + a11[i] * w11;
                    acc[1][4 + i] += b00[i] * w00 + b01[i] * w01 + b10[i] * w10 + b11[i] * w11;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            float* dst = out + (size_t)pt * 128 + d * 64;
#pragma unroll
            for (int l = 0; l < 2; ++l) {
                *(float4_t*)(dst + l * 32 + 4 * piece) = (float4_t){acc[l][0], acc[l][1], acc[l][2], acc[l][3]};
                *(float4_t*)(dst + l * 32 + 16 + 4 * piece) = (float4_t){acc[l][4], acc[l][5], acc[l][6], acc[l][7]};
            }
        }
    }
    if (stats && lane == 0) { atomicAdd(&stats[0], n_staged); atomicAdd(&stats[1], n_direct); atomicAdd(&stats[2], n_tex); }
}

// E2: as E1, but (i) the boxes are fetched by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write), (ii) the
// geometry and the colour coarse plane of an orientation share ONE box (both have 0.24 m cells), and (iii) the boxes of block
// b + 1 are requested while block b works through its six fine planes, so that the only global latency a block still waits
// for is that of its 6 fine-plane steps (12 dependent steps in map B).  128-byte texel pitch in LDS (what the DMA writes).
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((address_space(1))) const void* glb_ptr_t;
struct Box { int lo_u, lo_v, w, T; };
// a wave-uniform zero the optimiser cannot fold: added to plane indices it keeps the planes' scalar loads inside the block loop
// (hoisted, ~100 loop-invariant plane scalars overflow the SGPR file and spill; eslam_common.h opaque_zero)
__device__ __forceinline__ int opaque_zero_u(int v) {
    int z = __builtin_amdgcn_readfirstlane(v);
    asm volatile("s_and_b32 %0, %0, 0" : "+s"(z) : : "scc");
    return z;
}
__device__ __forceinline__ Box block_box(const Plane& P, float uf, float vf, float ul, float vl) {
    int a0, a1, b0, b1, c0, c1, d0, d1; float t_;
    axis(uf, P.w, a0, a1, t_); axis(ul, P.w, b0, b1, t_);
    axis(vf, P.h, c0, c1, t_); axis(vl, P.h, d0, d1, t_);
    Box b;
    b.lo_u = min(a0, b0); b.lo_v = min(c0, d0);
    b.w = max(a1, b1) - b.lo_u + 1;
    b.T = b.w * (max(c1, d1) - b.lo_v + 1);
    return b;
}
__global__ __launch_bounds__(256, 2) void gather_dma_kernel(const Planes planes, const float* __restrict__ pts, int N,
                                                            float* __restrict__ out, int* __restrict__ stats) {
    __shared__ __attribute__((aligned(16))) float stage[4][6][ECAP * 32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tile = blockIdx.x * 4 + wave;
    if (tile * 64 >= N) return;
    const int pl = lane >> 2, piece = lane & 3;
    const int tl = lane >> 3, tp = lane & 7;
    int n_staged = 0, n_direct = 0, n_tex = 0;
    // request the boxes of block b (its points' coordinates are loaded here, again: two extra point loads per block)
    auto request = [&](int b, Box box[3]) {
        const int oz = opaque_zero_u(b);
        const int p0 = tile * 64 + b * 16;
        const float xf = pts[p0 * 3], yf = pts[p0 * 3 + 1], zf = pts[p0 * 3 + 2];
        const float xl = pts[(p0 + 15) * 3], yl = pts[(p0 + 15) * 3 + 1], zl = pts[(p0 + 15) * 3 + 2];
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            const Plane& P = planes.p[2 * o + oz];
            box[o] = block_box(P, (o == 2) ? yf : xf, (o == 0) ? yf : zf, (o == 2) ? yl : xl, (o == 0) ? yl : zl);
            if (box[o].T > ECAP) continue;                     // wave-uniform
            const float rw = 1.0f / (float)box[o].w;
#pragma unroll
            for (int i = 0; i < ECAP / 8; ++i) {
                if (8 * i >= box[o].T) break;                  // wave-uniform
                const int t = min(tl + 8 * i, box[o].T - 1);
                const int ty = (int)(((float)t + 0.5f) * rw), tx = t - ty * box[o].w;
                const unsigned off = ((box[o].lo_v + ty) * P.w + box[o].lo_u + tx) * 32u + 4u * tp;
#pragma unroll
                for (int d = 0; d < 2; ++d)
                    __builtin_amdgcn_global_load_lds((glb_ptr_t)(planes.p[2 * (3 * d + o) + oz].data + off),
                                                     (lds_ptr_t)(stage[wave][3 * d + o] + i * 256), 16, 0, 0);
            }
        }
    };
    Box box[3], nbox[3];
    request(0, box);
#pragma unroll 1
    for (int b = 0; b < 4; ++b) {
        const int oz = opaque_zero_u(b);
        const int pt = tile * 64 + b * 16 + pl;
        const float x = pts[pt * 3], y = pts[pt * 3 + 1], z = pts[pt * 3 + 2];
        float acc[2][2][8];
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int l = 0; l < 2; ++l)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[d][l][i] = 0.f;
        // the boxes of this block have landed (requested one block ago; every younger load has been consumed)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- coarse planes of both decoders from LDS ----
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            const Plane& P = planes.p[2 * o + oz];
            const float u = (o == 2) ? y : x, v = (o == 0) ? y : z;
            int x0, x1, y0, y1; float tx, ty;
            axis(u, P.w, x0, x1, tx); axis(v, P.h, y0, y1, ty);
            const float w00 = (1 - tx) * (1 - ty), w01 = tx * (1 - ty), w10 = (1 - tx) * ty, w11 = tx * ty;
            const bool st = box[o].T <= ECAP;
            if (st) { n_staged++; n_tex += box[o].T; } else n_direct++;
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                float4_t a00, b00, a01, b01, a10, b10, a11, b11;
                if (st) {
                    const float* L = stage[wave][3 * d + o] + 4 * piece;
                    const int l00 = ((y0 - box[o].lo_v) * box[o].w + (x0 - box[o].lo_u)) * 32, l01 = l00 + (x1 - x0) * 32;
                    const int l10 = l00 + (y1 - y0) * box[o].w * 32, l11 = l10 + (x1 - x0) * 32;
                    a00 = *(const float4_t*)(L + l00); b00 = *(const float4_t*)(L + l00 + 16);
                    a01 = *(const float4_t*)(L + l01); b01 = *(const float4_t*)(L + l01 + 16);
                    a10 = *(const float4_t*)(L + l10); b10 = *(const float4_t*)(L + l10 + 16);
                    a11 = *(const float4_t*)(L + l11); b11 = *(const float4_t*)(L + l11 + 16);
                } else {
                    const float* D = planes.p[2 * (3 * d + o) + oz].data;
                    const unsigned t00 = (y0 * P.w + x0) * 32u, t01 = (y0 * P.w + x1) * 32u, t10 = (y1 * P.w + x0) * 32u, t11 = (y1 * P.w + x1) * 32u;
                    const unsigned ca = 4u * piece, cb = 16u + 4u * piece;
                    a00 = *(const float4_t*)(D + t00 + ca); b00 = *(const float4_t*)(D + t00 + cb);
                    a01 = *(const float4_t*)(D + t01 + ca); b01 = *(const float4_t*)(D + t01 + cb);
                    a10 = *(const float4_t*)(D + t10 + ca); b10 = *(const float4_t*)(D + t10 + cb);
                    a11 = *(const float4_t*)(D + t11 + ca); b11 = *(const float4_t*)(D + t11 + cb);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[d][0][i] += a00[i] * w00 + a01[i] * w01 + a10[i] * w10 + a11[i] * w11;
                    acc[d][0][4 + i] += b00[i] * w00 + b01[i] * w01 + b10[i] * w10 + b11[i] * w11;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // every LDS read of this block's boxes has returned (their values were consumed above): the image may be overwritten
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (b + 1 < 4) request(b + 1, nbox);
        // ---- fine planes: direct gather, map B ----
#pragma unroll
        for (int d = 0; d < 2; ++d) {
#pragma unroll
            for (int o = 0; o < 3; ++o) {
                __builtin_amdgcn_sched_barrier(0);
                const Plane& P = planes.p[2 * (3 * d + o) + 1 + oz];
                const float u = (o == 2) ? y : x, v = (o == 0) ? y : z;
                int x0, x1, y0, y1; float tx, ty;
                axis(u, P.w, x0, x1, tx); axis(v, P.h, y0, y1, ty);
                const float w00 = (1 - tx) * (1 - ty), w01 = tx * (1 - ty), w10 = (1 - tx) * ty, w11 = tx * ty;
                const unsigned t00 = (y0 * P.w + x0) * 32u, t01 = (y0 * P.w + x1) * 32u, t10 = (y1 * P.w + x0) * 32u, t11 = (y1 * P.w + x1) * 32u;
                const unsigned ca = 4u * piece, cb = 16u + 4u * piece;
                const float4_t a00 = *(const float4_t*)(P.data + t00 + ca), b00 = *(const float4_t*)(P.data + t00 + cb);
                const float4_t a01 = *(const float4_t*)(P.data + t01 + ca), b01 = *(const float4_t*)(P.data + t01 + cb);
                const float4_t a10 = *(const float4_t*)(P.data + t10 + ca), b10 = *(const float4_t*)(P.data + t10 + cb);
                const float4_t a11 = *(const float4_t*)(P.data + t11 + ca), b11 = *(const float4_t*)(P.data + t11 + cb);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[d][1][i] += a00[i] * w00 + a01[i] * w01 + a10[i] * w10 + a11[i] * w11;
                    acc[d][1][4 + i] += b00[i] * w00 + b01[i] * w01 + b10[i] * w10 + b11[i] * w11;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            float* dst = out + (size_t)pt * 128 + d * 64;
#pragma unroll
            for (int l = 0; l < 2; ++l) {
                *(float4_t*)(dst + l * 32 + 4 * piece) = (float4_t){acc[d][l][0], acc[d][l][1], acc[d][l][2], acc[d][l][3]};
                *(float4_t*)(dst + l * 32 + 16 + 4 * piece) = (float4_t){acc[d][l][4], acc[d][l][5], acc[d][l][6], acc[d][l][7]};
            }
        }
#pragma unroll
        for (int o = 0; o < 3; ++o) box[o] = nbox[o];
    }
    if (stats && lane == 0) { atomicAdd(&stats[0], n_staged); atomicAdd(&stats[1], n_direct); atomicAdd(&stats[2], n_tex); }
}

static double urand(unsigned long long& s) { s = s * 6364136223846793005ull + 1442695040888963407ull; return (double)(s >> 11) / 9007199254740992.0; }

int main(int argc, char** argv) {
    const int R = argc > 1 ? atoi(argv[1]) : 4096, S = argc > 2 ? atoi(argv[2]) : 64;
    const int N = R * S;
    const float bound[3][2] = {{-1.9f, 7.94f}, {-2.2f, 4.52f}, {-2.5f, 2.54f}};
    const int dims[12][2] = {{27, 41}, {111, 164}, {21, 41}, {84, 164}, {21, 27}, {84, 111},
                             {27, 41}, {223, 328}, {21, 41}, {168, 328}, {21, 27}, {168, 223}};
    Planes P;
    size_t total = 0;
    for (int i = 0; i < 12; ++i) total += (size_t)dims[i][0] * dims[i][1] * 32;
    std::vector<float> hp(total);
    unsigned long long seed = 12345;
    for (auto& v : hp) v = (float)(urand(seed) - 0.5) * 0.02f;
    float* dplanes;
    CK(hipMalloc(&dplanes, total * 4));
    CK(hipMemcpy(dplanes, hp.data(), total * 4, hipMemcpyHostToDevice));
    size_t off = 0;
    for (int i = 0; i < 12; ++i) { P.p[i].data = dplanes + off; P.p[i].h = dims[i][0]; P.p[i].w = dims[i][1]; off += (size_t)dims[i][0] * dims[i][1] * 32; }
    // rays from the room centre, Replica pinhole (fx = fy = 600, 1200 x 680), depth U(0.5, 2.5), 56 + 8 samples
    std::vector<float> hpts((size_t)N * 3);
    const float c[3] = {(bound[0][0] + bound[0][1]) / 2, (bound[1][0] + bound[1][1]) / 2, (bound[2][0] + bound[2][1]) / 2};
    for (int r = 0; r < R; ++r) {
        const double px = urand(seed) * 1200, py = urand(seed) * 680, dep = 0.5 + 2.0 * urand(seed);
        const double dir[3] = {(px - 599.5) / 600.0, -(py - 339.5) / 600.0, -1.0};
        std::vector<double> zs(S);
        const int ns = S - 8;
        for (int s = 0; s < ns; ++s) zs[s] = 1.2 * dep * (s + urand(seed)) / ns;
        for (int s = 0; s < 8; ++s) zs[ns + s] = dep - 0.09 + 0.18 * (s + urand(seed)) / 8;
        std::sort(zs.begin(), zs.end());
        for (int s = 0; s < S; ++s)
            for (int a = 0; a < 3; ++a) {
                const double w = c[a] + dir[a] * zs[s];
                hpts[((size_t)r * S + s) * 3 + a] = (float)((w - bound[a][0]) / (bound[a][1] - bound[a][0]) * 2 - 1);
            }
    }
    float *dpts, *dout;
    CK(hipMalloc(&dpts, hpts.size() * 4));
    CK(hipMemcpy(dpts, hpts.data(), hpts.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&dout, (size_t)N * 128 * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int nblk = (N / 64 + 3) / 4;
    std::vector<float> ref((size_t)N * 128), got((size_t)N * 128);
    // dynamic LDS only caps the occupancy, so that the variants are compared at the same waves per SIMD
    const int lds = argc > 3 ? atoi(argv[3]) : 80 * 1024;
    printf("dynamic LDS %d B per workgroup -> at most %d waves per SIMD\n", lds, lds ? 160 * 1024 / lds : 8);
    auto run = [&](const char* name, auto kern, bool is_ref) {
        CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(nblk), dim3(256), lds, 0, P, dpts, N, dout);
        CK(hipDeviceSynchronize());
        float best = 1e9f, sum = 0;
        for (int i = 0; i < 20; ++i) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(kern, dim3(nblk), dim3(256), lds, 0, P, dpts, N, dout);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            best = fminf(best, ms); sum += ms;
        }
        CK(hipMemcpy(got.data(), dout, got.size() * 4, hipMemcpyDeviceToHost));
        double err = 0;
        if (is_ref) ref = got; else for (size_t i = 0; i < got.size(); ++i) err = fmax(err, fabs((double)got[i] - ref[i]));
        printf("%-34s  min %7.1f us  mean %7.1f us   %.2f TB/s of 16-B-per-lane requests   max|diff vs A| %.2e\n", name, best * 1e3,
               sum / 20 * 1e3, (double)N * 6144 / (best * 1e-3) / 1e12, err);
    };
    printf("gather of %d x %d points over the 12 room0 planes (%.1f MB), + %.0f MB of feature stores\n", R, S, total * 4 / 1e6, N * 512 / 1e6);
    run("A  lane = 16q + r   (kernel today)", gather_kernel<0, 0>, true);
    run("A  ..., loads free to hoist", gather_kernel<0, 1>, false);
    run("B  lane = 4p + q    (quad = 64 B)", gather_kernel<1, 0>, false);
    run("B  ..., loads free to hoist", gather_kernel<1, 1>, false);
    run("D  lane = 8p + o    (8 lanes = line)", gather_kernel<2, 0>, false);
    run("D  ..., loads free to hoist", gather_kernel<2, 1>, false);
    // map E: the staged kernel has its own (static) LDS; its signature carries a statistics pointer
    int* dstats;
    CK(hipMalloc(&dstats, 16));
    auto run_e = [&](const char* name, auto kern) {
        CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        const int lds_e = lds > 52 * 1024 ? lds - 52 * 1024 : 0;      // the static staging image counts towards the cap
        CK(hipMemset(dstats, 0, 16));
        hipLaunchKernelGGL(kern, dim3(nblk), dim3(256), lds_e, 0, P, dpts, N, dout, dstats);
        int hs[4];
        CK(hipMemcpy(hs, dstats, 16, hipMemcpyDeviceToHost));
        for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(nblk), dim3(256), lds_e, 0, P, dpts, N, dout, (int*)nullptr);
        CK(hipDeviceSynchronize());
        float best = 1e9f, sum = 0;
        for (int i = 0; i < 20; ++i) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(kern, dim3(nblk), dim3(256), lds_e, 0, P, dpts, N, dout, (int*)nullptr);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            best = fminf(best, ms); sum += ms;
        }
        CK(hipMemcpy(got.data(), dout, got.size() * 4, hipMemcpyDeviceToHost));
        double err = 0;
        for (size_t i = 0; i < got.size(); ++i) err = fmax(err, fabs((double)got[i] - ref[i]));
        printf("%-34s  min %7.1f us  mean %7.1f us   coarse (block, plane) pairs staged %d / direct %d, %.1f texels per staged box   max|diff vs A| %.2e\n",
               name, best * 1e3, sum / 20 * 1e3, hs[0], hs[1], hs[0] ? (double)hs[2] / hs[0] : 0.0, err);
    };
    run_e("E0 staged kernel, staging off", gather_staged_kernel<0>);
    run_e("E1 coarse planes staged via LDS", gather_staged_kernel<1>);
    run_e("E2 LDS-DMA boxes, one block ahead", gather_dma_kernel);
    return 0;
}
