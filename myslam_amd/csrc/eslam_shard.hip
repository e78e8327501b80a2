// Ray-sharded mapping iteration (new: the reference is single-GPU), the pieces that let a rank go
// from its rays to ONE gradient all-reduce without a collective or a host round trip in between:
//
//   mark_rays_kernel        which texels CAN receive gradient from the iteration's WHOLE batch, from ray geometry alone (every
//                           rank holds all rays of the iteration: same get_samples draw): a conservative superset of what the
//                           ranks' scatters touch, identical on every rank, known before anything is sampled or rendered
//   blocks_count / _compact the marked texels as an ascending index list + its length, on the device
//   blocks_move_dev_kernel  gather / scatter / clear of the listed 128-byte blocks with the length read on the device: the
//                           exchange buffer has a fixed capacity, only [tail | count x 32] floats of it are all-reduced
//
// (Round 2 agreed on the union with an int32 all-reduce between forward and backward and learnt its size through nonzero() on
// the host: two collectives, three eager launches and a host wait per iteration - 79 % overhead on a 1024 x 96 shard.)
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
__global__ __launch_bounds__(256) void mark_rays_kernel(const PlaneSet planes, const Bound bnd, const float* __restrict__ rays_o,
                                                        const float* __restrict__ rays_d, const float* __restrict__ gt_depth,
                                                        int R, float c15, const BlockBase32 base, uint8_t* __restrict__ touched) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int ray = blockIdx.x * 4 + wave; ray < R; ray += gridDim.x * 4) {
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

// ---- ascending list of the marked blocks -------------------------------------------------------------------
#define CHUNK_BYTES 4096          // bytes of `touched` per workgroup: 256 threads x 16
__device__ __forceinline__ unsigned nonzero_bytes16(const uint8_t* __restrict__ touched, int64_t n, int64_t first, unsigned& mask) {
    // mask: bit i = byte first + i is non-zero (i < 16, inside [0, n))
    mask = 0u;
    if (first + 16 <= n) {
        const uint4 v = *(const uint4*)(touched + first);
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int b = 0; b < 4; ++b)
                if ((w[q] >> (8 * b)) & 0xFFu) mask |= 1u << (4 * q + b);
    } else {
        for (int i = 0; i < 16 && first + i < n; ++i)
            if (touched[first + i]) mask |= 1u << i;
    }
    return (unsigned)__builtin_popcount(mask);
}

__global__ __launch_bounds__(256) void blocks_count_kernel(const uint8_t* __restrict__ touched, int64_t n,
                                                           unsigned* __restrict__ chunk_counts) {
    __shared__ unsigned ws[4];
    unsigned mask;
    const unsigned c = nonzero_bytes16(touched, n, (int64_t)blockIdx.x * CHUNK_BYTES + threadIdx.x * 16, mask);
    const unsigned t = (unsigned)__builtin_amdgcn_readlane((int)wave_incl_sum_u(c), WAVE - 1);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) chunk_counts[blockIdx.x] = (ws[0] + ws[1]) + (ws[2] + ws[3]);
}

// idx_out [capacity = n]: ascending indices of the non-zero bytes; meta [0] = their number, meta [1] += 1 (a stamp the host
// compares with its own iteration count after the copy to pinned memory: the number is this iteration's)
__global__ __launch_bounds__(256) void blocks_compact_kernel(const uint8_t* __restrict__ touched, int64_t n,
                                                             const unsigned* __restrict__ chunk_counts, int nchunks,
                                                             int32_t* __restrict__ idx_out, int32_t* __restrict__ meta) {
    __shared__ unsigned ws[4], base_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned before = 0;
    for (int c = threadIdx.x; c < (int)blockIdx.x; c += 256) before += chunk_counts[c];
    const unsigned bt = (unsigned)__builtin_amdgcn_readlane((int)wave_incl_sum_u(before), WAVE - 1);
    if (lane == 0) ws[wave] = bt;
    __syncthreads();
    if (threadIdx.x == 0) base_s = (ws[0] + ws[1]) + (ws[2] + ws[3]);
    __syncthreads();
    const int64_t first = (int64_t)blockIdx.x * CHUNK_BYTES + threadIdx.x * 16;
    unsigned mask;
    const unsigned c = nonzero_bytes16(touched, n, first, mask);
    const unsigned incl = wave_incl_sum_u(c);
    __syncthreads();
    if (lane == WAVE - 1) ws[wave] = incl;
    __syncthreads();
    unsigned pos = base_s + incl - c;
    for (int w = 0; w < wave; ++w) pos += ws[w];
    while (mask) {
        const int b = __builtin_ctz(mask);
        mask &= mask - 1u;
        idx_out[pos++] = (int32_t)(first + b);
    }
    if ((int)blockIdx.x == nchunks - 1 && threadIdx.x == 255) {
        meta[0] = (int32_t)pos;            // the last thread of the last chunk ends at the total
        meta[1] += 1;
    }
}

// MODE 0: buf <- [tail | listed blocks of flat]; 1: the inverse; 2: zero the listed blocks of flat and the tail.
// buf = [tail_pad floats: the dense tail (decoder / beta / pose gradients, loss sums), zero padded | 32 floats per block]
template <int MODE>
__global__ __launch_bounds__(256) void blocks_move_dev_kernel(float* __restrict__ flat, const int32_t* __restrict__ idx,
                                                              const int32_t* __restrict__ meta, float* __restrict__ tail,
                                                              int n_tail, int tail_pad, float* __restrict__ buf) {
    const int n_idx = meta[0];
    const int sub = threadIdx.x & 7;
    for (int i = blockIdx.x * 32 + (threadIdx.x >> 3); i < n_idx; i += gridDim.x * 32) {
        float4_t* a = (float4_t*)(flat + (int64_t)idx[i] * 32 + sub * 4);
        if (MODE == 2) { *a = (float4_t){0.f, 0.f, 0.f, 0.f}; continue; }
        float4_t* b = (float4_t*)(buf + tail_pad + (int64_t)i * 32 + sub * 4);
        if (MODE == 0) *b = *a;
        else *a = *b;
    }
    for (int i = blockIdx.x * 256 + threadIdx.x; i < tail_pad; i += gridDim.x * 256) {
        if (MODE == 0) buf[i] = i < n_tail ? tail[i] : 0.0f;
        else if (MODE == 1) { if (i < n_tail) tail[i] = buf[i]; }
        else if (i < n_tail) tail[i] = 0.0f;
    }
}

// ---- host side ----------------------------------------------------------------------------------------------
static int planes_channels_last_exact(const eslam_plane_t* planes, const char* who) {
    for (int i = 0; i < NPL; ++i)
        if (planes[i].stride_c != 1 || planes[i].stride_x != ESLAM_C_DIM || planes[i].stride_y != (int64_t)ESLAM_C_DIM * planes[i].w) {
            eslam_set_error("%s: plane %d is not channels_last (one block = one texel's 32 channels)", who, i);
            return 1;
        }
    return 0;
}

extern "C" int eslam_mark_rays(const eslam_plane_t* planes, const float* bound6_host, const float* rays_o, const float* rays_d,
                               const float* gt_depth, int R, double truncation, const int64_t* block_base_host, int64_t n_blocks,
                               uint8_t* touched, eslam_stream_t stream) {
    if (!planes || !bound6_host || !block_base_host || !touched || n_blocks <= 0) {
        eslam_set_error("eslam_mark_rays: null argument");
        return 1;
    }
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(touched, 0, (size_t)n_blocks, st) != hipSuccess) {
        eslam_set_error("eslam_mark_rays: memset failed");
        return 2;
    }
    if (R <= 0) return 0;
    if (!rays_o || !rays_d || !gt_depth) {
        eslam_set_error("eslam_mark_rays: null ray argument");
        return 1;
    }
    if (planes_channels_last_exact(planes, "eslam_mark_rays")) return 1;
    PlaneSet ps;
    BlockBase32 bb;
    for (int i = 0; i < NPL; ++i) {
        ps.p[i] = planes[i];
        bb.b[i] = block_base_host[i];
        if (planes[i].h < 2 || planes[i].w < 2 || bb.b[i] < 0 || bb.b[i] + (int64_t)planes[i].h * planes[i].w > n_blocks) {
            eslam_set_error("eslam_mark_rays: block range of plane %d exceeds n_blocks", i);
            return 1;
        }
    }
    Bound bnd;
    for (int k = 0; k < 3; ++k) { bnd.lo[k] = bound6_host[2 * k]; bnd.hi[k] = bound6_host[2 * k + 1]; }
    const int nwg = (R + 3) / 4 < 2048 ? (R + 3) / 4 : 2048;
    hipLaunchKernelGGL(mark_rays_kernel, dim3(nwg), dim3(256), 0, st, ps, bnd, rays_o, rays_d, gt_depth, R,
                       (float)(1.5 * truncation), bb, touched);
    return eslam_check_launch("mark_rays_kernel");
}

extern "C" int64_t eslam_blocks_compact_scratch_words(int64_t n_blocks) {
    return n_blocks < 0 ? -1 : (n_blocks + CHUNK_BYTES - 1) / CHUNK_BYTES;
}

extern "C" int eslam_blocks_compact(const uint8_t* touched, int64_t n_blocks, uint32_t* scratch, int32_t* idx, int32_t* meta,
                                    eslam_stream_t stream) {
    if (!touched || !scratch || !idx || !meta || n_blocks <= 0 || n_blocks >= ((int64_t)1 << 31) || ((uintptr_t)touched & 15)) {
        eslam_set_error("eslam_blocks_compact: null / unaligned argument or n_blocks outside (0, 2^31)");
        return 1;
    }
    const int nchunks = (int)((n_blocks + CHUNK_BYTES - 1) / CHUNK_BYTES);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(blocks_count_kernel, dim3(nchunks), dim3(256), 0, st, touched, n_blocks, scratch);
    hipLaunchKernelGGL(blocks_compact_kernel, dim3(nchunks), dim3(256), 0, st, touched, n_blocks, scratch, nchunks, idx, meta);
    return eslam_check_launch("blocks_compact_kernel");
}

static int blocks_move_dev(int mode, float* flat, const int32_t* idx, const int32_t* meta, int64_t capacity, float* tail,
                           int64_t n_tail, int64_t tail_pad, float* buf, eslam_stream_t stream) {
    if (!flat || !idx || !meta || (n_tail > 0 && !tail) || (mode != 2 && !buf) || n_tail < 0 || tail_pad < n_tail || (tail_pad & 3) ||
        capacity <= 0 || tail_pad >= ((int64_t)1 << 30) || ((uintptr_t)flat & 15) || ((uintptr_t)buf & 15)) {
        eslam_set_error("eslam_blocks_*_dev: null / unaligned argument, or tail_pad not a multiple of 4 floats >= n_tail");
        return 1;
    }
    const int64_t want = (capacity + 31) / 32;
    const dim3 grid((unsigned)(want < 2048 ? want : 2048));
    hipStream_t st = (hipStream_t)stream;
    if (mode == 0) hipLaunchKernelGGL(blocks_move_dev_kernel<0>, grid, dim3(256), 0, st, flat, idx, meta, tail, (int)n_tail, (int)tail_pad, buf);
    else if (mode == 1) hipLaunchKernelGGL(blocks_move_dev_kernel<1>, grid, dim3(256), 0, st, flat, idx, meta, tail, (int)n_tail, (int)tail_pad, buf);
    else hipLaunchKernelGGL(blocks_move_dev_kernel<2>, grid, dim3(256), 0, st, flat, idx, meta, tail, (int)n_tail, (int)tail_pad, buf);
    return eslam_check_launch("blocks_move_dev_kernel");
}

extern "C" int eslam_blocks_pack_dev(const float* flat, const int32_t* idx, const int32_t* meta, int64_t capacity, const float* tail,
                                     int64_t n_tail, int64_t tail_pad, float* buf, eslam_stream_t stream) {
    return blocks_move_dev(0, (float*)flat, idx, meta, capacity, (float*)tail, n_tail, tail_pad, buf, stream);
}
extern "C" int eslam_blocks_unpack_dev(float* flat, const int32_t* idx, const int32_t* meta, int64_t capacity, float* tail,
                                       int64_t n_tail, int64_t tail_pad, const float* buf, eslam_stream_t stream) {
    return blocks_move_dev(1, flat, idx, meta, capacity, tail, n_tail, tail_pad, (float*)buf, stream);
}
extern "C" int eslam_blocks_zero_dev(float* flat, const int32_t* idx, const int32_t* meta, int64_t capacity, float* tail,
                                     int64_t n_tail, eslam_stream_t stream) {
    return blocks_move_dev(2, flat, idx, meta, capacity, tail, n_tail, (n_tail + 3) & ~(int64_t)3, nullptr, stream);
}
