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
#include "eslam_shard_dev.h"

__global__ __launch_bounds__(256) void mark_rays_kernel(const MarkArgs m) { mark_rays_block(m, blockIdx.x, gridDim.x); }

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

// idx_out [capacity = n]: ascending indices of the non-zero bytes; meta [0] = their number, meta [1] += 1 - a stamp: the host
// compares it with its own count of launches before it trusts the number.  host_meta (optional): device-visible pinned host
// memory that receives the same two words (number first, stamp last, system scope), so that no copy node is needed.
// clear: every workgroup zeroes its own chunk behind its read (the count pass is over, no other workgroup reads it): the next
// iteration's marking starts from a clean map without a memset node.
// (One launch with every workgroup counting the chunks in front of it itself was tried: 86 us instead of 5 + 6.)
__global__ __launch_bounds__(256) void blocks_compact_kernel(uint8_t* __restrict__ touched, int64_t n,
                                                             const unsigned* __restrict__ chunk_counts, int nchunks,
                                                             int32_t* __restrict__ idx_out, int32_t* __restrict__ meta,
                                                             int32_t* __restrict__ host_meta, int clear) {
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
    if (clear && mask) {
        if (first + 16 <= n) *(uint4*)(touched + first) = make_uint4(0u, 0u, 0u, 0u);
        else for (int i = 0; i < 16 && first + i < n; ++i) touched[first + i] = 0;
    }
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
        const int32_t stamp = meta[1] + 1;
        meta[0] = (int32_t)pos;            // the last thread of the last chunk ends at the total
        meta[1] = stamp;
        if (host_meta) {
            __hip_atomic_store(host_meta, (int32_t)pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(host_meta + 1, stamp, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// MODE 0: buf <- [tail | listed blocks of flat]; 1: the inverse; 2: zero the listed blocks of flat and the tail.
// buf = [tail_pad floats: the dense tail (decoder / beta / pose gradients, loss sums), zero padded | 32 floats per block]
template <int MODE>
__global__ __launch_bounds__(256) void blocks_move_dev_kernel(float* __restrict__ flat, const int32_t* __restrict__ idx,
                                                              const int32_t* __restrict__ meta, float* __restrict__ tail,
                                                              int n_tail, int tail_pad, float* __restrict__ buf,
                                                              uint32_t* __restrict__ step_bump) {
    // the last launch of a ray-sharded iteration (MODE 0) advances the iteration's random-number step: the sampler and the
    // set-size replay read it on different streams, so nothing may write it while either can still run
    if (step_bump && blockIdx.x == 0 && threadIdx.x == 0) step_bump[0] += 1u;
    if (MODE == 2) {
        clear_blocks_block(ClearArgs{flat, idx, meta, tail, n_tail}, blockIdx.x, gridDim.x);
        return;
    }
    const int n_idx = meta[0];
    const int sub = threadIdx.x & 7;
    for (int i = blockIdx.x * 32 + (threadIdx.x >> 3); i < n_idx; i += gridDim.x * 32) {
        float4_t* a = (float4_t*)(flat + (int64_t)idx[i] * 32 + sub * 4);
        float4_t* b = (float4_t*)(buf + tail_pad + (int64_t)i * 32 + sub * 4);
        if (MODE == 0) *b = *a;
        else *a = *b;
    }
    for (int i = blockIdx.x * 256 + threadIdx.x; i < tail_pad; i += gridDim.x * 256) {
        if (MODE == 0) buf[i] = i < n_tail ? tail[i] : 0.0f;
        else if (i < n_tail) tail[i] = buf[i];
    }
}

// ---- host side ----------------------------------------------------------------------------------------------
// fills m from the C-ABI arguments; 0 = ok
int eslam_shard_mark_args(const char* who, const eslam_plane_t* planes, const float* bound6_host, const float* rays_o,
                          const float* rays_d, const float* gt_depth, int R, double truncation, const int64_t* block_base_host,
                          int64_t n_blocks, uint8_t* touched, MarkArgs* m) {
    if (!planes || !bound6_host || !block_base_host || !touched || n_blocks <= 0 || !rays_o || !rays_d || !gt_depth || R <= 0) {
        eslam_set_error("%s: null argument (texel marking)", who);
        return 1;
    }
    for (int i = 0; i < NPL; ++i) {
        if (planes[i].stride_c != 1 || planes[i].stride_x != ESLAM_C_DIM || planes[i].stride_y != (int64_t)ESLAM_C_DIM * planes[i].w) {
            eslam_set_error("%s: plane %d is not channels_last (one block = one texel's 32 channels)", who, i);
            return 1;
        }
        m->planes.p[i] = planes[i];
        m->base.b[i] = block_base_host[i];
        if (planes[i].h < 2 || planes[i].w < 2 || block_base_host[i] < 0 ||
            block_base_host[i] + (int64_t)planes[i].h * planes[i].w > n_blocks) {
            eslam_set_error("%s: block range of plane %d exceeds n_blocks", who, i);
            return 1;
        }
    }
    for (int k = 0; k < 3; ++k) { m->bnd.lo[k] = bound6_host[2 * k]; m->bnd.hi[k] = bound6_host[2 * k + 1]; }
    m->rays_o = rays_o; m->rays_d = rays_d; m->gt_depth = gt_depth; m->R = R;
    m->c15 = (float)(1.5 * truncation);
    m->touched = touched;
    return 0;
}

extern "C" int eslam_mark_rays(const eslam_plane_t* planes, const float* bound6_host, const float* rays_o, const float* rays_d,
                               const float* gt_depth, int R, double truncation, const int64_t* block_base_host, int64_t n_blocks,
                               uint8_t* touched, eslam_stream_t stream) {
    if (!touched || n_blocks <= 0) {
        eslam_set_error("eslam_mark_rays: null argument");
        return 1;
    }
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(touched, 0, (size_t)n_blocks, st) != hipSuccess) {
        eslam_set_error("eslam_mark_rays: memset failed");
        return 2;
    }
    if (R <= 0) return 0;
    MarkArgs m;
    if (int rc = eslam_shard_mark_args("eslam_mark_rays", planes, bound6_host, rays_o, rays_d, gt_depth, R, truncation,
                                       block_base_host, n_blocks, touched, &m))
        return rc;
    const int nwg = (R + 3) / 4 < 2048 ? (R + 3) / 4 : 2048;
    hipLaunchKernelGGL(mark_rays_kernel, dim3(nwg), dim3(256), 0, st, m);
    return eslam_check_launch("mark_rays_kernel");
}

extern "C" int64_t eslam_blocks_compact_scratch_words(int64_t n_blocks) {
    return n_blocks < 0 ? -1 : (n_blocks + CHUNK_BYTES - 1) / CHUNK_BYTES;
}

extern "C" int eslam_blocks_compact(uint8_t* touched, int64_t n_blocks, uint32_t* scratch, int32_t* idx, int32_t* meta,
                                    int32_t* host_meta_dev, int clear_touched, eslam_stream_t stream) {
    if (!touched || !scratch || !idx || !meta || n_blocks <= 0 || n_blocks >= ((int64_t)1 << 31) || ((uintptr_t)touched & 15)) {
        eslam_set_error("eslam_blocks_compact: null / unaligned argument or n_blocks outside (0, 2^31)");
        return 1;
    }
    const int nchunks = (int)((n_blocks + CHUNK_BYTES - 1) / CHUNK_BYTES);
    hipLaunchKernelGGL(blocks_count_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, touched, n_blocks, scratch);
    hipLaunchKernelGGL(blocks_compact_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, touched, n_blocks, scratch, nchunks,
                       idx, meta, host_meta_dev, clear_touched ? 1 : 0);
    return eslam_check_launch("blocks_compact_kernel");
}

// Two int32 of pinned host memory the device can write (hipHostMalloc, mapped + coherent): *host_ptr for the host to poll,
// *dev_ptr for eslam_blocks_compact's host_meta_dev.  Zeroed.
extern "C" int eslam_host_meta_alloc(void** host_ptr, void** dev_ptr) {
    if (!host_ptr || !dev_ptr) {
        eslam_set_error("eslam_host_meta_alloc: null argument");
        return 1;
    }
    void* h = nullptr;
    void* d = nullptr;
    if (hipHostMalloc(&h, 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess || !h) {
        eslam_set_error("eslam_host_meta_alloc: hipHostMalloc failed: %s", hipGetErrorString(hipGetLastError()));
        return 2;
    }
    if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess || !d) {
        (void)hipHostFree(h);
        eslam_set_error("eslam_host_meta_alloc: hipHostGetDevicePointer failed");
        return 2;
    }
    for (int i = 0; i < 16; ++i) ((volatile int32_t*)h)[i] = 0;
    *host_ptr = h;
    *dev_ptr = d;
    return 0;
}
extern "C" int eslam_host_meta_free(void* host_ptr) {
    if (host_ptr && hipHostFree(host_ptr) != hipSuccess) {
        eslam_set_error("eslam_host_meta_free: hipHostFree failed");
        return 2;
    }
    return 0;
}

static int blocks_move_dev(int mode, float* flat, const int32_t* idx, const int32_t* meta, int64_t capacity, float* tail,
                           int64_t n_tail, int64_t tail_pad, float* buf, uint32_t* step_bump, eslam_stream_t stream) {
    if (!flat || !idx || !meta || (n_tail > 0 && !tail) || (mode != 2 && !buf) || n_tail < 0 || tail_pad < n_tail || (tail_pad & 3) ||
        capacity <= 0 || tail_pad >= ((int64_t)1 << 30) || ((uintptr_t)flat & 15) || ((uintptr_t)buf & 15)) {
        eslam_set_error("eslam_blocks_*_dev: null / unaligned argument, or tail_pad not a multiple of 4 floats >= n_tail");
        return 1;
    }
    const int64_t want = (capacity + 31) / 32;
    const dim3 grid((unsigned)(want < 2048 ? want : 2048));
    hipStream_t st = (hipStream_t)stream;
    if (mode == 0) hipLaunchKernelGGL(blocks_move_dev_kernel<0>, grid, dim3(256), 0, st, flat, idx, meta, tail, (int)n_tail, (int)tail_pad, buf, step_bump);
    else if (mode == 1) hipLaunchKernelGGL(blocks_move_dev_kernel<1>, grid, dim3(256), 0, st, flat, idx, meta, tail, (int)n_tail, (int)tail_pad, buf, step_bump);
    else hipLaunchKernelGGL(blocks_move_dev_kernel<2>, grid, dim3(256), 0, st, flat, idx, meta, tail, (int)n_tail, (int)tail_pad, buf, step_bump);
    return eslam_check_launch("blocks_move_dev_kernel");
}

extern "C" int eslam_blocks_pack_dev(const float* flat, const int32_t* idx, const int32_t* meta, int64_t capacity, const float* tail,
                                     int64_t n_tail, int64_t tail_pad, float* buf, uint32_t* step_bump, eslam_stream_t stream) {
    return blocks_move_dev(0, (float*)flat, idx, meta, capacity, (float*)tail, n_tail, tail_pad, buf, step_bump, stream);
}
extern "C" int eslam_blocks_unpack_dev(float* flat, const int32_t* idx, const int32_t* meta, int64_t capacity, float* tail,
                                       int64_t n_tail, int64_t tail_pad, const float* buf, eslam_stream_t stream) {
    return blocks_move_dev(1, flat, idx, meta, capacity, tail, n_tail, tail_pad, (float*)buf, nullptr, stream);
}
extern "C" int eslam_blocks_zero_dev(float* flat, const int32_t* idx, const int32_t* meta, int64_t capacity, float* tail,
                                     int64_t n_tail, eslam_stream_t stream) {
    return blocks_move_dev(2, flat, idx, meta, capacity, tail, n_tail, (n_tail + 3) & ~(int64_t)3, nullptr, nullptr, stream);
}
