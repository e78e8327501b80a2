// Plane layout change, all 12 planes in one launch: NCHW-contiguous (what reference src/ESLAM.py:199-210 allocates: one
// texel's 32 channels are h*w floats apart) <-> channels-last (one texel = one 128-byte line, what every gather and scatter
// of this library wants).  The render path calls it per render_batch_ray on planes in the reference's own layout - the
// mapper swaps the plane Parameters every frame (src/Mapper.py:254-266), so nothing is cached across calls - and once more
// on the gradients on the way back: 2 x 27 MB (room0) at the HBM rate instead of a 5x slower gather and a scatter that
// needs one float atomic per ELEMENT (1.52 ms instead of 0.28 ms per iteration at 4096 x 64).
#include "eslam_common.h"

#define RL_TEX 64                 // texels per tile

// FIELD 0: dst.data <- src.data (plane values); FIELD 1: dst.grad <- src.grad (gradients, on the way back)
// TO_CL: src is NCHW-contiguous, dst channels-last; else the other way round.  A tile = RL_TEX texels x 32 channels through
// LDS: the NCHW side moves 256 contiguous bytes per wave and channel, the channels-last side 16 bytes per lane.
template <bool TO_CL>
__global__ __launch_bounds__(256) void planes_relayout_kernel(const PlaneSet src, const PlaneSet dst, const int field) {
    __shared__ float tile[ESLAM_C_DIM][RL_TEX + 1];
    const eslam_plane_t& S = src.p[blockIdx.y];
    const eslam_plane_t& D = dst.p[blockIdx.y];
    const float* __restrict__ s = field ? S.grad : S.data;
    float* __restrict__ d = field ? D.grad : (float*)D.data;
    const int64_t ntex = (int64_t)S.h * S.w;
    const int64_t ntiles = (ntex + RL_TEX - 1) / RL_TEX;
    const int tid = threadIdx.x;
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int64_t t0 = t * RL_TEX;
        const int nt = (int)min((int64_t)RL_TEX, ntex - t0);
        if (TO_CL) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = (tid >> 6) + 4 * i, x = tid & 63;
                if (x < nt) tile[c][x] = s[(int64_t)c * ntex + t0 + x];
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int idx = tid + 256 * j, x = idx >> 3, c4 = (idx & 7) * 4;
                if (x < nt)
                    *(float4_t*)(d + (t0 + x) * ESLAM_C_DIM + c4) = (float4_t){tile[c4][x], tile[c4 + 1][x], tile[c4 + 2][x], tile[c4 + 3][x]};
            }
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int idx = tid + 256 * j, x = idx >> 3, c4 = (idx & 7) * 4;
                if (x < nt) {
                    const float4_t v = *(const float4_t*)(s + (t0 + x) * ESLAM_C_DIM + c4);
                    tile[c4][x] = v[0]; tile[c4 + 1][x] = v[1]; tile[c4 + 2][x] = v[2]; tile[c4 + 3][x] = v[3];
                }
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = (tid >> 6) + 4 * i, x = tid & 63;
                if (x < nt) d[(int64_t)c * ntex + t0 + x] = tile[c][x];
            }
        }
        __syncthreads();
    }
}

static bool dense_cl(const eslam_plane_t& p) {
    return p.stride_c == 1 && p.stride_x == ESLAM_C_DIM && p.stride_y == (int64_t)ESLAM_C_DIM * p.w;
}
static bool dense_nchw(const eslam_plane_t& p) {
    return p.stride_x == 1 && p.stride_y == p.w && p.stride_c == (int64_t)p.h * p.w;
}

extern "C" int eslam_planes_relayout(const eslam_plane_t* src, const eslam_plane_t* dst, int field, eslam_stream_t stream) {
    if (!src || !dst || (field != 0 && field != 1)) {
        eslam_set_error("eslam_planes_relayout: null argument or field not 0 / 1");
        return 1;
    }
    PlaneSet a, b;
    int to_cl = -1;
    int64_t most = 0;
    for (int i = 0; i < NPL; ++i) {
        a.p[i] = src[i];
        b.p[i] = dst[i];
        const void* sp = field ? (const void*)src[i].grad : (const void*)src[i].data;
        const void* dp = field ? (const void*)dst[i].grad : (const void*)dst[i].data;
        if (!sp || !dp) {
            eslam_set_error("eslam_planes_relayout: plane %d has no %s pointer", i, field ? "grad" : "data");
            return 1;
        }
        if (src[i].h != dst[i].h || src[i].w != dst[i].w || src[i].h < 1 || src[i].w < 1) {
            eslam_set_error("eslam_planes_relayout: plane %d: shapes differ (%d x %d vs %d x %d)", i, src[i].h, src[i].w, dst[i].h, dst[i].w);
            return 1;
        }
        int dir;
        if (dense_nchw(src[i]) && dense_cl(dst[i])) dir = 1;
        else if (dense_cl(src[i]) && dense_nchw(dst[i])) dir = 0;
        else {
            eslam_set_error("eslam_planes_relayout: plane %d: one side must be NCHW-contiguous and the other channels-last, both dense", i);
            return 1;
        }
        if (to_cl >= 0 && dir != to_cl) {
            eslam_set_error("eslam_planes_relayout: the 12 planes must all go the same way");
            return 1;
        }
        to_cl = dir;
        const void* clp = dir ? dp : sp;
        if (((uintptr_t)clp & 15) != 0) {
            eslam_set_error("eslam_planes_relayout: the channels-last side of plane %d is not 16-byte aligned", i);
            return 1;
        }
        const int64_t nt = ((int64_t)src[i].h * src[i].w + RL_TEX - 1) / RL_TEX;
        most = nt > most ? nt : most;
    }
    dim3 grid((unsigned)(most < 1024 ? most : 1024), NPL);
    if (to_cl) hipLaunchKernelGGL(planes_relayout_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, a, b, field);
    else hipLaunchKernelGGL(planes_relayout_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, a, b, field);
    return eslam_check_launch("planes_relayout_kernel");
}
