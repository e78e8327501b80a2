// Mixed precision (BASELINE.json configs[4]: "fp16 planes + bf16 MFMA decoders"): the kernels themselves are the LOWP
// instantiations of render_fwd_kernel / mlp_bwd_kernel (eslam_decode_tile.h describes the tile); this file holds what is
// left: refreshing the half copies of the planes from their float32 masters, and the inference-only entry point of round 1.
#include "eslam_common.h"

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));

// all 12 planes in ONE launch (blockIdx.y = plane): float32 master -> half copy, both channels-last (same linear order)
__global__ __launch_bounds__(256) void planes_to_half_kernel(const PlaneSet planes) {
    const eslam_plane_t& P = planes.p[blockIdx.y];
    const int64_t n8 = (int64_t)P.h * P.w * (ESLAM_C_DIM / 8);
    const float4_t* __restrict__ src = (const float4_t*)P.data;
    half8_t* __restrict__ dst = (half8_t*)P.data_f16;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const float4_t a = src[2 * i], b = src[2 * i + 1];
        dst[i] = (half8_t){(_Float16)a[0], (_Float16)a[1], (_Float16)a[2], (_Float16)a[3],
                           (_Float16)b[0], (_Float16)b[1], (_Float16)b[2], (_Float16)b[3]};
    }
}

int eslam_validate_planes(const eslam_plane_t* planes, int first, int count);
int eslam_planes_lowp(const eslam_plane_t* planes);

extern "C" int eslam_planes_to_half(const eslam_plane_t* planes, eslam_stream_t stream) {
    if (!planes) {
        eslam_set_error("eslam_planes_to_half: null argument");
        return 1;
    }
    if (eslam_validate_planes(planes, 0, NPL)) return 1;
    if (eslam_planes_lowp(planes) != 1) {
        if (eslam_planes_lowp(planes) == 0) eslam_set_error("eslam_planes_to_half: the planes carry no half copies (data_f16)");
        return 1;
    }
    PlaneSet ps;
    int64_t most = 0;
    for (int i = 0; i < NPL; ++i) {
        ps.p[i] = planes[i];
        if (((uintptr_t)planes[i].data & 15) != 0) {
            eslam_set_error("eslam_planes_to_half: plane %d is not 16-byte aligned", i);
            return 1;
        }
        const int64_t n8 = (int64_t)planes[i].h * planes[i].w * (ESLAM_C_DIM / 8);
        most = n8 > most ? n8 : most;
    }
    const int64_t want = (most + 255) / 256;
    hipLaunchKernelGGL(planes_to_half_kernel, dim3((unsigned)(want < 1024 ? want : 1024), NPL), dim3(256), 0,
                       (hipStream_t)stream, ps);
    return eslam_check_launch("planes_to_half_kernel");
}


// round 1's inference-only entry: planes_f16[i].data points at the HALF data.  Kept as a thin wrapper over eslam_render_fwd
// with the half pointers moved into data_f16.
extern "C" int eslam_render_fwd_lowp(const eslam_plane_t* planes_f16, const eslam_decoders_t* dec, const float* bound6_host,
                                     const float* rays_o, const float* rays_d, const float* z_vals, int R, int S,
                                     float* depth, float* rgb, float* sdf, eslam_stream_t stream) {
    if (R <= 0) return 0;
    if (!planes_f16) {
        eslam_set_error("eslam_render_fwd_lowp: null argument");
        return 1;
    }
    eslam_plane_t p[NPL];
    for (int i = 0; i < NPL; ++i) {
        p[i] = planes_f16[i];
        p[i].data_f16 = planes_f16[i].data;
        p[i].grad = nullptr;
    }
    return eslam_render_fwd(p, dec, bound6_host, rays_o, rays_d, z_vals, R, S, depth, rgb, sdf, nullptr, nullptr, nullptr, nullptr,
                            stream);
}
