// Mixed-precision forward (BASELINE.json configs[4]: "fp16 planes + bf16 MFMA decoders", a tolerance study, inference
// only): the same fused gather -> MLPs -> composite as render_fwd_kernel with
//   * planes stored as IEEE half, channels-last: a texel is 64 B, a lane's 8 channels are ONE 16-byte load per corner
//     (4 loads per plane instead of 8, half the bytes); the bilinear sum is accumulated in float32;
//   * decoders on bf16 MFMA: the 8 gathered channels of a level, rounded to bf16, are exactly lane (col = point,
//     k-slot = octet)'s B fragment of v_mfma_f32_16x16x32_bf16, so layer 1 is 2 MFMAs (one per level) instead of 16;
//     layers 2 and 3 use v_mfma_f32_16x16x16_bf16 whose k = 4*(lane>>4) + j order is the accumulator's row order
//     (rows 4q + reg), so again each layer's accumulator feeds the next layer without a shuffle; accumulation, biases,
//     activations, alpha and the transmittance scan stay float32.
// 4 MFMAs of 16 cycles per 16 points and decoder instead of 24 of 32 cycles.
#include "eslam_common.h"

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef short short8_t __attribute__((ext_vector_type(8)));
typedef short short4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ short f2bf(float x) {          // round-to-nearest-even float32 -> bfloat16 bits
    unsigned u = __builtin_bit_cast(unsigned, x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (short)(u >> 16);
}

// LDS image of one decoder in bf16 (shorts): W1 [16][64] @0, W2 [16][16] @1024, W3pad [4][16] @1280; float biases kept
// in a float tail: b1 [16], b2 [16], b3pad [4]
#define LP_W1 0
#define LP_W2 1024
#define LP_W3 1280
#define LP_SHORTS 1344
#define LP_BIAS_FLOATS 36

struct LowpLds {
    short w[2][LP_SHORTS];
    float b[2][LP_BIAS_FLOATS];
};

__device__ __forceinline__ void stage_lowp(LowpLds& L, const eslam_decoders_t& dec, int tid) {
    for (int d = 0; d < 2; ++d) {
        const float* w1 = d ? dec.cw1 : dec.w1;
        const float* b1 = d ? dec.cb1 : dec.b1;
        const float* w2 = d ? dec.cw2 : dec.w2;
        const float* b2 = d ? dec.cb2 : dec.b2;
        const float* w3 = d ? dec.cw3 : dec.w3;
        const float* b3 = d ? dec.cb3 : dec.b3;
        const int nout = d ? 3 : 1;
        for (int i = tid; i < 1024; i += 256) L.w[d][LP_W1 + i] = f2bf(w1[i]);
        for (int i = tid; i < 256; i += 256) L.w[d][LP_W2 + i] = f2bf(w2[i]);
        for (int i = tid; i < 64; i += 256) L.w[d][LP_W3 + i] = (i < nout * 16) ? f2bf(w3[i]) : (short)0;
        for (int i = tid; i < 16; i += 256) { L.b[d][i] = b1[i]; L.b[d][16 + i] = b2[i]; }
        for (int i = tid; i < 4; i += 256) L.b[d][32 + i] = (i < nout) ? b3[i] : 0.0f;
    }
}

// 8 half channels of one corner -> float32 weighted accumulate
__device__ __forceinline__ void acc_half8(const half8_t v, float w, float acc[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] += (float)v[i] * w;
}

__device__ __forceinline__ void gather8_half(const eslam_plane_t& P, float u, float v, int q, float acc[8]) {
    const AxisCoord ax = axis_coord(u, P.w);
    const AxisCoord ay = axis_coord(v, P.h);
    const unsigned sy = (unsigned)P.stride_y, sx = (unsigned)P.stride_x;
    const unsigned r0 = ay.i0 * sy, r1 = ay.i1 * sy, c0 = ax.i0 * sx, c1 = ax.i1 * sx, q8 = 8u * q;
    const _Float16* __restrict__ data = (const _Float16*)P.data;
    const half8_t t00 = *(const half8_t*)(data + r0 + c0 + q8);
    const half8_t t01 = *(const half8_t*)(data + r0 + c1 + q8);
    const half8_t t10 = *(const half8_t*)(data + r1 + c0 + q8);
    const half8_t t11 = *(const half8_t*)(data + r1 + c1 + q8);
    acc_half8(t00, (1.0f - ax.t) * (1.0f - ay.t), acc);
    acc_half8(t01, ax.t * (1.0f - ay.t), acc);
    acc_half8(t10, (1.0f - ax.t) * ay.t, acc);
    acc_half8(t11, ax.t * ay.t, acc);
}

__global__ __launch_bounds__(256, 4) void render_fwd_lowp_kernel(const PlaneSet planes, const eslam_decoders_t dec,
                                                                  const Bound bnd, const float* __restrict__ rays_o,
                                                                  const float* __restrict__ rays_d,
                                                                  const float* __restrict__ z_vals, int R, int S,
                                                                  float* __restrict__ depth_out, float* __restrict__ rgb_out,
                                                                  float* __restrict__ sdf_out) {
    __shared__ __attribute__((aligned(16))) LowpLds L;
    stage_lowp(L, dec, threadIdx.x);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int ray = blockIdx.x * 4 + wave;
    if (ray >= R) return;
    const float ox = rays_o[ray * 3 + 0], oy = rays_o[ray * 3 + 1], oz = rays_o[ray * 3 + 2];
    const float dx = rays_d[ray * 3 + 0], dy = rays_d[ray * 3 + 1], dz = rays_d[ray * 3 + 2];
    const float beta = dec.beta[0];
    const float* zrow = z_vals + (int64_t)ray * S;
    float trans_in = 1.0f, acc_depth = 0.f, acc_r = 0.f, acc_g = 0.f, acc_b = 0.f;

    for (int c0 = 0; c0 < S; c0 += WAVE) {
        const int nvalid = min(WAVE, S - c0);
        const int nblk = (nvalid + 15) >> 4;
        float4_t out[2];
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            out[d] = *(const float4_t*)(&L.b[d][32]);
#pragma unroll 1
            for (int b = 0; b < nblk; ++b) {
                const int oz0 = opaque_zero(b);
                const float zb = zrow[min(c0 + 16 * b + r, S - 1)];
                const float px = norm_coord(ox + dx * zb, bnd.lo[0], bnd.hi[0]);
                const float py = norm_coord(oy + dy * zb, bnd.lo[1], bnd.hi[1]);
                const float pz = norm_coord(oz + dz * zb, bnd.lo[2], bnd.hi[2]);
                float feat[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) feat[i] = 0.0f;
#pragma unroll
                for (int lvl = 0; lvl < 2; ++lvl) {
#pragma unroll
                    for (int o = 0; o < 3; ++o) {
                        const eslam_plane_t& P = planes.p[2 * (3 * d + o) + lvl + oz0];
                        gather8_half(P, ORIENT_U(o, px, py, pz), ORIENT_V(o, px, py, pz), q, feat + 8 * lvl);
                    }
                }
                // layer 1: two 16x16x32 bf16 MFMAs (one per level)
                float4_t a1 = *(const float4_t*)(&L.b[d][4 * q]);
#pragma unroll
                for (int lvl = 0; lvl < 2; ++lvl) {
                    short8_t bf, wf = *(const short8_t*)(&L.w[d][LP_W1 + r * 64 + lvl * 32 + 8 * q + oz0]);
#pragma unroll
                    for (int i = 0; i < 8; ++i) bf[i] = f2bf(feat[lvl * 8 + i]);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, bf, a1, 0, 0, 0);
                }
                // layer 2: 16x16x16 bf16, k = 4q + j  <->  accumulator rows 4q + reg
                short4_t h1b, w2f = *(const short4_t*)(&L.w[d][LP_W2 + r * 16 + 4 * q]);
#pragma unroll
                for (int i = 0; i < 4; ++i) h1b[i] = f2bf(fmaxf(a1[i], 0.0f));
                float4_t a2 = *(const float4_t*)(&L.b[d][16 + 4 * q]);
                a2 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(w2f, h1b, a2, 0, 0, 0);
                // layer 3: padded output rows of block b placed at rows 4b..4b+3, accumulated over the blocks
                short4_t h2b, w3f = *(const short4_t*)(&L.w[d][LP_W3 + (r & 3) * 16 + 4 * q]);
#pragma unroll
                for (int i = 0; i < 4; ++i) h2b[i] = f2bf(fmaxf(a2[i], 0.0f));
                if ((r >> 2) != b) w3f = (short4_t){0, 0, 0, 0};
                out[d] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(w3f, h2b, out[d], 0, 0, 0);
            }
        }
        const bool valid = lane < nvalid;
        const int s = c0 + lane;
        const float z = valid ? zrow[s] : 0.0f;
        const float sdf = tanhf(out[0][0]);
        const float cr = sigmoidf_(out[1][0]), cg = sigmoidf_(out[1][1]), cb = sigmoidf_(out[1][2]);
        if (valid) sdf_out[(int64_t)ray * S + s] = sdf;
        float alpha = 1.0f - expf(-beta * sigmoidf_(-sdf * beta));
        if (!valid) alpha = 0.0f;
        const float fac = valid ? (1.0f - alpha) + 1e-10f : 1.0f;
        const float pin = wave_incl_prod(fac, lane);
        float pex = __shfl_up(pin, 1, WAVE);
        if (lane == 0) pex = 1.0f;
        const float w = alpha * (trans_in * pex);
        acc_depth += wave_sum(w * z);
        acc_r += wave_sum(w * cr);
        acc_g += wave_sum(w * cg);
        acc_b += wave_sum(w * cb);
        trans_in *= __shfl(pin, 63, WAVE);
    }
    if (lane == 0) {
        depth_out[ray] = acc_depth;
        rgb_out[ray * 3 + 0] = acc_r;
        rgb_out[ray * 3 + 1] = acc_g;
        rgb_out[ray * 3 + 2] = acc_b;
    }
}

int eslam_validate_planes(const eslam_plane_t* planes, int first, int count);

extern "C" int eslam_render_fwd_lowp(const eslam_plane_t* planes_f16, const eslam_decoders_t* dec, const float* bound6_host,
                                     const float* rays_o, const float* rays_d, const float* z_vals, int R, int S,
                                     float* depth, float* rgb, float* sdf, eslam_stream_t stream) {
    if (R <= 0) return 0;
    if (S <= 0 || S > ESLAM_MAX_SAMPLES) {
        eslam_set_error("eslam_render_fwd_lowp: S=%d outside [1,%d]", S, ESLAM_MAX_SAMPLES);
        return 1;
    }
    if (!planes_f16 || !dec || !bound6_host || !rays_o || !rays_d || !z_vals || !depth || !rgb || !sdf) {
        eslam_set_error("eslam_render_fwd_lowp: null argument");
        return 1;
    }
    if (eslam_validate_planes(planes_f16, 0, NPL)) return 1;
    for (int i = 0; i < NPL; ++i) {
        const eslam_plane_t& p = planes_f16[i];
        if (p.stride_c != 1 || p.stride_x != ESLAM_C_DIM || ((uintptr_t)p.data & 15) != 0 || (p.stride_y & 7) != 0) {
            eslam_set_error("eslam_render_fwd_lowp: plane %d must be a channels-last half tensor (64-byte texels)", i);
            return 1;
        }
    }
    PlaneSet ps;
    for (int i = 0; i < NPL; ++i) ps.p[i] = planes_f16[i];
    Bound bnd;
    for (int k = 0; k < 3; ++k) { bnd.lo[k] = bound6_host[2 * k]; bnd.hi[k] = bound6_host[2 * k + 1]; }
    hipStream_t st = (hipStream_t)stream;
    eslam_prof_begin(PROF_RENDER_FWD, st);
    hipLaunchKernelGGL(render_fwd_lowp_kernel, dim3((R + 3) / 4), dim3(256), 0, st, ps, *dec, bnd, rays_o, rays_d, z_vals, R,
                       S, depth, rgb, sdf);
    eslam_prof_end(PROF_RENDER_FWD, st);
    return eslam_check_launch("render_fwd_lowp_kernel");
}
