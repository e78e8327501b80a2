// Shared device helpers for the ESLAM render kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/eslam_hip.h"

#define WAVE 64
#define NPL ESLAM_N_PLANES

typedef float float4_t __attribute__((ext_vector_type(4)));

// Kernel-argument copy of the 12 plane descriptors (kernarg segment -> scalar loads, wave-uniform).
struct PlaneSet {
    eslam_plane_t p[NPL];
};

struct Bound {
    float lo[3];
    float hi[3];
};

// Decoder weights staged in LDS, one copy per workgroup.  Row-major as in the reference tensors.
// Layout (floats):  per decoder d (0 = sdf, 1 = rgb) at d*DEC_LDS:
//   W1 [16][64] @0, b1 [16] @1024, W2 [16][16] @1040, b2 [16] @1296, W3 [4][16] @1312 (rows >= n_out zero),
//   b3 [4] @1376 (entries >= n_out zero)
#define DEC_W1 0
#define DEC_B1 1024
#define DEC_W2 1040
#define DEC_B2 1296
#define DEC_W3 1312
#define DEC_B3 1376
#define DEC_LDS 1380

// One decoder's 1364 parameters into its LDS image, by 256 threads: all of a thread's loads are issued before its first LDS
// store (4 + 1 + 1 independent loads).  The first version copied array after array - twelve load -> store loops, each
// exposing a full L2 round trip: 12 k of a forward wave's 88 k cycles (in-kernel stamps), in every workgroup of every kernel
// that decodes.
struct DecStage { float a[4], b, c; int dst; };
__device__ __forceinline__ DecStage stage_load_256(const eslam_decoders_t& dec, int d, int tid) {
    const float* w1 = d ? dec.cw1 : dec.w1;
    const float* b1 = d ? dec.cb1 : dec.b1;
    const float* w2 = d ? dec.cw2 : dec.w2;
    const float* b2 = d ? dec.cb2 : dec.b2;
    const float* w3 = d ? dec.cw3 : dec.w3;
    const float* b3 = d ? dec.cb3 : dec.b3;
    const int nout = d ? 3 : 1;
    DecStage s;
#pragma unroll
    for (int k = 0; k < 4; ++k) s.a[k] = w1[tid + 256 * k];
    s.b = w2[tid];
    // the small arrays, one element per thread: [0,16) b1, [16,32) b2, [32,96) W3 padded to 4 rows, [96,100) b3 padded
    s.c = 0.0f;
    s.dst = -1;
    if (tid < 16) { s.c = b1[tid]; s.dst = DEC_B1 + tid; }
    else if (tid < 32) { s.c = b2[tid - 16]; s.dst = DEC_B2 + tid - 16; }
    else if (tid < 96) { if (tid - 32 < nout * 16) s.c = w3[tid - 32]; s.dst = DEC_W3 + tid - 32; }
    else if (tid < 100) { if (tid - 96 < nout) s.c = b3[tid - 96]; s.dst = DEC_B3 + tid - 96; }
    return s;
}
__device__ __forceinline__ void stage_store_256(float* L, const DecStage& s, int tid) {
#pragma unroll
    for (int k = 0; k < 4; ++k) L[DEC_W1 + tid + 256 * k] = s.a[k];
    L[DEC_W2 + tid] = s.b;
    if (s.dst >= 0) L[s.dst] = s.c;
}
__device__ __forceinline__ void stage_decoder_weights_256(float* L, const eslam_decoders_t& dec, int d, int tid) {
    const DecStage s = stage_load_256(dec, d, tid);
    stage_store_256(L, s, tid);
}

// one decoder only (the importance sampler needs the SDF decoder alone), at lds[0 .. DEC_LDS)
__device__ __forceinline__ void stage_decoder_weights_one(float* L, const eslam_decoders_t& dec, int d, int tid, int nthreads) {
    if (nthreads == 256) {
        stage_decoder_weights_256(L, dec, d, tid);
        return;
    }
    const float* w1 = d ? dec.cw1 : dec.w1;
    const float* b1 = d ? dec.cb1 : dec.b1;
    const float* w2 = d ? dec.cw2 : dec.w2;
    const float* b2 = d ? dec.cb2 : dec.b2;
    const float* w3 = d ? dec.cw3 : dec.w3;
    const float* b3 = d ? dec.cb3 : dec.b3;
    const int nout = d ? 3 : 1;
    for (int i = tid; i < 1024; i += nthreads) L[DEC_W1 + i] = w1[i];
    for (int i = tid; i < 256; i += nthreads) L[DEC_W2 + i] = w2[i];
    for (int i = tid; i < 16; i += nthreads) {
        L[DEC_B1 + i] = b1[i];
        L[DEC_B2 + i] = b2[i];
    }
    for (int i = tid; i < 64; i += nthreads) L[DEC_W3 + i] = (i < nout * 16) ? w3[i] : 0.0f;
    for (int i = tid; i < 4; i += nthreads) L[DEC_B3 + i] = (i < nout) ? b3[i] : 0.0f;
}

// both decoders: image of decoder d at lds + d * DEC_LDS
__device__ __forceinline__ void stage_decoder_weights(float* lds, const eslam_decoders_t& dec, int tid, int nthreads) {
    if (nthreads == 256) {       // both decoders' loads in flight together
        const DecStage s0 = stage_load_256(dec, 0, tid), s1 = stage_load_256(dec, 1, tid);
        stage_store_256(lds, s0, tid);
        stage_store_256(lds + DEC_LDS, s1, tid);
        return;
    }
    stage_decoder_weights_one(lds, dec, 0, tid, nthreads);
    stage_decoder_weights_one(lds + DEC_LDS, dec, 1, tid, nthreads);
}

// normalize_3d_coordinate (reference src/common.py:215-217), same operation order.
__device__ __forceinline__ float norm_coord(float p, float lo, float hi) {
    return __fsub_rn(__fmul_rn(__fdiv_rn(__fsub_rn(p, lo), __fsub_rn(hi, lo)), 2.0f), 1.0f);
}

// grid_sample(align_corners=True, padding_mode='border') coordinate rule for one axis of size n:
// ix = clamp(((u+1)/2)*(n-1), 0, n-1);  i0 = floor(ix); i1 = min(i0+1, n-1); t = ix - i0.
// `inside` is the gradient gate of ATen's clip_coordinates_set_grad (strictly inside only).
struct AxisCoord {
    int i0, i1;
    float t;
    bool inside;
};

__device__ __forceinline__ AxisCoord axis_coord(float u, int n) {
    const float nm1 = (float)(n - 1);
    float x = ((u + 1.0f) * 0.5f) * nm1;
    AxisCoord a;
    a.inside = (x > 0.0f) && (x < nm1);
    x = fminf(fmaxf(x, 0.0f), nm1);
    const float f = floorf(x);
    a.i0 = (int)f;
    a.i1 = min(a.i0 + 1, n - 1);
    a.t = x - f;
    return a;
}

// Which two point coordinates index plane orientation o (0 = xy, 1 = xz, 2 = yz): first -> width, second -> height
// (reference src/networks/decoders.py:79-81).
#define ORIENT_U(o, x, y, z) ((o) == 2 ? (y) : (x))
#define ORIENT_V(o, x, y, z) ((o) == 0 ? (y) : (z))

// Ordering point for LDS traffic between lanes of ONE wave (no other wave touches the region): DS operations of a
// wave are executed in issue order, so this only has to stop the compiler from moving accesses across it.
#define WAVE_SYNC()                                              \
    do {                                                         \
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   \
        __builtin_amdgcn_wave_barrier();                         \
    } while (0)

// A wave-uniform zero the optimiser cannot fold (used to pin scalar loads inside loops, see gather_features).
__device__ __forceinline__ int opaque_zero(int loop_var) {
    int z = __builtin_amdgcn_readfirstlane(loop_var);
    // s_and_b32 writes SCC: without the clobber the compiler may keep a comparison result live in SCC across this statement
    // (it did, once: a `last ? 0 : b + 1` next to it selected the wrong arm in one of two unrolled copies of a loop)
    asm volatile("s_and_b32 %0, %0, 0" : "+s"(z) : : "scc");
    return z;
}

// Wave-wide sums and prefix products run on DPP (data-parallel primitives: the operand of a VALU instruction comes from
// another lane of the same 16-lane row, or - row_bcast - from the last lane of the previous row / half), ~10 cycles a step,
// instead of __shfl_* (ds_bpermute_b32 through the LDS crossbar, ~60-100 cycles a step when each step waits for the last):
// a ray's epilogue in the forward kernel alone is 15 such reductions.  The six steps are LLVM's own lowering of a wave64
// scan on gfx9 (row_shr 1, 2, 4, 8, then row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3).
// All 64 lanes must be active (as for __shfl_*).  Profiling switch: -DESLAM_NO_DPP restores the __shfl_* forms.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_from(float ident, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, ident), __builtin_bit_cast(int, v), CTRL,
                                                                 ROW_MASK, 0xf, false));
}

// inclusive prefix sum over the 64 lanes
__device__ __forceinline__ float wave_incl_sum(float v) {
    v += dpp_from<0x111, 0xf>(0.0f, v);
    v += dpp_from<0x112, 0xf>(0.0f, v);
    v += dpp_from<0x114, 0xf>(0.0f, v);
    v += dpp_from<0x118, 0xf>(0.0f, v);
    v += dpp_from<0x142, 0xa>(0.0f, v);
    v += dpp_from<0x143, 0xc>(0.0f, v);
    return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#ifdef ESLAM_NO_DPP
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, WAVE);
    return v;
#else
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wave_incl_sum(v)), WAVE - 1));
#endif
}

// inclusive product scan over the 64 lanes of a wave
__device__ __forceinline__ float wave_incl_prod(float v, int lane) {
#ifdef ESLAM_NO_DPP
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        float o = __shfl_up(v, d, WAVE);
        if (lane >= d) v *= o;
    }
    return v;
#else
    (void)lane;
    v *= dpp_from<0x111, 0xf>(1.0f, v);
    v *= dpp_from<0x112, 0xf>(1.0f, v);
    v *= dpp_from<0x114, 0xf>(1.0f, v);
    v *= dpp_from<0x118, 0xf>(1.0f, v);
    v *= dpp_from<0x142, 0xa>(1.0f, v);
    v *= dpp_from<0x143, 0xc>(1.0f, v);
    return v;
#endif
}

// inclusive suffix sum (lane i gets sum over lanes >= i): row_shl steps inside the 16-lane rows, then the totals of the later
// rows (lane 0 of a row holds its row's total) added through the scalar unit - DPP has no broadcast in this direction
__device__ __forceinline__ float wave_incl_suffix_sum(float v, int lane) {
#ifdef ESLAM_NO_DPP
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        float o = __shfl_down(v, d, WAVE);
        if (lane + d < WAVE) v += o;
    }
    return v;
#else
    v += dpp_from<0x101, 0xf>(0.0f, v);
    v += dpp_from<0x102, 0xf>(0.0f, v);
    v += dpp_from<0x104, 0xf>(0.0f, v);
    v += dpp_from<0x108, 0xf>(0.0f, v);
    const float t1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
    const float t2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
    const float t3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
    const int row = lane >> 4;
    return v + (row == 0 ? t1 + (t2 + t3) : row == 1 ? t2 + t3 : row == 2 ? t3 : 0.0f);
#endif
}

// lane i <- lane i-1 (lane 0 <- first) / lane i <- lane i+1 (lane 63 <- last); one value of a fixed lane in every lane
__device__ __forceinline__ float wave_up1(float v, float first) {
#ifdef ESLAM_NO_DPP
    const float o = __shfl_up(v, 1, WAVE);
    return (threadIdx.x & (WAVE - 1)) == 0 ? first : o;
#else
    return dpp_from<0x138, 0xf>(first, v);          // wave_shr:1
#endif
}
__device__ __forceinline__ float wave_down1(float v, float last) {
#ifdef ESLAM_NO_DPP
    const float o = __shfl_down(v, 1, WAVE);
    return (threadIdx.x & (WAVE - 1)) == WAVE - 1 ? last : o;
#else
    return dpp_from<0x130, 0xf>(last, v);           // wave_shl:1
#endif
}
template <int LANE>
__device__ __forceinline__ float wave_lane(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), LANE));
}

// integer forms for the scatter's sort phases: min / max over the wave (every lane gets the result), inclusive prefix sum
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_from_i(int ident, int v) {
    return __builtin_amdgcn_update_dpp(ident, v, CTRL, ROW_MASK, 0xf, false);
}
#define ESLAM_DPP_REDUCE_I(v, OP, IDENT)                 \
    v = OP(v, dpp_from_i<0x111, 0xf>(IDENT, v));         \
    v = OP(v, dpp_from_i<0x112, 0xf>(IDENT, v));         \
    v = OP(v, dpp_from_i<0x114, 0xf>(IDENT, v));         \
    v = OP(v, dpp_from_i<0x118, 0xf>(IDENT, v));         \
    v = OP(v, dpp_from_i<0x142, 0xa>(IDENT, v));         \
    v = OP(v, dpp_from_i<0x143, 0xc>(IDENT, v));
__device__ __forceinline__ int wave_min_i(int v) {
    ESLAM_DPP_REDUCE_I(v, min, 0x7FFFFFFF)
    return __builtin_amdgcn_readlane(v, WAVE - 1);
}
__device__ __forceinline__ int wave_max_i(int v) {
    ESLAM_DPP_REDUCE_I(v, max, (int)0x80000000)
    return __builtin_amdgcn_readlane(v, WAVE - 1);
}
__device__ __forceinline__ int iadd_(int a, int b) { return a + b; }
__device__ __forceinline__ unsigned wave_incl_sum_u(unsigned x) {
    int v = (int)x;
    ESLAM_DPP_REDUCE_I(v, iadd_, 0)
    return (unsigned)v;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ float4_t mfma16(float a, float b, float4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// per-kernel HIP-event timing (eslam_api.hip); PROF_* ids index eslam_profile_name()
enum { PROF_RENDER_FWD = 0, PROF_COMPOSITE_BWD, PROF_MLP_BWD, PROF_DEC_REDUCE, PROF_SCATTER, PROF_COORD_BWD, PROF_LOSS,
       PROF_SAMPLE_Z, PROF_IMPORTANCE_Z, PROF_DECODE_FWD, PROF_ADAM, PROF_KF_OVERLAP };
void eslam_prof_begin(int id, hipStream_t st);
void eslam_prof_end(int id, hipStream_t st);

// ESLAM_DETERMINISTIC=1 (read once per process): fixed-order loss reduction and fixed-point plane-gradient scatter
extern "C" int eslam_deterministic(void);

// thread-local error string shared by the API translation units
void eslam_set_error(const char* fmt, ...);
int eslam_check_launch(const char* what);
