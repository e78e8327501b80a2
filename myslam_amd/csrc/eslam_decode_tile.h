// Tri-plane gather + fused decoder MLPs for one wave-tile of 64 points (forward direction).
//
// Lane roles inside a wave (lane l = 0..63); the tile is cut into 4 blocks of 16 points:
//   "sample role":  lane l owns point l of the tile (per-point scalars: z, sdf, alpha, ...).
//   "gather role":  for block b lane l works on point 16b + (l >> 2) and on piece g = l & 3 of every 32-channel texel:
//                   channels 4g..4g+3 and 16+4g..16+4g+3 (two 16-byte loads per bilinear corner).
//   "MFMA role":    lane l = 16 q + r holds column r (= point 16b + r) and k-slot q of the B operand.
//
// Gather role: the four lanes of a point are CONSECUTIVE lanes, so every quad of lanes reads 64 contiguous bytes of
// one 128-B texel line.  The texture addresser serves a 16-B-per-lane load one quad per cycle and looks up the lines a
// quad touches one after the other: with the four lanes of a quad on four different POINTS (the MFMA role's own layout,
// which round 1 gathered in) the same loads ran 1.8x slower - 112 vs 63 us for the bench workload's 12.6 M texel reads
// (tools/ubench_gather.hip, profiles/r02/).  The 16 gathered values of a lane then move to the MFMA role with one
// ds_bpermute_b32 each (a lane rotation l = 4p + g -> 16g + p: no LDS memory, 2 LDS-pipe cycles per register).
// Planes in the reference's NCHW layout have no texel lines: there consecutive lanes stay on consecutive points (one
// channel plane per load) and gather role = MFMA role.
//
// MLP (fp32 MFMA 16x16x4, exact fp32 FMA chains): the layers are evaluated TRANSPOSED,
//   H1^T[16 x 16pts] = W1[16 x 64] . feat^T[64 x 16pts],
// so that (i) the gathered registers are the B operand (B[k][col]: col = l & 15 = point, k-slot = l >> 4 = q) with
// the K index permuted to  k(ks, q) = level*32 + ch(q, i),  ks = level*8 + i,  ch(q, i) = i < 4 ? 4q+i : 12+4q+i,
// and the A operand W1[j = l & 15][k(ks, q)] is read from LDS in that same permutation; and (ii) each layer's
// accumulator (rows 4q+reg, col = point) is directly the next layer's B operand with k(ks, q) = 4q + ks.
// The output layer is zero-padded to 16 rows and block b's copy of it is placed at rows 4b..4b+3, all four blocks
// accumulating into ONE accumulator: afterwards lane l (sample role!) holds the outputs of point l in acc[0..2].
// No LDS round trip and no cross-lane shuffle anywhere between the gather and the per-point epilogue.
#pragma once
#include "eslam_common.h"

// channel i (0..7) of piece g of a 32-channel texel: 4g..4g+3, then 16+4g..16+4g+3
__device__ __forceinline__ int piece_channel(int g, int i) { return i < 4 ? 4 * g + i : 12 + 4 * g + i; }

template <bool CL>
__device__ __forceinline__ void gather8(const eslam_plane_t& P, float u, float v, int q, float acc[8]) {
    const AxisCoord ax = axis_coord(u, P.w);
    const AxisCoord ay = axis_coord(v, P.h);
    const float w00 = (1.0f - ax.t) * (1.0f - ay.t);
    const float w01 = ax.t * (1.0f - ay.t);
    const float w10 = (1.0f - ax.t) * ay.t;
    const float w11 = ax.t * ay.t;
    // 32-bit unsigned element offsets from the wave-uniform plane base: lets the loads use the
    // SGPR-base + VGPR-offset addressing form (one address VGPR per load instead of two)
    const unsigned sy = (unsigned)P.stride_y, sx = (unsigned)P.stride_x;
    const unsigned r0 = ay.i0 * sy, r1 = ay.i1 * sy;
    const unsigned c0 = ax.i0 * sx, c1 = ax.i1 * sx;
    const float* __restrict__ data = P.data;
    if (CL) {
        const unsigned q4 = 4u * q;
        const unsigned o00 = r0 + c0 + q4, o01 = r0 + c1 + q4, o10 = r1 + c0 + q4, o11 = r1 + c1 + q4;
        const float4_t a00 = *(const float4_t*)(data + o00), b00 = *(const float4_t*)(data + o00 + 16u);
        const float4_t a01 = *(const float4_t*)(data + o01), b01 = *(const float4_t*)(data + o01 + 16u);
        const float4_t a10 = *(const float4_t*)(data + o10), b10 = *(const float4_t*)(data + o10 + 16u);
        const float4_t a11 = *(const float4_t*)(data + o11), b11 = *(const float4_t*)(data + o11 + 16u);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[i] += a00[i] * w00 + a01[i] * w01 + a10[i] * w10 + a11[i] * w11;
            acc[4 + i] += b00[i] * w00 + b01[i] * w01 + b10[i] * w10 + b11[i] * w11;
        }
    } else {
        const unsigned sc = (unsigned)P.stride_c;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const unsigned co = (unsigned)piece_channel(q, i) * sc;
            acc[i] += data[co + r0 + c0] * w00 + data[co + r0 + c1] * w01 + data[co + r1 + c0] * w10 +
                      data[co + r1 + c1] * w11;
        }
    }
}

// Per-decoder MFMA operand fragments, read from the LDS weight image (eslam_common.h layout).
struct DecFrag {
    float w1[16];     // W1[r][lvl*32 + piece_channel(q, i)], index lvl*8 + i
    float4_t w2;      // W2[r][4q .. 4q+3]
    float4_t w3;      // W3pad[r & 3][4q .. 4q+3]
    float4_t b1, b2;  // b[4q .. 4q+3]
    float4_t b3;      // b3pad[0..3]
};

__device__ __forceinline__ void load_dec_frag(DecFrag& f, const float* L, int r, int q) {
#pragma unroll
    for (int lvl = 0; lvl < 2; ++lvl) {
        const float4_t a = *(const float4_t*)(L + DEC_W1 + r * 64 + lvl * 32 + 4 * q);
        const float4_t b = *(const float4_t*)(L + DEC_W1 + r * 64 + lvl * 32 + 16 + 4 * q);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f.w1[lvl * 8 + i] = a[i];
            f.w1[lvl * 8 + 4 + i] = b[i];
        }
    }
    f.w2 = *(const float4_t*)(L + DEC_W2 + r * 16 + 4 * q);
    f.w3 = *(const float4_t*)(L + DEC_W3 + (r & 3) * 16 + 4 * q);
    f.b1 = *(const float4_t*)(L + DEC_B1 + 4 * q);
    f.b2 = *(const float4_t*)(L + DEC_B2 + 4 * q);
    f.b3 = *(const float4_t*)(L + DEC_B3);
}

// hidden activations of one 16-point block: h^T rows 4q+reg, col = point r
__device__ __forceinline__ void mlp_hidden(const DecFrag& f, const float feat[16], float4_t& h1, float4_t& h2) {
    float4_t a1 = f.b1;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) a1 = mfma16(f.w1[ks], feat[ks], a1);
#pragma unroll
    for (int i = 0; i < 4; ++i) h1[i] = fmaxf(a1[i], 0.0f);
    float4_t a2 = f.b2;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) a2 = mfma16(f.w2[ks], h1[ks], a2);
#pragma unroll
    for (int i = 0; i < 4; ++i) h2[i] = fmaxf(a2[i], 0.0f);
}

// output layer of block b accumulated into the tile-wide accumulator `out` (sample role on return)
__device__ __forceinline__ void mlp_out_accum(const DecFrag& f, const float4_t& h2, int b, int r, float4_t& out) {
    const bool mine = (r >> 2) == b;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) out = mfma16(mine ? f.w3[ks] : 0.0f, h2[ks], out);
}

// Gather the 64 features of decoder `d` (0 geometry planes, 1 colour planes) for the lane's block-role point.
// `opaque0` is an SGPR holding 0 that the compiler cannot see through (see opaque_zero()): indexing the descriptor
// table with it keeps the per-plane scalar loads and the values derived from them INSIDE the caller's loop.
// Without it LICM hoists ~100 loop-invariant scalars of the 12 planes out of the point-block loop, which
// overflows the SGPR file and ends in VGPR spills to scratch.
//
// The six planes of a decoder are gathered through a software pipeline: the 8 x 16-B loads of the next
// GATHER_PIPELINE planes are issued before the FMAs of plane p (which then wait with vmcnt(8*GATHER_PIPELINE)).  A wave is a chain of 48 dependent
// load -> FMA steps per ray; with one plane in flight each step exposed the full L2 / Infinity-Cache latency.
// sched_barrier keeps the compiler from hoisting further ahead (all 48 loads of a block need > 256 VGPRs).
#ifndef GATHER_PIPELINE
#define GATHER_PIPELINE 1
#endif

struct PlaneTaps {              // the 4 corners x 8 channels of one lane and their bilinear weights
    float4_t a00, b00, a01, b01, a10, b10, a11, b11;
    float w00, w01, w10, w11;
};

__device__ __forceinline__ void issue_taps(const eslam_plane_t& P, float u, float v, int q, PlaneTaps& t) {
    const AxisCoord ax = axis_coord(u, P.w);
    const AxisCoord ay = axis_coord(v, P.h);
    t.w00 = (1.0f - ax.t) * (1.0f - ay.t);
    t.w01 = ax.t * (1.0f - ay.t);
    t.w10 = (1.0f - ax.t) * ay.t;
    t.w11 = ax.t * ay.t;
    const unsigned sy = (unsigned)P.stride_y, sx = (unsigned)P.stride_x;
    const unsigned r0 = ay.i0 * sy, r1 = ay.i1 * sy;
    const unsigned c0 = ax.i0 * sx, c1 = ax.i1 * sx;
    const float* __restrict__ data = P.data;
    const unsigned q4 = 4u * q;
    const unsigned o00 = r0 + c0 + q4, o01 = r0 + c1 + q4, o10 = r1 + c0 + q4, o11 = r1 + c1 + q4;
    t.a00 = *(const float4_t*)(data + o00); t.b00 = *(const float4_t*)(data + o00 + 16u);
    t.a01 = *(const float4_t*)(data + o01); t.b01 = *(const float4_t*)(data + o01 + 16u);
    t.a10 = *(const float4_t*)(data + o10); t.b10 = *(const float4_t*)(data + o10 + 16u);
    t.a11 = *(const float4_t*)(data + o11); t.b11 = *(const float4_t*)(data + o11 + 16u);
}

__device__ __forceinline__ void accumulate_taps(const PlaneTaps& t, float acc[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        acc[i] += t.a00[i] * t.w00 + t.a01[i] * t.w01 + t.a10[i] * t.w10 + t.a11[i] * t.w11;
        acc[4 + i] += t.b00[i] * t.w00 + t.b01[i] * t.w01 + t.b10[i] * t.w10 + t.b11[i] * t.w11;
    }
}

template <bool CL>
__device__ __forceinline__ void gather_features(const PlaneSet& planes, int d, float x, float y, float z, int q,
                                                float feat[16], int opaque0) {
#pragma unroll
    for (int i = 0; i < 16; ++i) feat[i] = 0.0f;
    if (CL && GATHER_PIPELINE) {
        // GATHER_PIPELINE = number of planes whose loads are in flight (ring of tap sets, all indices compile-time)
        constexpr int D = GATHER_PIPELINE + 1;
        PlaneTaps taps[D];
#pragma unroll
        for (int k = 0; k < D - 1 && k < 6; ++k)
            issue_taps(planes.p[2 * (3 * d + (k % 3)) + (k / 3) + opaque0], ORIENT_U(k % 3, x, y, z),
                       ORIENT_V(k % 3, x, y, z), q, taps[k % D]);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int kn = k + D - 1;
            if (kn < 6)
                issue_taps(planes.p[2 * (3 * d + (kn % 3)) + (kn / 3) + opaque0], ORIENT_U(kn % 3, x, y, z),
                           ORIENT_V(kn % 3, x, y, z), q, taps[kn % D]);
            accumulate_taps(taps[k % D], feat + 8 * (k / 3));
            __builtin_amdgcn_sched_barrier(0);
        }
        return;
    }
#pragma unroll
    for (int lvl = 0; lvl < 2; ++lvl) {
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            const eslam_plane_t& P = planes.p[2 * (3 * d + o) + lvl + opaque0];
            gather8<CL>(P, ORIENT_U(o, x, y, z), ORIENT_V(o, x, y, z), q, feat + 8 * lvl);
            // keep the 8 x 16-B loads of the next plane from being hoisted above this plane's FMAs: with all 48
            // loads of a block in flight the kernel needs > 256 VGPRs (1 wave / SIMD); per plane it fits in 128.
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// The same gather for a wave that decodes one 16-point block after the other (channels-last planes): plane 0 of the NEXT
// block is requested before this block's features are stored and its MLP runs.  `carry` holds plane 0 of this block, in
// flight, on entry, and plane 0 of block (dn, xn, yn, zn) on return (the last block of a tile requests the tile's first
// block again, unused: a branch around 8 of 384 loads costs more than they do).  Without it every block started with
// an exposed texel latency, and - a wave's loads and stores retire in order on one vmcnt counter - its first wait also
// waited for the previous block's four feature stores.
__device__ __forceinline__ void issue_plane0(const PlaneSet& planes, int d, float x, float y, float z, int q, int opaque0,
                                             PlaneTaps& t) {
    issue_taps(planes.p[6 * d + opaque0], ORIENT_U(0, x, y, z), ORIENT_V(0, x, y, z), q, t);
}

__device__ __forceinline__ void gather_features_chain(const PlaneSet& planes, int d, float x, float y, float z, int q,
                                                      float feat[16], int opaque0, PlaneTaps& carry, int dn, float xn, float yn,
                                                      float zn) {
#pragma unroll
    for (int i = 0; i < 16; ++i) feat[i] = 0.0f;
    PlaneTaps odd;                         // planes 1, 3, 5; planes 0, 2, 4 use `carry`
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int kn = k + 1;
        if (kn < 6) {
            issue_taps(planes.p[2 * (3 * d + (kn % 3)) + (kn / 3) + opaque0], ORIENT_U(kn % 3, x, y, z),
                       ORIENT_V(kn % 3, x, y, z), q, (kn & 1) ? odd : carry);
        } else {
            issue_plane0(planes, dn, xn, yn, zn, q, opaque0, carry);
        }
        accumulate_taps((k & 1) ? odd : carry, feat + 8 * (k / 3));
        __builtin_amdgcn_sched_barrier(0);
    }
}

// store the lane's 16 features (piece q of both levels) of decoder d for point `pt` into feat_out [N,128]
// NT: streaming (non-temporal) stores for the features a forward pass saves (134 MB at 4096 x 64, read back once, by the
// backward pass): they no longer displace plane texels from the XCD's 4 MB L2 on their way out.  Measured at 4096 x 64 /
// 8192 x 96: forward 92 -> 89 / 234 -> 218 us, and the kernels behind it gain too (decoder backward 80 -> 74 / 215 -> 204,
// scatter 118 -> 113 / 305 -> 292).  Non-temporal LOADS of the features in the backward pass (+2 / +25 us there) and
// non-temporal stores of the feature GRADIENTS, which three scatter workgroups re-read through L2 (+8 us in the scatter),
// were measured and dropped (profiles/r02/n_*).
template <bool NT = false>
__device__ __forceinline__ void store_features(float* feat_out, int64_t pt, int d, int q, const float feat[16]) {
    float* dst = feat_out + pt * 128 + d * 64 + 4 * q;
#pragma unroll
    for (int lvl = 0; lvl < 2; ++lvl) {
        float4_t a, b;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a[i] = feat[lvl * 8 + i];
            b[i] = feat[lvl * 8 + 4 + i];
        }
        if (NT) {
            __builtin_nontemporal_store(a, (float4_t*)(dst + lvl * 32));
            __builtin_nontemporal_store(b, (float4_t*)(dst + lvl * 32 + 16));
        } else {
            *(float4_t*)(dst + lvl * 32) = a;
            *(float4_t*)(dst + lvl * 32 + 16) = b;
        }
    }
}

// The same stores as buffer stores that are ALWAYS issued: `byte_off` = the row's byte offset + d * 256 + q * 16, or an offset
// beyond the descriptor's range for a row that does not exist (the hardware drops the store).  A predicated global store sits
// behind a branch, the compiler cannot count it, and every s_waitcnt for an OLDER load then waits for the stores as well (a
// wave's loads and stores retire in order on one counter): with counted stores the wait names how many may stay in flight.
typedef unsigned uint4_bits __attribute__((ext_vector_type(4)));
#define ESLAM_OOB_OFFSET 0xFFFFFF00u
template <bool NT = false>
__device__ __forceinline__ void store_features_buffer(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off, const float feat[16]) {
#pragma unroll
    for (int j = 0; j < 4; ++j)          // piece j: level j >> 1, half j & 1 - the order of store_features
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4_bits, (float4_t){feat[4 * j], feat[4 * j + 1], feat[4 * j + 2], feat[4 * j + 3]}),
                                               rsrc, (int)byte_off + 64 * j, 0, NT ? 2 : 0);      // aux bit 1 = nt
}

// the same 16 values read back (backward pass), gather role
__device__ __forceinline__ void load_features(const float* feat_in, int64_t pt, int d, int q, float feat[16]) {
    const float* src = feat_in + pt * 128 + d * 64 + 4 * q;
#pragma unroll
    for (int lvl = 0; lvl < 2; ++lvl) {
        const float4_t a = *(const float4_t*)(src + lvl * 32), b = *(const float4_t*)(src + lvl * 32 + 16);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            feat[lvl * 8 + i] = a[i];
            feat[lvl * 8 + 4 + i] = b[i];
        }
    }
}

// Lane roles (see the top of this file).  CL = channels-last planes.
template <bool CL> __device__ __forceinline__ int gather_point(int lane) { return CL ? lane >> 2 : lane & 15; }
template <bool CL> __device__ __forceinline__ int gather_piece(int lane) { return CL ? lane & 3 : lane >> 4; }

// gather role -> MFMA role: lane 16q + r takes the registers of lane 4r + q
template <bool CL, int NREG>
__device__ __forceinline__ void to_mfma_role(float v[NREG], int lane) {
    if (!CL) return;
    const int src = (((lane & 15) << 2) | (lane >> 4)) << 2;
#pragma unroll
    for (int i = 0; i < NREG; ++i) v[i] = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(v[i])));
}

// MFMA role -> gather role: lane 4p + g takes the registers of lane 16g + p
template <bool CL, int NREG>
__device__ __forceinline__ void to_gather_role(float v[NREG], int lane) {
    if (!CL) return;
    const int src = (((lane & 3) << 4) | (lane >> 2)) << 2;
#pragma unroll
    for (int i = 0; i < NREG; ++i) v[i] = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(v[i])));
}

// ---------------------------------------------------------------------------------------------------------
// Mixed precision (BASELINE.json configs[4]: fp16 planes + bf16 MFMA decoders): the same tile with
//   * texels read from the planes' IEEE-half copies (64-byte texels): gather-role lane (point l >> 2, piece g = l & 3) loads
//     channels 8g..8g+7 of a corner as ONE 16-byte load, a quad of lanes one whole texel; float32 accumulation;
//   * decoders on bf16 MFMA with float32 accumulation: layer 1 is one v_mfma_f32_16x16x32_bf16 per level (the 8 gathered
//     channels of MFMA-role lane (r, q), rounded to bf16, ARE its B fragment: k = 8q + j), layers 2 and 3
//     v_mfma_f32_16x16x16_bf16, whose k = 4 (l >> 4) + j order is the accumulator's row order (rows 4q + reg): each layer's
//     accumulator feeds the next without a shuffle, as in the float32 tile.  4 MFMAs of 16 cycles per 16 points and
//     decoder instead of 24 of 32 cycles.
// LDS image per decoder (shorts): W1 [16][64] @0, W2 [16][16] @1024, W3pad [4][16] @1280; float biases behind both
// decoders' shorts: b1 [16], b2 [16], b3pad [4] per decoder.
// ---------------------------------------------------------------------------------------------------------
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef short short8_t __attribute__((ext_vector_type(8)));
typedef short short4_t __attribute__((ext_vector_type(4)));
#define LP_W1 0
#define LP_W2 1024
#define LP_W3 1280
#define LP_SHORTS 1344
#define LP_BIAS_FLOATS 36
#define LP_LDS_FLOATS (LP_SHORTS + 2 * LP_BIAS_FLOATS)      // both decoders: 2 * 1344 shorts = 1344 floats, + biases

__device__ __forceinline__ short f2bf(float x) {          // round-to-nearest-even float32 -> bfloat16 bits
    unsigned u = __builtin_bit_cast(unsigned, x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (short)(u >> 16);
}
__device__ __forceinline__ short4_t pack4(float a, float b, float c, float d) { return (short4_t){f2bf(a), f2bf(b), f2bf(c), f2bf(d)}; }

__device__ __forceinline__ short* lp_weights(float* lds, int d) { return (short*)lds + d * LP_SHORTS; }
__device__ __forceinline__ float* lp_biases(float* lds, int d) { return lds + LP_SHORTS + d * LP_BIAS_FLOATS; }

__device__ __forceinline__ void stage_decoder_weights_lowp(float* lds, const eslam_decoders_t& dec, int tid, int nthreads) {
    if (nthreads == 256) {       // every load of the thread (both decoders) in flight before the first LDS store, as stage_load_256
        const DecStage s0 = stage_load_256(dec, 0, tid), s1 = stage_load_256(dec, 1, tid);
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            const DecStage& s = d ? s1 : s0;
            short* W = lp_weights(lds, d);
            float* B = lp_biases(lds, d);
#pragma unroll
            for (int k = 0; k < 4; ++k) W[LP_W1 + tid + 256 * k] = f2bf(s.a[k]);
            W[LP_W2 + tid] = f2bf(s.b);
            // stage_load_256's small-array slots: [0,16) b1, [16,32) b2, [32,96) W3 padded, [96,100) b3 padded
            if (tid < 16) B[tid] = s.c;
            else if (tid < 32) B[tid] = s.c;                       // B[16 + (tid - 16)]
            else if (tid < 96) W[LP_W3 + tid - 32] = f2bf(s.c);
            else if (tid < 100) B[32 + tid - 96] = s.c;
        }
        return;
    }
    for (int d = 0; d < 2; ++d) {
        const float* w1 = d ? dec.cw1 : dec.w1;
        const float* b1 = d ? dec.cb1 : dec.b1;
        const float* w2 = d ? dec.cw2 : dec.w2;
        const float* b2 = d ? dec.cb2 : dec.b2;
        const float* w3 = d ? dec.cw3 : dec.w3;
        const float* b3 = d ? dec.cb3 : dec.b3;
        const int nout = d ? 3 : 1;
        short* W = lp_weights(lds, d);
        float* B = lp_biases(lds, d);
        for (int i = tid; i < 1024; i += nthreads) W[LP_W1 + i] = f2bf(w1[i]);
        for (int i = tid; i < 256; i += nthreads) W[LP_W2 + i] = f2bf(w2[i]);
        for (int i = tid; i < 64; i += nthreads) W[LP_W3 + i] = (i < nout * 16) ? f2bf(w3[i]) : (short)0;
        for (int i = tid; i < 16; i += nthreads) { B[i] = b1[i]; B[16 + i] = b2[i]; }
        for (int i = tid; i < 4; i += nthreads) B[32 + i] = (i < nout) ? b3[i] : 0.0f;
    }
}

struct DecFragLP {
    short8_t w1[2];   // W1[r][lvl*32 + 8q .. 8q+7]
    short4_t w2, w3;  // W2[r][4q .. 4q+3], W3pad[r & 3][4q .. 4q+3]
    float4_t b1, b2, b3;
};

__device__ __forceinline__ void load_dec_frag_lp(DecFragLP& f, const short* W, const float* B, int r, int q) {
    f.w1[0] = *(const short8_t*)(W + LP_W1 + r * 64 + 8 * q);
    f.w1[1] = *(const short8_t*)(W + LP_W1 + r * 64 + 32 + 8 * q);
    f.w2 = *(const short4_t*)(W + LP_W2 + r * 16 + 4 * q);
    f.w3 = *(const short4_t*)(W + LP_W3 + (r & 3) * 16 + 4 * q);
    f.b1 = *(const float4_t*)(B + 4 * q);
    f.b2 = *(const float4_t*)(B + 16 + 4 * q);
    f.b3 = *(const float4_t*)(B + 32);
}

// feat: MFMA role, feat[lvl*8 + j] = channel 8q + j of the level.  a1 / a2: pre-activations (float32), D layout.
__device__ __forceinline__ void mlp_hidden_lp(const DecFragLP& f, const float feat[16], float4_t& a1, float4_t& a2) {
    a1 = f.b1;
#pragma unroll
    for (int lvl = 0; lvl < 2; ++lvl) {
        short8_t bf;
#pragma unroll
        for (int i = 0; i < 8; ++i) bf[i] = f2bf(feat[lvl * 8 + i]);
        a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.w1[lvl], bf, a1, 0, 0, 0);
    }
    const short4_t h1b = pack4(fmaxf(a1[0], 0.f), fmaxf(a1[1], 0.f), fmaxf(a1[2], 0.f), fmaxf(a1[3], 0.f));
    a2 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(f.w2, h1b, f.b2, 0, 0, 0);
}

__device__ __forceinline__ void mlp_out_accum_lp(const DecFragLP& f, const float4_t& a2, int b, int r, float4_t& out) {
    const short4_t h2b = pack4(fmaxf(a2[0], 0.f), fmaxf(a2[1], 0.f), fmaxf(a2[2], 0.f), fmaxf(a2[3], 0.f));
    const short4_t w3 = ((r >> 2) == b) ? f.w3 : (short4_t){0, 0, 0, 0};
    out = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(w3, h2b, out, 0, 0, 0);
}

__device__ __forceinline__ void gather8_half(const eslam_plane_t& P, float u, float v, int g, float acc[8]) {
    const AxisCoord ax = axis_coord(u, P.w);
    const AxisCoord ay = axis_coord(v, P.h);
    const unsigned sy = (unsigned)P.stride_y, sx = (unsigned)P.stride_x;
    const unsigned r0 = ay.i0 * sy, r1 = ay.i1 * sy, c0 = ax.i0 * sx, c1 = ax.i1 * sx, g8 = 8u * g;
    const _Float16* __restrict__ data = (const _Float16*)P.data_f16;
    const half8_t t00 = *(const half8_t*)(data + r0 + c0 + g8);
    const half8_t t01 = *(const half8_t*)(data + r0 + c1 + g8);
    const half8_t t10 = *(const half8_t*)(data + r1 + c0 + g8);
    const half8_t t11 = *(const half8_t*)(data + r1 + c1 + g8);
    const float w00 = (1.0f - ax.t) * (1.0f - ay.t), w01 = ax.t * (1.0f - ay.t), w10 = (1.0f - ax.t) * ay.t, w11 = ax.t * ay.t;
#pragma unroll
    for (int i = 0; i < 8; ++i)
        acc[i] += (float)t00[i] * w00 + (float)t01[i] * w01 + (float)t10[i] * w10 + (float)t11[i] * w11;
}

// the 64 features of decoder d for the lane's gather-role point: feat[lvl*8 + i] = channel 8g + i of the level.
// (The request chain of gather_features_chain was tried here too: 216 instead of 166 VGPRs, 2 waves per SIMD instead of 3,
// forward 82 -> 90 us at 5000 x 56.  Dropped.)
__device__ __forceinline__ void gather_features_half(const PlaneSet& planes, int d, float x, float y, float z, int g,
                                                     float feat[16], int opaque0) {
#pragma unroll
    for (int i = 0; i < 16; ++i) feat[i] = 0.0f;
#pragma unroll
    for (int lvl = 0; lvl < 2; ++lvl) {
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            gather8_half(planes.p[2 * (3 * d + o) + lvl + opaque0], ORIENT_U(o, x, y, z), ORIENT_V(o, x, y, z), g, feat + 8 * lvl);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// Features of the mixed-precision path are SAVED AS bf16 - the very values the decoders' first layer consumed - in
// feat_out viewed as [N,128] shorts (half the bytes of the float32 path's buffer): natural channel order, gather-role lane
// (point, piece g) owns channels 8g..8g+7 of each level (16 bytes; a quad of lanes 64 contiguous bytes).  The backward
// pass's recompute is then bit-identical to the forward pass, and both of its reads of the features move half the bytes.
__device__ __forceinline__ void store_features_lp(float* feat_out, int64_t pt, int d, int g, const float feat[16]) {
    short* dst = (short*)feat_out + pt * 128 + d * 64 + 8 * g;
#pragma unroll
    for (int lvl = 0; lvl < 2; ++lvl) {
        short8_t v;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = f2bf(feat[lvl * 8 + i]);
        __builtin_nontemporal_store(v, (short8_t*)(dst + lvl * 32));      // streamed, as in store_features<true>
    }
}

// hidden layers from B fragments that are already bf16 (the saved features of the backward pass)
__device__ __forceinline__ void mlp_hidden_lp_bf(const DecFragLP& f, const short8_t bf[2], float4_t& a1, float4_t& a2) {
    a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.w1[0], bf[0], f.b1, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.w1[1], bf[1], a1, 0, 0, 0);
    const short4_t h1b = pack4(fmaxf(a1[0], 0.f), fmaxf(a1[1], 0.f), fmaxf(a1[2], 0.f), fmaxf(a1[3], 0.f));
    a2 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(f.w2, h1b, f.b2, 0, 0, 0);
}
