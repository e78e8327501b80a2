// Backward kernels (K8 of SURVEY.md section 2.1): autograd of composite -> decoders -> tri-plane lookup.
//
//   composite_bwd_kernel   per ray : upstream (g_depth, g_rgb, g_sdf) -> pre-activation output grads g_o[N,4], g_beta
//   mlp_bwd_kernel         per 64-point tile and decoder: recompute hidden layers, back-propagate to the 64
//                          features (g_feat[N,128]) and to the decoder parameters (per-wave slabs); fp32 MFMA
//   dec_grad_reduce_kernel slabs -> flat decoder gradient
//   scatter (eslam_scatter.hip)  g_feat -> plane gradients
//   coord_bwd_kernel       optional: gradient w.r.t. the sample position -> rays_o / rays_d (pose) or points
#include <stdlib.h>
#include "eslam_decode_tile.h"
#include "eslam_loss_final.h"
#include "eslam_dec_reduce.h"

// ---------------------------------------------------------------------------------------------------------
// composite backward (autograd of reference src/utils/Renderer.py:140-153), one 64-sample chunk of one ray, sample role
// ---------------------------------------------------------------------------------------------------------
// Upstream of a ray: d L / d depth, d L / d rgb (gd, gr, gg, gb); per sample d L / d sdf (g_sdf_s).  Returns the
// pre-activation output gradients of the sample (o[0..2] colour, o[3] sdf) and adds to gbeta_acc.  `carry` is the suffix
// sum of gw*w over the LATER chunks (chunks are visited last to first), `trans_in` the transmittance entering the chunk.
struct RayUp { float gd, gr, gg, gb; };

__device__ __forceinline__ float4_t composite_bwd_chunk(const RayUp up, float beta, bool valid, float sd, float z, float cr,
                                                        float cg, float cb, float g_sdf_s, float trans_in, float& carry,
                                                        float& gbeta_acc, int lane) {
    const float sg = sigmoidf_(-sd * beta);
    const float e = expf(-beta * sg);
    const float alpha = valid ? 1.0f - e : 0.0f;
    const float fac = valid ? (1.0f - alpha) + 1e-10f : 1.0f;
    const float pin = wave_incl_prod(fac, lane);
    const float pex = wave_up1(pin, 1.0f);
    const float T = trans_in * pex;
    const float w = alpha * T;
    const float gw = up.gd * z + up.gr * cr + up.gg * cg + up.gb * cb;
    const float v = valid ? gw * w : 0.0f;
    // exclusive suffix sum_{k>i} gw_k w_k, formed WITHOUT subtracting v_i from an inclusive sum: w decays
    // geometrically along the ray, so (inclusive - own) would lose the small tail in the rounding of the
    // dominant own term, and g_alpha is itself a cancelling difference of two terms of the size of gw.
    const float vn = wave_down1(v, 0.0f);
    const float after_local = wave_incl_suffix_sum(vn, lane);
    const float after = after_local + carry;
    const float g_alpha = gw * T - after / fac;
    carry += wave_lane<0>(after_local) + wave_lane<0>(v);
    float4_t o = (float4_t){0.f, 0.f, 0.f, 0.f};
    if (valid) {
        const float ds = sg * (1.0f - sg);                // sigmoid'
        const float dalpha_dsdf = -beta * beta * e * ds;
        const float dalpha_dbeta = e * (sg - beta * ds * sd);
        gbeta_acc += g_alpha * dalpha_dbeta;
        const float g_sdf_tot = g_sdf_s + g_alpha * dalpha_dsdf;
        o[0] = w * up.gr * cr * (1.0f - cr);
        o[1] = w * up.gg * cg * (1.0f - cg);
        o[2] = w * up.gb * cb * (1.0f - cb);
        o[3] = g_sdf_tot * (1.0f - sd * sd);
    }
    return o;
}

// decode-mode variant: g_o = g_raw * activation'(raw)      (autograd of decoders.py:103,123)
__global__ void decode_act_bwd_kernel(const float* __restrict__ raw, const float* __restrict__ g_raw, int64_t N,
                                      float* __restrict__ g_o) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const float4_t v = *(const float4_t*)(raw + 4 * i);
    const float4_t g = *(const float4_t*)(g_raw + 4 * i);
    float4_t o;
    o[0] = g[0] * v[0] * (1.0f - v[0]);
    o[1] = g[1] * v[1] * (1.0f - v[1]);
    o[2] = g[2] * v[2] * (1.0f - v[2]);
    o[3] = g[3] * (1.0f - v[3] * v[3]);
    *(float4_t*)(g_o + 4 * i) = o;
}

// ---------------------------------------------------------------------------------------------------------
// decoder MLP backward (autograd of reference src/networks/decoders.py:97-103 / 117-123), with the composite backward
// and - when the loss is the fused mapping loss - the loss gradient formed in the same wave
// ---------------------------------------------------------------------------------------------------------
#define TP 20                      // row pitch (floats) of the 16x16 transpose tiles: conflict-free, 16-B aligned
#define TPF 68                     // row pitch of the 16-point x 64-feature tile (floats)
#define TPH 72                     // the same tile in bf16 (shorts): 144-byte rows keep the 16-byte writes aligned
#ifndef BWD_STAMPS
#define BWD_STAMPS 0             // profiling only: per-phase cycle counts (s_memtime) of one wave per decoder, printed at the end
#endif
#ifndef BWD_FBK_LDS
#define BWD_FBK_LDS 1              // A/B switch: 0 = re-read the block's feature rows from global memory for the g_W1 contraction
#endif

struct RayBwdIn {                  // MODE >= 1: what the composite backward of a ray reads
    const float* z_vals;           // [R,S]
    const float* sdf;              // [R,S]   tanh outputs of the forward pass
    const float* raw_rgb;          // [R,S,3] sigmoid outputs
    const float* beta;             // [1]
    const float* g_depth;          // [R]   upstream, any of the three may be NULL (= 0); with MODE 2 they are ADDED to the
    const float* g_rgb;            // [R,3] loss's own gradients
    const float* g_sdf;            // [R,S]
    float* beta_parts;             // [gridDim.x] partial sums of g_beta (written by the sdf decoder's workgroups)
    int R, S;
};

// MODE 0: tiles of 64 free points, pre-activation output gradients g_o [N,4] from memory (decode_act_bwd_kernel).
// MODE 1: a wave owns a RAY: it runs the composite backward of its chunks (upstream gradients from memory) and feeds the
//         result to the MLP backward from registers - the sample role of the one IS the B operand of the other.  Both
//         decoders' workgroups (blockIdx.y) repeat the scalar composite work; neither writes g_o.
// MODE 2: as 1, and the upstream gradients are those of the mapping loss (src/Mapper.py:110-144,337-346), formed here from
//         the global set sizes in li.acc: loss gradient, composite backward and decoder backward are ONE launch.
// WGRAD = false: decoders are frozen (tracking, reference src/Tracker.py:111-112): only g_feat is produced, the
// parameter-gradient contractions (28 of the 68 MFMAs per block), their LDS transposes and the slabs are skipped.
// LOWP: the mixed-precision tile (eslam_decode_tile.h): hidden layers recomputed and every product of the backward pass on
// bf16 MFMA (16x16x32 / 16x16x16) with float32 accumulation - 15 MFMAs of 16 cycles per 16 points and decoder instead of
// 68 of 32 cycles; features, activations' masks, biases and all accumulators stay float32.
#ifndef BWD_BUFSTORE
#define BWD_BUFSTORE 1    // A/B switch: 0 = the feature-gradient rows leave through predicated global stores (round 2)
#endif
#ifndef BWD_TILE_AHEAD
#define BWD_TILE_AHEAD 1  // A/B switch: 0 = a tile requests its own first block of rows (and waits for them) at its head
#endif
#ifndef BWD_ABLATE
#define BWD_ABLATE 0      // profiling only (make variant VFLAGS=-DBWD_ABLATE=n): 1 no g_feat stores, 2 no weight gradients, 4 no feature loads
#endif
template <int MODE, bool WGRAD, bool LOWP>
__global__ __launch_bounds__(256, 2) void mlp_bwd_kernel(const eslam_decoders_t dec, const float* __restrict__ feat,
                                                      const float* __restrict__ g_o, int64_t N,
                                                      float* __restrict__ g_feat, float* __restrict__ slabs,
                                                      const RayBwdIn rb, const LossGradIn li) {
    __shared__ __attribute__((aligned(16))) float wlds[2 * DEC_LDS];
    __shared__ __attribute__((aligned(16))) float tiles[4][4][16 * TP];   // per wave: gz1, gz2, h1, h2 (as [pt][j])
    __shared__ __attribute__((aligned(16))) float gtile[4][64 * 4];       // per wave: g_o of the tile [pt][o]
    // per wave: the block's features [pt][feature] for the g_W1 contraction (BWD_FBK_LDS): float32, or bf16 at half the pitch
    __shared__ __attribute__((aligned(16))) float ftile[(WGRAD && BWD_FBK_LDS) ? 4 : 1][(WGRAD && BWD_FBK_LDS) ? (LOWP ? 16 * TPH / 2 : 16 * TPF) : 4];
    if (LOWP) stage_decoder_weights_lowp(wlds, dec, threadIdx.x, blockDim.x);
    else stage_decoder_weights(wlds, dec, threadIdx.x, blockDim.x);
    __syncthreads();

    const int d = blockIdx.y;                         // 0 = sdf decoder, 1 = colour decoder
    const int nout = d ? 3 : 1;
    const float* L = wlds + d * DEC_LDS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, q = lane >> 4;           // MFMA role
    const int gp = lane >> 2, gq = lane & 3;          // gather role: feat / g_feat rows are read and written a quad per row
    float* tz1 = tiles[wave][0];
    float* tz2 = tiles[wave][1];
    float* th1 = tiles[wave][2];
    float* th2 = tiles[wave][3];
    float* gt = gtile[wave];

    DecFrag f;
    float w3col[4], w2t[4], w1t[4][4];
    DecFragLP fl;
    short4_t w3colp = {0, 0, 0, 0}, w2tp = {0, 0, 0, 0}, w1tp[4];
    if (LOWP) {
        const short* W = lp_weights(wlds, d);
        load_dec_frag_lp(fl, W, lp_biases(wlds, d), r, q);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            if (q == 0) w3colp[jj] = W[LP_W3 + jj * 16 + r];                 // A[i = j = r][k = o = jj]: W3pad[o][j]
            w2tp[jj] = W[LP_W2 + (4 * q + jj) * 16 + r];                     // A[i = k' = r][k = j = 4q+jj]: W2[j][k']
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) w1tp[mb][jj] = W[LP_W1 + (4 * q + jj) * 64 + 16 * mb + r];   // W1[j][f = 16 mb + r]
        }
    } else {
    load_dec_frag(f, L, r, q);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        w3col[ks] = L[DEC_W3 + ks * 16 + r];                    // W3pad[o = ks][j = r]
        w2t[ks] = L[DEC_W2 + (4 * q + ks) * 16 + r];            // W2[j = 4q+ks][k' = r]
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)                          // row r of row block mb <-> feature (mb>>1)*32 + 16*(mb&1) + r:
            w1t[mb][ks] = L[DEC_W1 + (4 * q + ks) * 64 + (mb >> 1) * 32 + 16 * (mb & 1) + r];   // lane q then holds piece q
    }
    }

    float4_t gW1[4], gW2, gW3, gb1, gb2;
#pragma unroll
    for (int i = 0; i < 4; ++i) gW1[i] = (float4_t){0.f, 0.f, 0.f, 0.f};
    gW2 = gW3 = gb1 = gb2 = (float4_t){0.f, 0.f, 0.f, 0.f};
    float gb3[3] = {0.f, 0.f, 0.f};
    float gbeta_acc = 0.0f;
#if BWD_STAMPS
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last = clock64();
#define STAMP(i) { const unsigned long long now_ = clock64(); st_acc[i] += now_ - st_last; st_last = now_; }
#else
#define STAMP(i)
#endif

    typedef unsigned uint4_t __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t gfrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)g_feat, 0, (int)((unsigned)N * 512u), 0x00020000);
    // four stores the hardware drops (offsets beyond the descriptor's range; distinct, so that none is eliminated): they put
    // "exactly four stores are younger than every load issued before" into the compiler's picture where a loop is entered
    auto dropped_stores = [&]() {
#pragma unroll
        for (int j = 0; j < 4; ++j) __builtin_amdgcn_raw_buffer_store_b128((uint4_t){0u, 0u, 0u, 0u}, gfrsrc, (int)(0xFFFFFF00u + 16u * j), 0, 0);
    };
    // the block's feature-gradient rows, gather role: four 16-byte stores per lane (both tiles: piece j at byte 64 j of the
    // decoder's half row), ALWAYS issued; N * 512 < 2^32 - 256 is checked at the launch
    auto store_rows = [&](const int64_t p0, const int b, const int nvalid, const float gf[16]) {
        if (BWD_ABLATE & 1) return;
        const unsigned off = (16 * b + gp < nvalid) ? (unsigned)(p0 + 16 * b + gp) * 512u + (unsigned)(d * 256 + gq * 16) : 0xFFFFFF00u;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4_t, (float4_t){gf[4 * j], gf[4 * j + 1], gf[4 * j + 2], gf[4 * j + 3]}),
                                                   gfrsrc, (int)off + 64 * j, 0, 0);
    };
    // MLP backward of one tile of <= 64 points starting at point p0; go[] = this decoder's pre-activation output
    // gradients of the lane's point (sample role)
    auto load_block = [&](int b, float4_t v[4], const int64_t pbase) {
        const int64_t pt = min(pbase + 16 * b + gp, N - 1);
        if (LOWP) {      // bf16 features (store_features_lp): channels 8gq..8gq+7 of each level, 16 bytes per level
            const short* fp = (const short*)feat + pt * 128 + d * 64 + 8 * gq;
            v[0] = *(const float4_t*)(fp);
            v[1] = *(const float4_t*)(fp + 32);
            v[2] = v[3] = (float4_t){0.f, 0.f, 0.f, 0.f};
            return;
        }
        if (BWD_ABLATE & 4) { v[0] = v[1] = v[2] = v[3] = (float4_t){0.1f, 0.2f, -0.1f, 0.3f}; return; }
        const float* fp = feat + pt * 128 + d * 64 + 4 * gq;     // piece gq = channels 4gq.., 16+4gq.. of a level
        v[0] = *(const float4_t*)(fp);
        v[1] = *(const float4_t*)(fp + 16);
        v[2] = *(const float4_t*)(fp + 32);
        v[3] = *(const float4_t*)(fp + 48);
    };
#define TOUCH(x) asm volatile("" :: "v"(x))
#define TOUCH_ROWS { TOUCH(fnext[0]); TOUCH(fnext[1]); TOUCH(fnext[2]); TOUCH(fnext[3]); }
    // fnext: the rows of the block that comes next - of this tile, or (BWD_TILE_AHEAD) block 0 of the tile at p0_next, requested
    // by this tile's last block in front of its stores and waited for by the CALLER (TOUCH_ROWS) before it requests anything
    constexpr bool AHEAD = BWD_BUFSTORE && BWD_TILE_AHEAD;
    float4_t fnext[4];
    auto tile_bwd = [&](const int64_t p0, const int nvalid, const float go[4], const int64_t p0_next) {
        const int nblk = (nvalid + 15) >> 4;
        __builtin_assume(nblk >= 1);
#pragma unroll
        for (int o = 0; o < 3; ++o) gb3[o] += go[o];
        *(float4_t*)(gt + lane * 4) = (float4_t){go[0], go[1], go[2], go[3]};

        // gather role: the 16 features of point 16b + gp, piece gq.  The rows of block b+1 are requested before block b is
        // computed: at 2 waves per SIMD nothing else hides the ~2k-cycle load latency.
        if (!AHEAD) load_block(0, fnext, p0);
        // A wave's loads and stores retire in order on ONE counter (vmcnt).  A block's rows are requested a block ahead, i.e.
        // BEFORE the previous block's four feature-gradient stores, so "all but the 4 youngest operations have retired" is all
        // the top of a block has to wait for - but the compiler can only say so if it can COUNT the stores: as predicated
        // global stores behind a branch (rows past the tile's end) it cannot, waits for vmcnt(0), and every block - and every
        // ray's first use of its prefetched inputs - sat out the store acknowledgement of the block before (~2 k and ~5 k
        // cycles: 45 % of a ray's ~30 k, profiles/r02/t_*).  The stores are therefore buffer stores that are ALWAYS issued - rows
        // past the end get an offset beyond the descriptor's range, which the hardware drops - and four such dropped stores
        // behind the first block's loads give the loop's entry the same shape as its back edge.
        if (BWD_BUFSTORE && !AHEAD) dropped_stores();
#pragma unroll 1
        for (int b = 0; b < nblk; ++b) {
            float ft[16];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ft[i] = fnext[0][i]; ft[4 + i] = fnext[1][i]; ft[8 + i] = fnext[2][i]; ft[12 + i] = fnext[3][i];
            }
            // (two blocks ahead: 250 VGPRs, no faster.)  ONE request site with a selected address: two sites behind an if / else
            // made the compiler wait for vmcnt(0)
            if (AHEAD) load_block(0, fnext, b + 1 < nblk ? p0 + 16 * (b + 1) : p0_next);
            else if (b + 1 < nblk) load_block(b + 1, fnext, p0);
            // the same rows again in the "feature on the lane" layout of the g_W1 contraction (B operand): requested
            // here so that the L1/L2 latency is covered by the 28 MFMAs of the recompute instead of stalling them later
            // float32 path: NOT re-read from memory but handed over through a wave-private LDS tile, written here in the
            // gather role and read back just before the contraction.  The global re-read (L1 / L2 hits) was waited for in the
            // middle of the block - behind the previous block's four feature-gradient stores, a wave's loads and stores
            // retiring in order - which exposed those stores' latency in every block (15 us of this kernel at 4096 x 64).
            constexpr bool FBK_LDS = WGRAD && BWD_FBK_LDS != 0;
            float4_t fbk[4];
            short4_t fbkp[4];
            if (FBK_LDS && LOWP) {          // bf16: the lane holds channels 8gq..8gq+7 of each level as 4 + 4 floats' worth of bits
                short* T = (short*)ftile[wave] + gp * TPH + 8 * gq;
                *(float4_t*)(T) = (float4_t){ft[0], ft[1], ft[2], ft[3]};
                *(float4_t*)(T + 32) = (float4_t){ft[4], ft[5], ft[6], ft[7]};
            } else if (FBK_LDS) {
                float* T = ftile[wave] + gp * TPF + 4 * gq;
#pragma unroll
                for (int j = 0; j < 4; ++j) *(float4_t*)(T + 16 * j) = (float4_t){ft[4 * j], ft[4 * j + 1], ft[4 * j + 2], ft[4 * j + 3]};
            } else if (WGRAD) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int64_t pk = min(p0 + 16 * b + 4 * q + ks, N - 1);
                    if (LOWP) fbkp[ks] = *(const short4_t*)((const short*)feat + pk * 128 + d * 64 + 4 * r);
                    else fbk[ks] = (BWD_ABLATE & 4) ? (float4_t){0.1f, 0.2f, 0.3f, 0.4f} : *(const float4_t*)(feat + pk * 128 + d * 64 + 4 * r);
                }
            }
            STAMP(1)
            if (LOWP) to_mfma_role<true, 8>(ft, lane);       // 8 registers of packed bf16 pairs
            else to_mfma_role<true, 16>(ft, lane);
            float4_t h1, h2, gz1, gz2;
            const float4_t zero4 = (float4_t){0.f, 0.f, 0.f, 0.f};
            float gf[16];
            if (LOWP) {
                float4_t a1, a2;
                short8_t bfeat[2];
#pragma unroll
                for (int lvl = 0; lvl < 2; ++lvl)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const unsigned u = __builtin_bit_cast(unsigned, ft[lvl * 4 + i]);
                        bfeat[lvl][2 * i] = (short)(u & 0xFFFFu);
                        bfeat[lvl][2 * i + 1] = (short)(u >> 16);
                    }
                mlp_hidden_lp_bf(fl, bfeat, a1, a2);
#pragma unroll
                for (int i = 0; i < 4; ++i) { h1[i] = fmaxf(a1[i], 0.f); h2[i] = fmaxf(a2[i], 0.f); }
                // g_h2^T = W3^T . g_o^T: B[k = o][col = point] lives on the q == 0 lanes; go is in sample role (lane = point of the tile)
                const float s0 = __shfl(go[0], 16 * b + r, WAVE), s1 = __shfl(go[1], 16 * b + r, WAVE), s2 = __shfl(go[2], 16 * b + r, WAVE);
                const short4_t bgo = (q == 0) ? pack4(s0, s1, s2, 0.f) : (short4_t){0, 0, 0, 0};
                const float4_t gh2 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(w3colp, bgo, zero4, 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) gz2[i] = a2[i] > 0.0f ? gh2[i] : 0.0f;
                const float4_t gh1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(w2tp, pack4(gz2[0], gz2[1], gz2[2], gz2[3]), zero4, 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) gz1[i] = a1[i] > 0.0f ? gh1[i] : 0.0f;
                gb1 += gz1;
                gb2 += gz2;
                // g_feat^T = W1^T . g_z1^T: row block mb = features 16 mb .. 16 mb + 15, lane (r, q) receives 16 mb + 4q + reg
                const short4_t bz1 = pack4(gz1[0], gz1[1], gz1[2], gz1[3]);
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) {
                    const float4_t acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(w1tp[mb], bz1, zero4, 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) gf[4 * mb + i] = acc[i];
                }
                to_gather_role<true, 16>(gf, lane);
                if (BWD_BUFSTORE) store_rows(p0, b, nvalid, gf);
                else if (p0 + 16 * b + gp < p0 + nvalid) {       // gather-role lane (point, g): features 16 mb + 4 g + i
                    float* dst = g_feat + (p0 + 16 * b + gp) * 128 + d * 64 + 4 * gq;
#pragma unroll
                    for (int mb = 0; mb < 4; ++mb)
                        *(float4_t*)(dst + 16 * mb) = (float4_t){gf[4 * mb], gf[4 * mb + 1], gf[4 * mb + 2], gf[4 * mb + 3]};
                }
            } else {
            mlp_hidden(f, ft, h1, h2);
            STAMP(2)

            // g_h2^T = W3^T . g_o^T   (K = (block', o); only block' == b contributes)
            float4_t gh2 = (float4_t){0.f, 0.f, 0.f, 0.f};
            const bool mine = (q == b);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) gh2 = mfma16(mine ? w3col[ks] : 0.0f, go[ks], gh2);
#pragma unroll
            for (int i = 0; i < 4; ++i) gz2[i] = h2[i] > 0.0f ? gh2[i] : 0.0f;
            // g_h1^T = W2^T . g_z2^T
            float4_t gh1 = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) gh1 = mfma16(w2t[ks], gz2[ks], gh1);
#pragma unroll
            for (int i = 0; i < 4; ++i) gz1[i] = h1[i] > 0.0f ? gh1[i] : 0.0f;
            gb1 += gz1;
            gb2 += gz2;

            // g_feat^T = W1^T . g_z1^T, four row blocks ordered so that MFMA-role lane (r, q) receives piece q of point r
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                float4_t acc = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) acc = mfma16(w1t[mb][ks], gz1[ks], acc);
#pragma unroll
                for (int i = 0; i < 4; ++i) gf[(mb >> 1) * 8 + 4 * (mb & 1) + i] = acc[i];
            }
            STAMP(3)
            to_gather_role<true, 16>(gf, lane);
            if (BWD_BUFSTORE) store_rows(p0, b, nvalid, gf);
            else if (!(BWD_ABLATE & 1) && p0 + 16 * b + gp < p0 + nvalid) store_features(g_feat, p0 + 16 * b + gp, d, gq, gf);
            }
            STAMP(4)
            if (!WGRAD || (BWD_ABLATE & 2)) continue;

            // transposes through LDS: D layout (rows 4q+reg, col = point r) -> [point][row]
            *(float4_t*)(tz1 + r * TP + 4 * q) = gz1;
            *(float4_t*)(tz2 + r * TP + 4 * q) = gz2;
            *(float4_t*)(th1 + r * TP + 4 * q) = h1;
            *(float4_t*)(th2 + r * TP + 4 * q) = h2;
            WAVE_SYNC();

            // parameter gradients: contraction over the block's 16 points (K = point 4q + ks)
            float az1[4], az2[4], bh1[4], bh2[4], ago[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int prow = 4 * q + ks;
                az1[ks] = tz1[prow * TP + r];
                az2[ks] = tz2[prow * TP + r];
                bh1[ks] = th1[prow * TP + r];
                bh2[ks] = th2[prow * TP + r];
                ago[ks] = (r < 4) ? gt[(16 * b + prow) * 4 + r] : 0.0f;
            }
            STAMP(5)
            if (LOWP) {      // the same contractions over the block's 16 points, K = 16 in one bf16 MFMA each
                gW2 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pack4(az2[0], az2[1], az2[2], az2[3]),
                                                                pack4(bh1[0], bh1[1], bh1[2], bh1[3]), gW2, 0, 0, 0);
                gW3 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pack4(ago[0], ago[1], ago[2], ago[3]),
                                                                pack4(bh2[0], bh2[1], bh2[2], bh2[3]), gW3, 0, 0, 0);
                const short4_t a1p = pack4(az1[0], az1[1], az1[2], az1[3]);
                if (FBK_LDS) {
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) fbkp[ks] = *(const short4_t*)((const short*)ftile[wave] + (4 * q + ks) * TPH + 4 * r);
                }
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
                    gW1[nb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a1p, (short4_t){fbkp[0][nb], fbkp[1][nb], fbkp[2][nb], fbkp[3][nb]},
                                                                        gW1[nb], 0, 0, 0);
            } else {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                gW2 = mfma16(az2[ks], bh1[ks], gW2);              // g_W2[j][k'] : rows j = 4q+reg, col k' = r
                gW3 = mfma16(ago[ks], bh2[ks], gW3);              // g_W3[o][j]  : rows o = 4q+reg (q = 0), col j = r
            }
            // g_W1[j][f]: B = features of point 4q+ks, columns permuted: column c of n-block nb <-> feature 4c + nb
            if (FBK_LDS) {      // (the tile was written before this block's first WAVE_SYNC)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) fbk[ks] = *(const float4_t*)(ftile[wave] + (4 * q + ks) * TPF + 4 * r);
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) gW1[nb] = mfma16(az1[ks], fbk[ks][nb], gW1[nb]);
            }
            }
            WAVE_SYNC();
            STAMP(6)
        }
    };

    if (MODE == 0) {
        const int64_t ntiles = (N + 63) / 64;
        int64_t tile = (int64_t)blockIdx.x * 4 + wave;
        if (AHEAD && tile < ntiles) {
            load_block(0, fnext, tile * 64);
            dropped_stores();
        }
        for (; tile < ntiles; tile += (int64_t)gridDim.x * 4) {
            const int64_t p0 = tile * 64;
            const int nvalid = (int)min((int64_t)64, N - p0);
            // sample role: pre-activation output gradients of this decoder
            float go[4] = {0.f, 0.f, 0.f, 0.f};
            if (lane < nvalid) {
                const float4_t g = *(const float4_t*)(g_o + (p0 + lane) * 4);
                if (d == 0) go[0] = g[3];
                else { go[0] = g[0]; go[1] = g[1]; go[2] = g[2]; }
            }
            if (AHEAD) TOUCH_ROWS
            tile_bwd(p0, nvalid, go, (tile + (int64_t)gridDim.x * 4) * 64);
        }
    } else {
        const int R = rb.R, S = rb.S;
        const int nchunk = (S + WAVE - 1) / WAVE;
        __builtin_assume(nchunk >= 1);             // (S >= 1 is checked at the launch; lets the compiler count a ray's stores)
        const float beta = rb.beta[0];
        LossW lw = {};
        LossK lk = {};
        float nd = 1.0f, ncol = 1.0f;
        if (MODE == 2) {
            lw = loss_scaled_weights(li.w, li.upstream);
            lk = loss_sdf_factors(lw, li.tr, li.acc);
            nd = li.acc[A_N_DEPTH]; ncol = li.acc[A_N_COLOR];
            if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && li.loss_out)
                li.loss_out[0] = loss_value_from_acc(li.w, li.acc);
        }
        // What the composite backward of a ray reads is REQUESTED ONE RAY AHEAD: a persistent wave walks its rays one after the
        // other, and a load -> scan -> decoder chain per ray put ~2.5 us of exposed latency in front of every ray's MFMAs
        // (8192 x 96: 228 us fused against 208 us for the three launches it replaced).  A wave's loads retire in issue
        // order, so the prefetched values have arrived by the time the current ray's later feature loads are waited for.
        struct RayIn { float gd, gr, gg, gb, gtd, dep, cr, cg, cb, tr, tg, tb; int mask; };
        struct ChunkIn { float sd, z, cr, cg, cb, gs; };
        auto load_ray = [&](int ray) {
            RayIn x;
            x.gd = rb.g_depth ? rb.g_depth[ray] : 0.0f;
            x.gr = rb.g_rgb ? rb.g_rgb[3 * ray + 0] : 0.0f;
            x.gg = rb.g_rgb ? rb.g_rgb[3 * ray + 1] : 0.0f;
            x.gb = rb.g_rgb ? rb.g_rgb[3 * ray + 2] : 0.0f;
            x.gtd = x.dep = x.cr = x.cg = x.cb = x.tr = x.tg = x.tb = 0.0f;
            x.mask = 1;
            if (MODE == 2) {
                x.gtd = li.gt_depth[ray];
                x.mask = li.ray_mask ? (int)li.ray_mask[ray] : 1;      // the byte as loaded: comparing it here would wait for it here
                x.dep = li.depth[ray];
                x.cr = li.rgb[3 * ray + 0]; x.cg = li.rgb[3 * ray + 1]; x.cb = li.rgb[3 * ray + 2];
                x.tr = li.gt_color[3 * ray + 0]; x.tg = li.gt_color[3 * ray + 1]; x.tb = li.gt_color[3 * ray + 2];
            }
            return x;
        };
        auto load_chunk = [&](int ray, int c) {
            ChunkIn x = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            const int s = c * WAVE + lane;
            if (s < S) {
                const int64_t i = (int64_t)ray * S + s;
                x.sd = rb.sdf[i];
                x.z = rb.z_vals[i];
                x.cr = rb.raw_rgb[i * 3 + 0];
                x.cg = rb.raw_rgb[i * 3 + 1];
                x.cb = rb.raw_rgb[i * 3 + 2];
                if (rb.g_sdf) x.gs = rb.g_sdf[i];
            }
            return x;
        };
        // S > 64: the sdf values of the chunks in front of the last one (pass A below), requested a ray ahead like the rest
        static_assert(ESLAM_MAX_SAMPLES <= 4 * WAVE, "pass A prefetch holds three chunks");
        struct PassA { float sd[3]; };
        auto load_pa = [&](int ray) {
            PassA x = {{0.f, 0.f, 0.f}};
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (k < nchunk - 1) x.sd[k] = rb.sdf[(int64_t)ray * S + k * WAVE + lane];      // chunks before the last are full
            return x;
        };
        const int stride = gridDim.x * 4;
        int ray = blockIdx.x * 4 + wave;
        RayIn rin = {};
        ChunkIn cin = {};
        PassA pin = {{0.f, 0.f, 0.f}};
        if (ray < R) {
            rin = load_ray(ray);
            cin = load_chunk(ray, nchunk - 1);
            pin = load_pa(ray);
            if (AHEAD) load_block(0, fnext, (int64_t)ray * S + (nchunk - 1) * WAVE);
        }
        constexpr bool COUNTED = BWD_BUFSTORE != 0;           // the waits below are counted by the compiler: see tile_bwd
        if (COUNTED) dropped_stores();
#define TOUCH_CHUNK(ci) { TOUCH(ci.sd); TOUCH(ci.z); TOUCH(ci.cr); TOUCH(ci.cg); TOUCH(ci.cb); TOUCH(ci.gs); }
        for (; ray < R; ray += stride) {
            const int64_t base = (int64_t)ray * S;
            // What is requested ahead - this ray's scalars and its last chunk a ray ago, a further chunk a tile ago - is waited
            // for HERE, in front of the next requests and behind the four stores that ended the tile in between: vmcnt(4).
            // (Waited for at its first use further down, the wait would cover the requests issued meanwhile as well - the
            // compiler cannot count those: optional pointers, rows past S - and expose their whole latency in every ray:
            // 5.5 k cycles of a ray's ~30 k, profiles/r02/t_*.)
            if (COUNTED) {
                TOUCH(rin.gd); TOUCH(rin.gr); TOUCH(rin.gg); TOUCH(rin.gb); TOUCH(rin.gtd); TOUCH(rin.dep); TOUCH(rin.cr); TOUCH(rin.cg);
                TOUCH(rin.cb); TOUCH(rin.tr); TOUCH(rin.tg); TOUCH(rin.tb); TOUCH(rin.mask);
                TOUCH_CHUNK(cin)
                TOUCH(pin.sd[0]); TOUCH(pin.sd[1]); TOUCH(pin.sd[2]);
                if (AHEAD) TOUCH_ROWS
            }
            ChunkIn ch = cin;
            RayIn rnx = {};
            ChunkIn cnx = {};
            PassA pnx = {{0.f, 0.f, 0.f}};
            if (ray + stride < R) { rnx = load_ray(ray + stride); cnx = load_chunk(ray + stride, nchunk - 1); pnx = load_pa(ray + stride); }
            RayUp up;
            up.gd = rin.gd; up.gr = rin.gr; up.gg = rin.gg; up.gb = rin.gb;
            const float gtd = rin.gtd;
            bool m = false;
            if (MODE == 2) {
                const bool mc = rin.mask != 0;
                m = mc && gtd > 0.0f;
                up.gd += loss_g_depth(m, gtd, rin.dep, lw, nd);
                up.gr += loss_g_color(mc, rin.tr, rin.cr, lw, ncol);
                up.gg += loss_g_color(mc, rin.tg, rin.cg, lw, ncol);
                up.gb += loss_g_color(mc, rin.tb, rin.cb, lw, ncol);
            }
            // pass A: transmittance product of every chunk; lane c keeps chunk c's
            float myprod = 1.0f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {                    // (the last chunk's product is never needed)
                if (c < nchunk - 1) {
                    const float sd = pin.sd[c];
                    const float alpha = 1.0f - expf(-beta * sigmoidf_(-sd * beta));
                    const float cp = wave_lane<63>(wave_incl_prod((1.0f - alpha) + 1e-10f, lane));
                    if (lane == c) myprod = cp;
                }
            }
            // pass B: chunks in reverse, carrying the suffix sum of gw*w
            float carry = 0.0f;
            int c = nchunk - 1;
            do {
                // S > 64: the chunk in front of this one is requested a tile ahead as well, and taken over behind the tile's stores.
                // (No value still in flight is carried around a loop: the compiler copies loop-carried registers at the loop's
                // entry, and a copy of a register that is still being loaded is a wait for everything requested so far.)
                ChunkIn cn = {};
                if (c > 0) cn = load_chunk(ray, c - 1);
                float trans_in = 1.0f;
                for (int k = 0; k < c; ++k) trans_in *= __shfl(myprod, k, WAVE);
                const bool valid = c * WAVE + lane < S;
                float gs = ch.gs;
                if (MODE == 2 && valid) gs += loss_g_sdf(m, ch.z, ch.sd, gtd, li.tr, lk);
                STAMP(7)
                const float4_t o = composite_bwd_chunk(up, beta, valid, ch.sd, ch.z, ch.cr, ch.cg, ch.cb, gs, trans_in, carry,
                                                       gbeta_acc, lane);
                float go[4] = {0.f, 0.f, 0.f, 0.f};
                if (d == 0) go[0] = o[3];
                else { go[0] = o[0]; go[1] = o[1]; go[2] = o[2]; }
                STAMP(0)
                // the tile after this one: the chunk in front of it, or the last chunk of the wave's next ray (rows past N are clamped)
                tile_bwd(base + c * WAVE, min(WAVE, S - c * WAVE), go,
                         c > 0 ? base + (c - 1) * WAVE : (int64_t)(ray + stride) * S + (nchunk - 1) * WAVE);
                if (c > 0) {
                    if (COUNTED) TOUCH_CHUNK(cn)
                    if (AHEAD) TOUCH_ROWS
                    ch = cn;
                }
            } while (--c >= 0);
            rin = rnx;
            cin = cnx;
            pin = pnx;
        }
#undef TOUCH_CHUNK
        // g_beta: one partial sum per workgroup of the sdf decoder (summed by dec_grad_reduce_kernel / beta_sum_kernel);
        // one atomic per ray on the single address of g_beta would serialise at the memory side (4096 of them: 50 us)
        __shared__ float beta_part[4];
        const float tot = wave_sum(gbeta_acc);
        if (lane == 0) beta_part[wave] = tot;
        __syncthreads();
        if (d == 0 && threadIdx.x == 0 && rb.beta_parts)
            rb.beta_parts[blockIdx.x] = (beta_part[0] + beta_part[1]) + (beta_part[2] + beta_part[3]);
    }

#if BWD_STAMPS
    if (blockIdx.x == 5 && threadIdx.x == 0)
        printf("stamps d=%d cycles: composite %llu | top of a block (wait for its feature rows) %llu | recompute issue %llu | backward mfma issue %llu | "
               "role change + stores (waits for the MFMA chain) %llu | lds transposes %llu | weight gradients %llu | ray prologue %llu\n",
               d, st_acc[0], st_acc[1], st_acc[2], st_acc[3], st_acc[4], st_acc[5], st_acc[6], st_acc[7]);
#endif
    if (!WGRAD) return;
    // the four waves' partial parameter gradients are summed in LDS; one slab row per workgroup: [wg][decoder][SLAB]
    __shared__ __attribute__((aligned(16))) float comb[4][SLAB];
    float* sl = comb[wave];
    for (int i = lane; i < SLAB; i += WAVE) sl[i] = 0.0f;       // unwritten padding columns must not hold NaN bit patterns
    WAVE_SYNC();
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int j = 4 * q + reg;
        float4_t v;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) v[nb] = gW1[nb][reg];
        *(float4_t*)(sl + SL_W1 + j * 64 + 4 * r) = v;
        sl[SL_W2 + j * 16 + r] = gW2[reg];
        if (q == 0 && reg < nout) sl[SL_W3 + reg * 16 + r] = gW3[reg];
    }
    // bias gradients: sum over the 16 point lanes
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        float a = gb1[reg], c = gb2[reg];
#pragma unroll
        for (int m = 8; m >= 1; m >>= 1) {
            a += __shfl_xor(a, m, WAVE);
            c += __shfl_xor(c, m, WAVE);
        }
        if (r == 0) {
            sl[SL_B1 + 4 * q + reg] = a;
            sl[SL_B2 + 4 * q + reg] = c;
        }
    }
#pragma unroll
    for (int o = 0; o < 3; ++o) {
        const float t = wave_sum(gb3[o]);
        if (lane == 0 && o < nout) sl[SL_B3 + o] = t;
    }
    __syncthreads();
    float* row = slabs + ((int64_t)blockIdx.x * 2 + d) * SLAB;
    for (int col = threadIdx.x; col < SLAB; col += 256)     // columns no wave writes (padding) are never read back
        row[col] = (comb[0][col] + comb[1][col]) + (comb[2][col] + comb[3][col]);
}

__global__ __launch_bounds__(1024) void beta_sum_kernel(const float* __restrict__ parts, int n, float* __restrict__ out) {
    __shared__ float bsum[16];
    float a = 0.f;
    for (int i = threadIdx.x; i < n; i += 1024) a += parts[i];
    a = wave_sum(a);
    if ((threadIdx.x & 63) == 0) bsum[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int k = 0; k < 16; ++k) t += bsum[k];
        out[0] = t;
    }
}

// slabs [nrows][2][SLAB] -> g_dec: eslam_dec_reduce.h.  Stand-alone launch, used when the scatter cannot carry the work
// (no plane gradients requested, deterministic mode, free points).
__global__ __launch_bounds__(1024) void dec_grad_reduce_kernel(const DecReduceArgs a) {
    __shared__ float red[16 * 64];
    dec_grad_reduce_block<1024>(a, blockIdx.x, blockIdx.y, red);
}

// ---------------------------------------------------------------------------------------------------------
// gradient w.r.t. sample positions (autograd of the grid coordinates + common.py:215-217 + Renderer.py:136-137)
// ---------------------------------------------------------------------------------------------------------
template <bool CL>
__device__ __forceinline__ void coord_grad8(const eslam_plane_t& P, float u, float v, int q, const float g[8],
                                            float& gu, float& gv) {
    // q: the lane's piece of the texel (gather role): channels piece_channel(q, 0..7)
    const AxisCoord ax = axis_coord(u, P.w);
    const AxisCoord ay = axis_coord(v, P.h);
    const int sy = (int)P.stride_y, sx = (int)P.stride_x, sc = (int)P.stride_c;
    const int r0 = ay.i0 * sy, r1 = ay.i1 * sy, c0 = ax.i0 * sx, c1 = ax.i1 * sx;
    float su = 0.f, sv = 0.f;
    if (CL) {
        const float* base = P.data + 4 * q;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const float4_t t00 = *(const float4_t*)(base + r0 + c0 + 16 * hh);
            const float4_t t01 = *(const float4_t*)(base + r0 + c1 + 16 * hh);
            const float4_t t10 = *(const float4_t*)(base + r1 + c0 + 16 * hh);
            const float4_t t11 = *(const float4_t*)(base + r1 + c1 + 16 * hh);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                su += g[4 * hh + i] * ((t01[i] - t00[i]) * (1.0f - ay.t) + (t11[i] - t10[i]) * ay.t);
                sv += g[4 * hh + i] * ((t10[i] - t00[i]) * (1.0f - ax.t) + (t11[i] - t01[i]) * ax.t);
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float* b = P.data + piece_channel(q, i) * sc;
            const float t00 = b[r0 + c0], t01 = b[r0 + c1], t10 = b[r1 + c0], t11 = b[r1 + c1];
            su += g[i] * ((t01 - t00) * (1.0f - ay.t) + (t11 - t10) * ay.t);
            sv += g[i] * ((t10 - t00) * (1.0f - ax.t) + (t11 - t01) * ax.t);
        }
    }
    // d(ix)/du = (w-1)/2 strictly inside, 0 where the border clamp is active (ATen clip_coordinates_set_grad)
    gu += ax.inside ? su * (0.5f * (float)(P.w - 1)) : 0.0f;
    gv += ay.inside ? sv * (0.5f * (float)(P.h - 1)) : 0.0f;
}

template <bool CL, bool RENDER>
__global__ __launch_bounds__(256, 2) void coord_bwd_kernel(const PlaneSet planes, const Bound bnd,
                                                        const float* __restrict__ rays_o,
                                                        const float* __restrict__ rays_d,
                                                        const float* __restrict__ z_vals, int R, int S,
                                                        const float* __restrict__ g_feat,
                                                        float* __restrict__ g_rays_o, float* __restrict__ g_rays_d) {
    // RENDER: one wave per ray, outputs g_rays_o/g_rays_d [R,3].
    // else:   one wave per 64 points (z_vals = pts [N,3], R = N), output g_rays_o = g_pts [N,3].
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = gather_point<CL>(lane), q = gather_piece<CL>(lane);     // gather role (eslam_decode_tile.h)
    const int64_t unit = (int64_t)blockIdx.x * 4 + wave;
    const int64_t nunits = RENDER ? R : ((int64_t)R + 63) / 64;
    if (unit >= nunits) return;
    float ox = 0.f, oy = 0.f, oz = 0.f, dx = 0.f, dy = 0.f, dz = 0.f;
    if (RENDER) {
        ox = rays_o[unit * 3 + 0]; oy = rays_o[unit * 3 + 1]; oz = rays_o[unit * 3 + 2];
        dx = rays_d[unit * 3 + 0]; dy = rays_d[unit * 3 + 1]; dz = rays_d[unit * 3 + 2];
    }
    const int64_t base = RENDER ? unit * S : unit * 64;
    const int scount = RENDER ? S : (int)min((int64_t)64, (int64_t)R - base);
    const float sc3[3] = {2.0f / (bnd.hi[0] - bnd.lo[0]), 2.0f / (bnd.hi[1] - bnd.lo[1]), 2.0f / (bnd.hi[2] - bnd.lo[2])};
    float go_acc[3] = {0.f, 0.f, 0.f}, gd_acc[3] = {0.f, 0.f, 0.f};
#pragma unroll 1
    for (int s0 = 0; s0 < scount; s0 += 16) {
        const int oz0 = opaque_zero(s0);      // keeps the 12 planes' scalar loads inside this loop (see gather_features)
        const int s = s0 + r;
        const bool valid = s < scount;
        const int sc_ = min(s, scount - 1);
        float x, y, z, zz = 0.f;
        if (RENDER) {
            zz = z_vals[base + sc_];
            x = ox + dx * zz; y = oy + dy * zz; z = oz + dz * zz;
        } else {
            x = z_vals[(base + sc_) * 3 + 0]; y = z_vals[(base + sc_) * 3 + 1]; z = z_vals[(base + sc_) * 3 + 2];
        }
        x = norm_coord(x, bnd.lo[0], bnd.hi[0]);
        y = norm_coord(y, bnd.lo[1], bnd.hi[1]);
        z = norm_coord(z, bnd.lo[2], bnd.hi[2]);
        float gp[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int d = 0; d < 2; ++d) {
#pragma unroll
            for (int lvl = 0; lvl < 2; ++lvl) {
                const float* gfp = g_feat + (base + sc_) * 128 + d * 64 + lvl * 32 + 4 * q;
                const float4_t ga = *(const float4_t*)gfp, gb = *(const float4_t*)(gfp + 16);
                const float g[8] = {ga[0], ga[1], ga[2], ga[3], gb[0], gb[1], gb[2], gb[3]};
#pragma unroll
                for (int o = 0; o < 3; ++o) {
                    const eslam_plane_t& P = planes.p[2 * (3 * d + o) + lvl + oz0];
                    float gu = 0.f, gv = 0.f;
                    coord_grad8<CL>(P, ORIENT_U(o, x, y, z), ORIENT_V(o, x, y, z), q, g, gu, gv);
                    __builtin_amdgcn_sched_barrier(0);
                    gp[o == 2 ? 1 : 0] += gu;      // first coordinate: x (xy, xz) or y (yz)
                    gp[o == 0 ? 1 : 2] += gv;      // second coordinate: y (xy) or z (xz, yz)
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float v = gp[k];
            v += __shfl_xor(v, CL ? 1 : 16, WAVE);  // sum the four pieces of the point
            v += __shfl_xor(v, CL ? 2 : 32, WAVE);
            v = valid ? v * sc3[k] : 0.0f;
            if (RENDER) {
                if (q == 0) { go_acc[k] += v; gd_acc[k] += v * zz; }
            } else if (q == 0 && valid) {
                g_rays_o[(base + s) * 3 + k] = v;
            }
        }
    }
    if (RENDER) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float a = wave_sum(go_acc[k]);
            const float b = wave_sum(gd_acc[k]);
            if (lane == 0) {
                g_rays_o[unit * 3 + k] = a;
                g_rays_d[unit * 3 + k] = b;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------
bool eslam_planes_channels_last(const eslam_plane_t* planes, int first, int count);
int eslam_validate_planes(const eslam_plane_t* planes, int first, int count);
int eslam_planes_lowp(const eslam_plane_t* planes);

static Bound make_bound(const float* b6) {
    Bound b;
    for (int k = 0; k < 3; ++k) {
        b.lo[k] = b6[2 * k];
        b.hi[k] = b6[2 * k + 1];
    }
    return b;
}

#define MLP_BWD_MAX_WG 256

int eslam_scatter_v2(const eslam_plane_t* planes, const Bound& bnd, const float* rays_o, const float* rays_d,
                     const float* z_or_pts, int64_t R, int S, bool render, const float* g_feat, const int* perm,
                     hipStream_t st, const DecReduceArgs* red);
bool eslam_scatter_can_reduce(bool render);


static int64_t align256(int64_t v) { return (v + 255) & ~(int64_t)255; }

// workspace layout: g_o [n,4] | g_feat [n,128] | slabs [MLP_BWD_MAX_WG*4][2][SLAB] + g_beta partials [n/4 + 1] |
//                   ray order [n] (int)
static int64_t slab_region_bytes(int64_t n_points) {
    return ((int64_t)MLP_BWD_MAX_WG * 4 * 2 * SLAB + n_points / 4 + 1) * 4;      // (g_beta partials need MLP_BWD_MAX_WG only)
}

extern "C" int64_t eslam_bwd_workspace_bytes(int64_t n_points) {
    if (n_points < 0) return -1;
    return align256(n_points * 4 * 4) + align256(n_points * 128 * 4) +
           align256(slab_region_bytes(n_points)) + align256(n_points * 4);
}

// mode 0: free points (g_o in the workspace); 1: rays, upstream gradients in rb; 2: rays, mapping-loss gradients from li
static int bwd_common(const eslam_plane_t* planes, const eslam_decoders_t* dec, const Bound& bnd, const float* rays_o,
                      const float* rays_d, const float* z_or_pts, int64_t R, int S, int mode, const float* feat,
                      float* g_o, float* g_feat, float* slabs, const int* perm, float* g_dec, float* g_out_a, float* g_out_b,
                      RayBwdIn rb, const LossGradIn* li, float* g_beta, hipStream_t st) {
    const bool render = mode != 0;
    const int64_t N = render ? R * S : R;
    PlaneSet ps;
    for (int i = 0; i < NPL; ++i) ps.p[i] = planes[i];
    const bool cl = eslam_planes_channels_last(planes, 0, NPL);

    // decoder MLP backward (rays: with the composite backward, and the loss gradient, in the same launch)
    const int64_t ntiles = render ? R : (N + 63) / 64;                 // a wave's unit of work: a ray, or 64 free points
    const int nwg = (int)((ntiles + 3) / 4 < MLP_BWD_MAX_WG ? (ntiles + 3) / 4 : MLP_BWD_MAX_WG);
    float* beta_parts = slabs + (int64_t)MLP_BWD_MAX_WG * 4 * 2 * SLAB;
    rb.beta_parts = render ? beta_parts : nullptr;
    const LossGradIn none = {};
    if (N * 512 >= ((int64_t)1 << 32) - 256) {     // the feature-gradient rows are addressed with 32-bit byte offsets (as in the scatter)
        eslam_set_error("backward: %lld points exceed the 32-bit offset range of the feature-gradient buffer (8.3 M): split the batch",
                        (long long)N);
        return 1;
    }
    eslam_prof_begin(PROF_MLP_BWD, st);
#define LAUNCH_MB(MD, WG, LP) \
    hipLaunchKernelGGL((mlp_bwd_kernel<MD, WG, LP>), dim3(nwg, 2), dim3(256), 0, st, *dec, feat, g_o, N, g_feat, slabs, rb, \
                       li ? *li : none)
    const int lowp = eslam_planes_lowp(planes);
    if (lowp < 0) return 1;
    if (lowp) {
        if (mode == 0 || g_out_a) {
            eslam_set_error("mixed precision: only the ray backward to planes and decoders is built (no point / pose gradients)");
            return 1;
        }
        if (mode == 1) { if (g_dec) LAUNCH_MB(1, true, true); else LAUNCH_MB(1, false, true); }
        else { if (g_dec) LAUNCH_MB(2, true, true); else LAUNCH_MB(2, false, true); }
    }
    else if (mode == 0) { if (g_dec) LAUNCH_MB(0, true, false); else LAUNCH_MB(0, false, false); }
    else if (mode == 1) { if (g_dec) LAUNCH_MB(1, true, false); else LAUNCH_MB(1, false, false); }
    else { if (g_dec) LAUNCH_MB(2, true, false); else LAUNCH_MB(2, false, false); }
#undef LAUNCH_MB
    eslam_prof_end(PROF_MLP_BWD, st);
    if (int rc = eslam_check_launch("mlp_bwd_kernel")) return rc;
    const int n_beta_parts = render ? nwg : 0;
    bool any_grad = false, all_grad = true;
    for (int i = 0; i < NPL; ++i) {
        any_grad |= planes[i].grad != nullptr;
        all_grad &= planes[i].grad != nullptr;
    }
    if (any_grad && !all_grad) {
        eslam_set_error("plane gradients must be requested for all 12 planes or for none");
        return 1;
    }
    DecReduceArgs red = {};
    red.slabs = slabs; red.nrows = nwg; red.g_dec = g_dec;
    red.beta_parts = render ? beta_parts : nullptr; red.n_beta_parts = n_beta_parts; red.g_beta = g_beta;
    // The slab reduction (44 workgroups, latency-bound, ~7 us as a launch of its own) rides in the scatter's grid when there
    // is one: its workgroups run beside the scatter's first ones instead of in front of them.
    const bool fold = g_dec && any_grad && eslam_scatter_can_reduce(render);
    if (g_dec && !fold) {
        eslam_prof_begin(PROF_DEC_REDUCE, st);
        hipLaunchKernelGGL(dec_grad_reduce_kernel, dim3(DEC_RED_COLBLOCKS, 2), dim3(1024), 0, st, red);
        eslam_prof_end(PROF_DEC_REDUCE, st);
        if (int rc = eslam_check_launch("dec_grad_reduce_kernel")) return rc;
    } else if (!g_dec && g_beta && render) {      // beta's gradient does not go through the decoders: still sum its partials
        hipLaunchKernelGGL(beta_sum_kernel, dim3(1), dim3(1024), 0, st, beta_parts, n_beta_parts, g_beta);
        if (int rc = eslam_check_launch("beta_sum_kernel")) return rc;
    }

    // plane gradients
    if (any_grad) {
        eslam_prof_begin(PROF_SCATTER, st);
        if (int rc = eslam_scatter_v2(planes, bnd, rays_o, rays_d, z_or_pts, R, S, render, g_feat, perm, st, fold ? &red : nullptr))
            return rc;
        eslam_prof_end(PROF_SCATTER, st);
    }
    // position gradients
    if (g_out_a) {
        const int64_t nunits = render ? R : (N + 63) / 64;
        dim3 grid((unsigned)((nunits + 3) / 4)), block(256);
#define LAUNCH(CLv, RD)                                                                                            \
    hipLaunchKernelGGL((coord_bwd_kernel<CLv, RD>), grid, block, 0, st, ps, bnd, rays_o, rays_d, z_or_pts, (int)R, \
                       S, g_feat, g_out_a, g_out_b)
        eslam_prof_begin(PROF_COORD_BWD, st);
        if (cl && render) LAUNCH(true, true);
        else if (cl) LAUNCH(true, false);
        else if (render) LAUNCH(false, true);
        else LAUNCH(false, false);
#undef LAUNCH
        eslam_prof_end(PROF_COORD_BWD, st);
        if (int rc = eslam_check_launch("coord_bwd_kernel")) return rc;
    }
    return 0;
}

static int render_bwd_impl(const char* who, const eslam_plane_t* planes, const eslam_decoders_t* dec,
                           const float* bound6_host, const float* rays_o, const float* rays_d, const float* z_vals, int R,
                           int S, const float* sdf, const float* raw_rgb, const float* feat, const float* g_depth,
                           const float* g_rgb, const float* g_sdf, const LossGradIn* li, float* g_dec, float* g_beta,
                           float* g_rays_o, float* g_rays_d, const int32_t* ray_order, void* workspace,
                           eslam_stream_t stream) {
    if (R <= 0) return 0;
    if (S <= 0 || S > ESLAM_MAX_SAMPLES) {
        eslam_set_error("%s: S=%d outside [1,%d]", who, S, ESLAM_MAX_SAMPLES);
        return 1;
    }
    if (!planes || !dec || !bound6_host || !rays_o || !rays_d || !z_vals || !sdf || !raw_rgb || !feat || !workspace) {
        eslam_set_error("%s: null argument", who);
        return 1;
    }
    if ((g_rays_o == nullptr) != (g_rays_d == nullptr)) {
        eslam_set_error("%s: g_rays_o and g_rays_d must both be given or both be NULL", who);
        return 1;
    }
    if ((int64_t)R * S * 128 >= ((int64_t)1 << 40)) {
        eslam_set_error("%s: batch too large", who);
        return 1;
    }
    if (eslam_validate_planes(planes, 0, NPL)) return 1;
    hipStream_t st = (hipStream_t)stream;
    const int64_t N = (int64_t)R * S;
    char* ws = (char*)workspace;
    float* g_o = (float*)ws;
    float* g_feat = (float*)(ws + align256(N * 16));
    float* slabs = (float*)(ws + align256(N * 16) + align256(N * 512));
    const int* perm = ray_order;
    bool scatter_runs = false;         // the order bundles rays for the plane-gradient scatter: tracking (no plane gradients) has no use for it
    for (int i = 0; i < NPL; ++i) scatter_runs = scatter_runs || planes[i].grad != nullptr;
    if (!perm && scatter_runs && (int64_t)ESLAM_RAY_ORDER_WORDS(R) <= N) {      // the forward pass did not leave the orders: compute them here (in a region of R S ints)
        int* own = (int*)(ws + align256(N * 16) + align256(N * 512) + align256(slab_region_bytes(N)));
        if (int rc = eslam_ray_order(rays_o, rays_d, R, own, st)) return rc;
        perm = own;
    }
    const Bound bnd = make_bound(bound6_host);
    RayBwdIn rb = {};
    rb.z_vals = z_vals; rb.sdf = sdf; rb.raw_rgb = raw_rgb; rb.beta = dec->beta;
    rb.g_depth = g_depth; rb.g_rgb = g_rgb; rb.g_sdf = g_sdf; rb.R = R; rb.S = S;
    return bwd_common(planes, dec, bnd, rays_o, rays_d, z_vals, R, S, li ? 2 : 1, feat, g_o, g_feat, slabs, perm, g_dec,
                      g_rays_o, g_rays_d, rb, li, g_beta, st);
}

extern "C" int eslam_render_bwd(const eslam_plane_t* planes, const eslam_decoders_t* dec, const float* bound6_host,
                                const float* rays_o, const float* rays_d, const float* z_vals, int R, int S,
                                const float* sdf, const float* raw_rgb, const float* feat, const float* g_depth,
                                const float* g_rgb, const float* g_sdf, float* g_dec, float* g_beta, float* g_rays_o,
                                float* g_rays_d, const int32_t* ray_order, void* workspace, eslam_stream_t stream) {
    return render_bwd_impl("eslam_render_bwd", planes, dec, bound6_host, rays_o, rays_d, z_vals, R, S, sdf, raw_rgb, feat,
                           g_depth, g_rgb, g_sdf, nullptr, g_dec, g_beta, g_rays_o, g_rays_d, ray_order, workspace, stream);
}

extern "C" int eslam_render_bwd_loss(const eslam_plane_t* planes, const eslam_decoders_t* dec, const float* bound6_host,
                                     const float* rays_o, const float* rays_d, const float* z_vals, int R, int S,
                                     const float* sdf, const float* raw_rgb, const float* feat, const float* depth,
                                     const float* rgb, const float* gt_depth, const float* gt_color, double truncation,
                                     const float* weights5_host, const uint8_t* ray_mask, const float* acc,
                                     const float* upstream, float* loss_out, const float* g_depth, const float* g_rgb,
                                     const float* g_sdf, float* g_dec, float* g_beta, float* g_rays_o, float* g_rays_d,
                                     const int32_t* ray_order, void* workspace, eslam_stream_t stream) {
    if (!depth || !rgb || !gt_depth || !gt_color || !weights5_host || !acc) {
        eslam_set_error("eslam_render_bwd_loss: null loss argument");
        return 1;
    }
    LossGradIn li = {};
    li.gt_depth = gt_depth; li.gt_color = gt_color; li.ray_mask = ray_mask; li.depth = depth; li.rgb = rgb;
    li.acc = acc; li.upstream = upstream; li.loss_out = loss_out;
    li.tr = make_trunc(truncation);
    li.w = LossW{weights5_host[0], weights5_host[1], weights5_host[2], weights5_host[3], weights5_host[4]};
    return render_bwd_impl("eslam_render_bwd_loss", planes, dec, bound6_host, rays_o, rays_d, z_vals, R, S, sdf, raw_rgb,
                           feat, g_depth, g_rgb, g_sdf, &li, g_dec, g_beta, g_rays_o, g_rays_d, ray_order, workspace, stream);
}

extern "C" int eslam_decode_bwd(const eslam_plane_t* planes, const eslam_decoders_t* dec, const float* bound6_host,
                                const float* pts, int64_t N, const float* raw, const float* feat, const float* g_raw,
                                float* g_dec, float* g_pts, void* workspace, eslam_stream_t stream) {
    if (N <= 0) return 0;
    if (!planes || !dec || !bound6_host || !pts || !raw || !feat || !g_raw || !workspace) {
        eslam_set_error("eslam_decode_bwd: null argument");
        return 1;
    }
    if (N >= ((int64_t)1 << 31)) {
        eslam_set_error("eslam_decode_bwd: N too large");
        return 1;
    }
    if (eslam_validate_planes(planes, 0, NPL)) return 1;
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    float* g_o = (float*)ws;
    float* g_feat = (float*)(ws + align256(N * 16));
    float* slabs = (float*)(ws + align256(N * 16) + align256(N * 512));
    const Bound bnd = make_bound(bound6_host);
    hipLaunchKernelGGL(decode_act_bwd_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, raw, g_raw, N, g_o);
    if (int rc = eslam_check_launch("decode_act_bwd_kernel")) return rc;
    return bwd_common(planes, dec, bnd, nullptr, nullptr, pts, N, 64, 0, feat, g_o, g_feat, slabs, nullptr, g_dec,
                      g_pts, nullptr, RayBwdIn{}, nullptr, nullptr, st);
}
