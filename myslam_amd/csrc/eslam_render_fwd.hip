// Forward kernels: fused tri-plane gather -> SDF/colour MLPs -> volumetric composite (render), and the
// decoder-only variant on free points (Decoders.forward / get_raw_sdf).
//
// One wave64 = one ray (render) or one tile of 64 points (decode); 4 waves per workgroup share an LDS copy of
// the decoder weights.  See eslam_decode_tile.h for the lane roles and the MFMA operand plan.
#include "eslam_decode_tile.h"
#include "eslam_loss_final.h"

// ---------------------------------------------------------------------------------------------------------
// render: replaces reference src/utils/Renderer.py:136-147 (+ decoders.py:64-146, common.py:204-218)
// ---------------------------------------------------------------------------------------------------------
// LOSS: the kernel also forms the sums of the callers' loss (src/Mapper.py:110-144,337-346) from the values it has in
// registers anyway - per sample the region of z against gt_depth and the squared SDF error, per ray the depth and colour
// errors - reduces them per workgroup and finishes acc [16] and the loss value with the ticket scheme of eslam_loss_value:
// one launch less per iteration (14 us), and sdf / depth / rgb are not read back for it.
struct LossIn {
    const float* gt_depth;
    const float* gt_color;
    const uint8_t* ray_mask;
    float* scratch;
    float* acc;
    float* loss;
    Trunc tr;
    LossW w;
    int det;               // ESLAM_DETERMINISTIC: fixed-order reduction of the workgroups' sums
};

// LOWP: the mixed-precision tile of eslam_decode_tile.h (fp16 plane copies, bf16 MFMA decoders); channels-last only.
template <bool CL, bool SAVE, bool LOSS, bool LOWP>
#ifndef FWD_FEAT_NT
#define FWD_FEAT_NT 1            // A/B switches (make variant VFLAGS=-DFWD_FEAT_NT=0 / -DFWD_CHAIN=0)
#endif
#ifndef FWD_CHAIN
#define FWD_CHAIN 1
#endif
#ifndef FWD_Z_PERMUTE
#define FWD_Z_PERMUTE 1        // a block's z values by lane permute from one load per chunk (0: one dependent load per block)
#endif
#ifndef FWD_BUFSTORE
#define FWD_BUFSTORE 1         // the saved features leave through always-issued buffer stores (countable: store_features_buffer)
#endif
#ifndef FWD_WAVES
#define FWD_WAVES 2            // waves per SIMD the gather kernels are compiled for: with one plane of loads in flight
                               // ahead of the FMAs the forward kernel needs 195 VGPRs; 2 waves/SIMD measured fastest
#endif
__global__ __launch_bounds__(256, FWD_WAVES) void render_fwd_kernel(const PlaneSet planes, const eslam_decoders_t dec,
                                                         const Bound bnd, const float* __restrict__ rays_o,
                                                         const float* __restrict__ rays_d,
                                                         const float* __restrict__ z_vals, int R, int S,
                                                         float* __restrict__ depth_out, float* __restrict__ rgb_out,
                                                         float* __restrict__ sdf_out, float* __restrict__ raw_rgb_out,
                                                         float* __restrict__ feat_out, const int* __restrict__ perm,
                                                         const LossIn li, uint32_t* __restrict__ rng_bump) {
    // the samplers drew this iteration's random numbers from (seed, *rng_bump) before this kernel started
    // (eslam_sample_z_all_rng): advance the step for the next iteration (of a replayed graph)
    if (rng_bump && blockIdx.x == 0 && threadIdx.x == 0) rng_bump[0] += 1u;
    __shared__ __attribute__((aligned(16))) float wlds[2 * DEC_LDS];
    static_assert(LP_LDS_FLOATS <= 2 * DEC_LDS, "the bf16 weight image shares the float32 image's LDS");
    if (LOWP) stage_decoder_weights_lowp(wlds, dec, threadIdx.x, blockDim.x);
    else stage_decoder_weights(wlds, dec, threadIdx.x, blockDim.x);
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int r = lane & 15, q = lane >> 4;                                  // MFMA role
    const int gp = gather_point<CL>(lane), gq = gather_piece<CL>(lane);     // gather role
    // (R * S * 512 < 2^32 - 256 is checked at the launch when features are saved)
    const __amdgpu_buffer_rsrc_t frsrc = __builtin_amdgcn_make_buffer_rsrc((void*)feat_out, 0, (int)((unsigned)R * (unsigned)S * 512u), 0x00020000);
    float lv[A_COUNT];                     // LOSS: this lane's share of the accumulators
#pragma unroll
    for (int k = 0; k < A_COUNT; ++k) lv[k] = 0.0f;
    // Workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share one, MI355X_MICROARCH.md).  Give every
    // XCD a contiguous run of the ray order, so that rays through neighbouring pixels - which read the same texels -
    // are resident on CUs behind the same L2.  Placement only affects speed: the grid has 8*cpx blocks, every logical
    // block is taken exactly once, surplus blocks exit.
    const int cpx = gridDim.x >> 3;
    const int lblock = (blockIdx.x & 7) * cpx + (blockIdx.x >> 3);
    if (!LOSS && lblock * 4 + wave >= R) return;
    // Persistent over the rays when the batch has more workgroups' worth of rays than the chip holds at once (the host caps the
    // grid at FWD_WAVES workgroups per CU): the decoder weights are staged once per resident workgroup instead of once per
    // four rays.  Round k gives every XCD the next contiguous run of the ray order.
    // (LOSS: idle waves still take part in the workgroup's reduction below)
    for (int slot = lblock * 4 + wave; slot < R; slot += (int)gridDim.x * 4) {
    const int ray = perm ? perm[slot] : slot;
    const float gtd = LOSS ? li.gt_depth[ray] : 0.0f;
    const bool in_batch = LOSS ? (li.ray_mask ? li.ray_mask[ray] != 0 : true) : false;
    const bool has_depth = in_batch && gtd > 0.0f;

    const float ox = rays_o[ray * 3 + 0], oy = rays_o[ray * 3 + 1], oz = rays_o[ray * 3 + 2];
    const float dx = rays_d[ray * 3 + 0], dy = rays_d[ray * 3 + 1], dz = rays_d[ray * 3 + 2];
    const float beta = dec.beta[0];
    const float* zrow = z_vals + (int64_t)ray * S;

    float trans_in = 1.0f;                 // transmittance entering the chunk
    float acc_depth = 0.0f, acc_r = 0.0f, acc_g = 0.0f, acc_b = 0.0f;

    for (int c0 = 0; c0 < S; c0 += WAVE) {
        const int nvalid = min(WAVE, S - c0);
        const int nblk = (nvalid + 15) >> 4;

        float4_t out[2];
        // gather role: normalised coordinates of point 16b + gp of the chunk
        // the chunk's z values, sample role (lane l: sample min(c0 + l, S - 1)); a block's points fetch theirs with a lane
        // permute - as a load per block the value sat on the critical path of the NEXT block's first texel request
        const float zs = zrow[min(c0 + lane, S - 1)];
        auto block_point = [&](int b, float& px, float& py, float& pz) {
            const float zb = FWD_Z_PERMUTE ? __shfl(zs, 16 * b + gp, WAVE) : zrow[min(c0 + 16 * b + gp, S - 1)];
            px = norm_coord(ox + dx * zb, bnd.lo[0], bnd.hi[0]);
            py = norm_coord(oy + dy * zb, bnd.lo[1], bnd.hi[1]);
            pz = norm_coord(oz + dz * zb, bnd.lo[2], bnd.hi[2]);
        };
        constexpr bool CHAIN = CL && !LOWP && FWD_CHAIN != 0;    // float32 channels-last planes: texel requests run one block ahead
        PlaneTaps carry;
        float cx = 0.f, cy = 0.f, cz = 0.f;
        if (CHAIN) {
            block_point(0, cx, cy, cz);
            issue_plane0(planes, 0, cx, cy, cz, gq, opaque_zero(c0), carry);
        }
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            out[d] = LOWP ? *(const float4_t*)(lp_biases(wlds, d) + 32) : *(const float4_t*)(wlds + d * DEC_LDS + DEC_B3);
#pragma unroll 1
            for (int b = 0; b < nblk; ++b) {
                const int oz0 = opaque_zero(b);
                const int sb = c0 + 16 * b + gp;
                float feat[16];
                if (CHAIN) {
                    const bool last = b + 1 == nblk;
                    float nx, ny, nz;
                    block_point(last ? 0 : b + 1, nx, ny, nz);
                    gather_features_chain(planes, d, cx, cy, cz, gq, feat, oz0, carry, last ? 1 : d, nx, ny, nz);
                    cx = nx; cy = ny; cz = nz;
                    if (SAVE) {
                        if (FWD_BUFSTORE) {
                            __builtin_amdgcn_sched_barrier(0);      // (left alone, the scheduler sinks the stores behind the MLP)
                            store_features_buffer<FWD_FEAT_NT != 0>(frsrc, sb < S ? (unsigned)(ray * S + sb) * 512u + (unsigned)(d * 256 + gq * 16)
                                                                                  : ESLAM_OOB_OFFSET, feat);
                            __builtin_amdgcn_sched_barrier(0);
                        } else if (sb < S) store_features<FWD_FEAT_NT != 0>(feat_out, (int64_t)ray * S + sb, d, gq, feat);
                    }
                    to_mfma_role<CL, 16>(feat, lane);
                    DecFrag f;
                    load_dec_frag(f, wlds + d * DEC_LDS + oz0, r, q);
                    float4_t h1, h2;
                    mlp_hidden(f, feat, h1, h2);
                    mlp_out_accum(f, h2, b, r, out[d]);
                    continue;
                }
                float px, py, pz;
                block_point(b, px, py, pz);
                if (LOWP) {
                    gather_features_half(planes, d, px, py, pz, gq, feat, oz0);
                    if (SAVE) {
                        if (sb < S) store_features_lp(feat_out, (int64_t)ray * S + sb, d, gq, feat);
                    }
                    to_mfma_role<true, 16>(feat, lane);
                    DecFragLP fl;
                    load_dec_frag_lp(fl, lp_weights(wlds, d) + oz0, lp_biases(wlds, d) + oz0, r, q);
                    float4_t a1, a2;
                    mlp_hidden_lp(fl, feat, a1, a2);
                    mlp_out_accum_lp(fl, a2, b, r, out[d]);
                    continue;
                }
                gather_features<CL>(planes, d, px, py, pz, gq, feat, oz0);
                if (SAVE) {
                    if (sb < S) store_features<FWD_FEAT_NT != 0>(feat_out, (int64_t)ray * S + sb, d, gq, feat);
                }
                to_mfma_role<CL, 16>(feat, lane);
                // operand fragments are re-read from LDS per block (9 ds_read_b128) instead of being kept live across
                // the gather, where they would push the kernel past 128 VGPRs
                DecFrag f;
                load_dec_frag(f, wlds + d * DEC_LDS + oz0, r, q);
                float4_t h1, h2;
                mlp_hidden(f, feat, h1, h2);
                mlp_out_accum(f, h2, b, r, out[d]);
            }
        }

        // ---- sample role: activations, alpha, transmittance scan, composite (Renderer.py:140-153) ----
        const bool valid = lane < nvalid;
        const int s = c0 + lane;
        const float z = valid ? (FWD_Z_PERMUTE ? zs : zrow[s]) : 0.0f;
        const float sdf = tanhf(out[0][0]);
        const float cr = sigmoidf_(out[1][0]), cg = sigmoidf_(out[1][1]), cb = sigmoidf_(out[1][2]);
        if (valid) {
            sdf_out[(int64_t)ray * S + s] = sdf;
            if (SAVE) {
                float* rr = raw_rgb_out + ((int64_t)ray * S + s) * 3;
                rr[0] = cr; rr[1] = cg; rr[2] = cb;
            }
        }
        const float sg = sigmoidf_(-sdf * beta);
        float alpha = 1.0f - expf(-beta * sg);
        if (!valid) alpha = 0.0f;
        const float fac = valid ? (1.0f - alpha) + 1e-10f : 1.0f;
        const float pin = wave_incl_prod(fac, lane);
        const float pex = wave_up1(pin, 1.0f);
        const float w = alpha * (trans_in * pex);
        acc_depth += wave_sum(w * z);
        acc_r += wave_sum(w * cr);
        acc_g += wave_sum(w * cg);
        acc_b += wave_sum(w * cb);
        trans_in *= wave_lane<63>(pin);
        if (LOSS && valid && has_depth) {              // Mapper.py:124-140
            const int reg = sdf_region(z, gtd, li.tr);
            if (reg == 0) { lv[A_N_FRONT] += 1.0f; const float e = sdf - 1.0f; lv[A_S_FRONT] += e * e; }
            else if (reg == 1) { lv[A_N_CENTER] += 1.0f; const float e = (z + sdf * li.tr.t) - gtd; lv[A_S_CENTER] += e * e; }
            else if (reg == 2) { lv[A_N_TAIL] += 1.0f; const float e = (z + sdf * li.tr.t) - gtd; lv[A_S_TAIL] += e * e; }
        }
    }
    if (lane == 0) {
        depth_out[ray] = acc_depth;
        rgb_out[ray * 3 + 0] = acc_r;
        rgb_out[ray * 3 + 1] = acc_g;
        rgb_out[ray * 3 + 2] = acc_b;
        if (LOSS) {                                    // Mapper.py:343,346
            if (has_depth) { const float e = gtd - acc_depth; lv[A_N_DEPTH] += 1.0f; lv[A_S_DEPTH] += e * e; }
            if (in_batch) {
                const float er = li.gt_color[3 * ray] - acc_r, eg = li.gt_color[3 * ray + 1] - acc_g,
                            eb = li.gt_color[3 * ray + 2] - acc_b;
                lv[A_S_COLOR] += (er * er + eg * eg) + eb * eb;
                lv[A_N_COLOR] += 3.0f;
            }
        }
    }
    }
    if (LOSS) {
        __shared__ float red[4][A_COUNT];
#pragma unroll
        for (int k = 0; k < A_COUNT; ++k) {
            const float t = wave_sum(lv[k]);
            if (lane == 0) red[wave][k] = t;
        }
        __syncthreads();
        float tot = 0.0f;
        if (threadIdx.x < A_COUNT)
            tot = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        if (li.det) loss_finalize_det(tot, li.scratch, li.acc, li.w, li.loss);
        else loss_finalize(tot, li.scratch, li.acc, li.w, li.loss);
    }
}

// ---------------------------------------------------------------------------------------------------------
// decode on free points: replaces reference src/networks/decoders.py:127-146 (and :87-105 when SDF_ONLY)
// ---------------------------------------------------------------------------------------------------------
template <bool CL, bool SDF_ONLY, bool SAVE>
__global__ __launch_bounds__(256, FWD_WAVES) void decode_fwd_kernel(const PlaneSet planes, const eslam_decoders_t dec,
                                                         const Bound bnd, const float* __restrict__ pts, int64_t N,
                                                         float* __restrict__ raw, float* __restrict__ feat_out,
                                                         const int mask_outside) {
    __shared__ __attribute__((aligned(16))) float wlds[2 * DEC_LDS];
    stage_decoder_weights(wlds, dec, threadIdx.x, blockDim.x);
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int r = lane & 15, q = lane >> 4;                                  // MFMA role
    const int gp = gather_point<CL>(lane), gq = gather_piece<CL>(lane);     // gather role
    const int64_t ntiles = (N + 63) / 64;

    for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < ntiles; tile += (int64_t)gridDim.x * 4) {
        const int64_t p0 = tile * 64;
        const int nvalid = (int)min((int64_t)64, N - p0);
        const int nblk = (nvalid + 15) >> 4;
        float4_t out[2];
#pragma unroll
        for (int d = 0; d < (SDF_ONLY ? 1 : 2); ++d) {
            out[d] = *(const float4_t*)(wlds + d * DEC_LDS + DEC_B3);
#pragma unroll 1
            for (int b = 0; b < nblk; ++b) {
                const int oz0 = opaque_zero(b);
                const int64_t pb = p0 + 16 * b + gp;
                const int64_t pc = min(pb, N - 1);
                const float px = norm_coord(pts[pc * 3 + 0], bnd.lo[0], bnd.hi[0]);
                const float py = norm_coord(pts[pc * 3 + 1], bnd.lo[1], bnd.hi[1]);
                const float pz = norm_coord(pts[pc * 3 + 2], bnd.lo[2], bnd.hi[2]);
                float feat[16];
                gather_features<CL>(planes, d, px, py, pz, gq, feat, oz0);
                if (SAVE) {
                    if (pb < N) store_features<FWD_FEAT_NT != 0>(feat_out, pb, d, gq, feat);
                }
                to_mfma_role<CL, 16>(feat, lane);
                DecFrag f;
                load_dec_frag(f, wlds + d * DEC_LDS + oz0, r, q);
                float4_t h1, h2;
                mlp_hidden(f, feat, h1, h2);
                mlp_out_accum(f, h2, b, r, out[d]);
            }
        }
        if (lane < nvalid) {
            float sdf = tanhf(out[0][0]);
            if (mask_outside) {     // Mesher.eval_points (Mesher.py:146-153): points not strictly inside the bound get -1
                const float* pp = pts + (p0 + lane) * 3;
                const bool in = pp[0] < bnd.hi[0] && pp[0] > bnd.lo[0] && pp[1] < bnd.hi[1] && pp[1] > bnd.lo[1] &&
                                pp[2] < bnd.hi[2] && pp[2] > bnd.lo[2];
                if (!in) sdf = -1.0f;
            }
            if (SDF_ONLY) {
                raw[p0 + lane] = sdf;
            } else {
                float4_t v;
                v[0] = sigmoidf_(out[1][0]);
                v[1] = sigmoidf_(out[1][1]);
                v[2] = sigmoidf_(out[1][2]);
                v[3] = sdf;
                *(float4_t*)(raw + (p0 + lane) * 4) = v;
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

static int render_fwd_common(const char* who, const eslam_plane_t* planes, const eslam_decoders_t* dec,
                             const float* bound6_host, const float* rays_o, const float* rays_d, const float* z_vals,
                             int R, int S, float* depth, float* rgb, float* sdf, float* raw_rgb, float* feat,
                             const int32_t* ray_order, const LossIn* li, uint32_t* rng_bump, eslam_stream_t stream) {
    if (R <= 0) return 0;
    if (S <= 0 || S > ESLAM_MAX_SAMPLES) {
        eslam_set_error("%s: S=%d outside [1,%d]", who, S, ESLAM_MAX_SAMPLES);
        return 1;
    }
    if (!planes || !dec || !bound6_host || !rays_o || !rays_d || !z_vals || !depth || !rgb || !sdf) {
        eslam_set_error("%s: null argument", who);
        return 1;
    }
    if ((raw_rgb == nullptr) != (feat == nullptr)) {
        eslam_set_error("%s: raw_rgb and feat must both be given or both be NULL", who);
        return 1;
    }
    if (eslam_validate_planes(planes, 0, NPL)) return 1;
    PlaneSet ps;
    for (int i = 0; i < NPL; ++i) ps.p[i] = planes[i];
    const Bound bnd = make_bound(bound6_host);
    const bool cl = eslam_planes_channels_last(planes, 0, NPL);
    const bool save = feat != nullptr;
    if (save && (int64_t)R * S * 512 >= ((int64_t)1 << 32) - 256) {       // saved features are addressed with 32-bit byte offsets
        eslam_set_error("%s: %lld points exceed the 32-bit offset range of the saved-feature buffer (8.3 M): split the batch", who,
                        (long long)R * S);
        return 1;
    }
    const int nblocks = (R + 3) / 4;
    static const int resident = [] {        // workgroups the chip holds at once: FWD_WAVES waves per SIMD = FWD_WAVES workgroups per CU
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        static const char* e = getenv("ESLAM_FWD_PERSIST");
        return (e && e[0] == '0') ? (1 << 30) : ((FWD_WAVES * cus) / 8) * 8;
    }();
    // persistent only when the rays divide evenly over the resident workgroups (4096, 8192 rays ...): a static assignment of
    // 2.44 rays per wave (5000 rays) runs three rounds where the hardware's own dispatch of 1250 workgroups needs 2.5
    // (forward 95 -> 108 us at 5000 x 56)
    int nwg = ((nblocks + 7) / 8) * 8;
    if (nwg > resident && nwg % resident == 0) nwg = resident;
    dim3 grid(nwg), block(256);
    hipStream_t st = (hipStream_t)stream;
    const LossIn none = {};
#define LAUNCH(CLv, SV, LS)                                                                                             \
    hipLaunchKernelGGL((render_fwd_kernel<CLv, SV, LS, false>), grid, block, 0, st, ps, *dec, bnd, rays_o, rays_d, z_vals, R, \
                       S, depth, rgb, sdf, raw_rgb, feat, (const int*)ray_order, LS ? *li : none, rng_bump)
#define LAUNCH_LP(SV, LS)                                                                                               \
    hipLaunchKernelGGL((render_fwd_kernel<true, SV, LS, true>), grid, block, 0, st, ps, *dec, bnd, rays_o, rays_d, z_vals, R, \
                       S, depth, rgb, sdf, raw_rgb, feat, (const int*)ray_order, LS ? *li : none, rng_bump)
    eslam_prof_begin(PROF_RENDER_FWD, st);
    const int lowp = eslam_planes_lowp(planes);
    if (lowp < 0) return 1;
    if (lowp) {
        if (li) { if (save) LAUNCH_LP(true, true); else LAUNCH_LP(false, true); }
        else { if (save) LAUNCH_LP(true, false); else LAUNCH_LP(false, false); }
    } else if (li) {
        if (cl && save) LAUNCH(true, true, true);
        else if (cl) LAUNCH(true, false, true);
        else if (save) LAUNCH(false, true, true);
        else LAUNCH(false, false, true);
    } else {
        if (cl && save) LAUNCH(true, true, false);
        else if (cl) LAUNCH(true, false, false);
        else if (save) LAUNCH(false, true, false);
        else LAUNCH(false, false, false);
    }
#undef LAUNCH
#undef LAUNCH_LP
    eslam_prof_end(PROF_RENDER_FWD, st);
    return eslam_check_launch("render_fwd_kernel");
}

extern "C" int eslam_render_fwd(const eslam_plane_t* planes, const eslam_decoders_t* dec, const float* bound6_host,
                                const float* rays_o, const float* rays_d, const float* z_vals, int R, int S,
                                float* depth, float* rgb, float* sdf, float* raw_rgb, float* feat,
                                const int32_t* ray_order, uint32_t* rng_bump, eslam_stream_t stream) {
    return render_fwd_common("eslam_render_fwd", planes, dec, bound6_host, rays_o, rays_d, z_vals, R, S, depth, rgb, sdf,
                             raw_rgb, feat, ray_order, nullptr, rng_bump, stream);
}

extern "C" int eslam_render_fwd_loss(const eslam_plane_t* planes, const eslam_decoders_t* dec, const float* bound6_host,
                                     const float* rays_o, const float* rays_d, const float* z_vals, int R, int S,
                                     float* depth, float* rgb, float* sdf, float* raw_rgb, float* feat,
                                     const int32_t* ray_order, const float* gt_depth, const float* gt_color,
                                     double truncation, const float* weights5_host, const uint8_t* ray_mask,
                                     float* scratch, float* acc, float* loss, uint32_t* rng_bump, eslam_stream_t stream) {
    if (!gt_depth || !gt_color || !weights5_host || !scratch || !acc) {
        eslam_set_error("eslam_render_fwd_loss: null loss argument");
        return 1;
    }
    if (R <= 0) {
        eslam_set_error("eslam_render_fwd_loss: empty batch");
        return 1;
    }
    LossIn li;
    li.gt_depth = gt_depth; li.gt_color = gt_color; li.ray_mask = ray_mask;
    li.scratch = scratch; li.acc = acc; li.loss = loss;
    li.det = eslam_deterministic();
    li.tr = make_trunc(truncation);
    li.w = LossW{weights5_host[0], weights5_host[1], weights5_host[2], weights5_host[3], weights5_host[4]};
    return render_fwd_common("eslam_render_fwd_loss", planes, dec, bound6_host, rays_o, rays_d, z_vals, R, S, depth, rgb,
                             sdf, raw_rgb, feat, ray_order, &li, rng_bump, stream);
}

extern "C" int eslam_decode_fwd(const eslam_plane_t* planes, const eslam_decoders_t* dec, const float* bound6_host,
                                const float* pts, int64_t N, int flags, float* raw, float* feat,
                                eslam_stream_t stream) {
    if (N <= 0) return 0;
    if (flags & ~(ESLAM_DECODE_SDF_ONLY | ESLAM_DECODE_MASK_OUTSIDE)) {
        eslam_set_error("eslam_decode_fwd: unknown flags 0x%x", flags);
        return 1;
    }
    const int sdf_only = flags & ESLAM_DECODE_SDF_ONLY;
    const int mask_outside = (flags & ESLAM_DECODE_MASK_OUTSIDE) ? 1 : 0;
    if (!planes || !dec || !bound6_host || !pts || !raw) {
        eslam_set_error("eslam_decode_fwd: null argument");
        return 1;
    }
    if (sdf_only && feat) {
        eslam_set_error("eslam_decode_fwd: feat cannot be saved in sdf_only mode");
        return 1;
    }
    if (eslam_validate_planes(planes, 0, sdf_only ? 6 : NPL)) return 1;
    PlaneSet ps;
    for (int i = 0; i < NPL; ++i) ps.p[i] = planes[sdf_only && i >= 6 ? i - 6 : i];
    const Bound bnd = make_bound(bound6_host);
    const bool cl = eslam_planes_channels_last(planes, 0, sdf_only ? 6 : NPL);
    const int64_t ntiles = (N + 63) / 64;
    const int64_t nwg = (ntiles + 3) / 4;
    dim3 grid((unsigned)(nwg < 8192 ? nwg : 8192)), block(256);
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH(CLv, SO, SV) \
    hipLaunchKernelGGL((decode_fwd_kernel<CLv, SO, SV>), grid, block, 0, st, ps, *dec, bnd, pts, N, raw, feat, \
                       mask_outside)
    eslam_prof_begin(PROF_DECODE_FWD, st);
    if (sdf_only) {
        if (cl) LAUNCH(true, true, false);
        else LAUNCH(false, true, false);
    } else if (feat) {
        if (cl) LAUNCH(true, false, true);
        else LAUNCH(false, false, true);
    } else {
        if (cl) LAUNCH(true, false, false);
        else LAUNCH(false, false, false);
    }
#undef LAUNCH
    eslam_prof_end(PROF_DECODE_FWD, st);
    return eslam_check_launch("decode_fwd_kernel");
}
