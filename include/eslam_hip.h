/*
 * eslam_hip.h - C ABI of the MI355X (gfx950) implementation of ESLAM's per-iteration rendering hot path.
 *
 * The reference (MohammadJohari/myslam) is 100 % Python on PyTorch and has no FFI of its own; its boundary for
 * this path is the in-process Python API  Renderer.render_batch_ray / get_samples / Decoders.forward.
 * This header is the C-ABI a binding for that API calls (myslam_amd/_hip.py is the ctypes binding we ship,
 * INTEGRATION.md shows the stub a reference maintainer would add).  Each entry point names the reference
 * lines it replaces (paths relative to the reference repository root).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to float32 / int64 data unless the name ends in _host;
 *   - `stream` is a hipStream_t (NULL = default stream); every call only enqueues work on it: no allocation,
 *     no synchronisation, no host<->device copy, so a caller may capture a call into a hipGraph;
 *   - return value 0 = ok, anything else = error; eslam_last_error() gives the message (thread local);
 *   - tri-planes are the reference's 6 lists x 2 levels flattened in `all_planes` order:
 *       index = 2*g + level,  g in (planes_xy, planes_xz, planes_yz, c_planes_xy, c_planes_xz, c_planes_yz),
 *     each a logical [1, C=32, h, w] tensor described by element strides, so both NCHW-contiguous tensors
 *     (what reference src/ESLAM.py:201-210 allocates) and channels-last ones (what myslam_amd.scene allocates;
 *     one texel = 128 contiguous bytes, the fast path) are accepted without a copy;
 *   - decoder parameters are the reference's tensors (src/networks/decoders.py:47-60), row-major [out, in].
 */
#ifndef ESLAM_HIP_H
#define ESLAM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ESLAM_ABI_VERSION 5
#define ESLAM_C_DIM 32          /* feature channels per plane (configs/ESLAM.yaml:77)               */
#define ESLAM_HIDDEN 16         /* decoder hidden width (src/networks/decoders.py:39)               */
#define ESLAM_FEAT (2 * ESLAM_C_DIM)   /* coarse || fine                                          */
#define ESLAM_N_PLANES 12
#define ESLAM_MAX_SAMPLES 256   /* samples per ray supported by the per-ray kernels                  */
#define ESLAM_RAY_ORDERS 3      /* eslam_ray_order writes one order per plane orientation (xy, xz, yz)    */
#define ESLAM_RAY_ORDER_WORDS(R) (ESLAM_RAY_ORDERS * (int64_t)(R) + 4)   /* int32 words of its output buffer */
/* floats in the flat decoder-gradient vector, in the order of eslam_decoders_t (beta excluded)      */
#define ESLAM_N_DEC_PARAMS (2 * (16 * 64 + 16 + 16 * 16 + 16) + (1 * 16 + 1) + (3 * 16 + 3))

typedef void* eslam_stream_t;

typedef struct {
    const float* data;      /* element [0,0,0,0]                                                  */
    float* grad;            /* same strides as data; kernels ACCUMULATE (+=) into it; may be NULL  */
    int32_t h, w;
    int64_t stride_c, stride_y, stride_x;   /* in elements                                         */
    const void* data_f16;   /* optional IEEE-half copy of the plane, channels-last with the SAME element strides
                               (stride_c = 1, stride_x = 32: 64-byte texels).  When all 12 planes carry one, eslam_render_fwd*,
                               eslam_render_bwd* run the mixed-precision path of BASELINE.json configs[4]: texels gathered from
                               the half copies (float32 accumulation), decoders on bf16 MFMA forward AND backward, plane
                               gradients accumulated in float32 into `grad` (the float32 master's gradient).  The saved
                               features `feat` then hold R*S*128 bf16 values (half the bytes of the float32 path's buffer). */
} eslam_plane_t;

typedef struct {            /* src/networks/decoders.py:47-60                                      */
    const float* w1;  const float* b1;      /* linears.0        [16,64],[16]                        */
    const float* w2;  const float* b2;      /* linears.1        [16,16],[16]                        */
    const float* w3;  const float* b3;      /* output_linear    [1,16],[1]                          */
    const float* cw1; const float* cb1;     /* c_linears.0      [16,64],[16]                        */
    const float* cw2; const float* cb2;     /* c_linears.1      [16,16],[16]                        */
    const float* cw3; const float* cb3;     /* c_output_linear  [3,16],[3]                          */
    const float* beta;                      /* [1] (a device copy of the int when not learnable)    */
} eslam_decoders_t;

const char* eslam_last_error(void);
int eslam_abi_version(void);

/* K1 - pixel pick + back-projection.  Replaces src/common.py:87-153 (get_samples and helpers) with the
 * torch.randint draw of common.py:108 lifted to the caller: indices[b*n] are flat positions inside the crop
 * window [H0,H1) x [W0,W1), row-major, image k owning indices[k*n .. (k+1)*n).
 * depths [b,H,W], colors [b,H,W,3] contiguous; c2ws [b,4,4] row-major.
 * Outputs rays_o/rays_d [b*n,3], depth [b*n], color [b*n,3].                                        */
int eslam_sample_rays(const int64_t* indices, int b, int n, int H0, int H1, int W0, int W1, int H, int W,
                      float fx, float fy, float cx, float cy, const float* c2ws, const float* depths,
                      const float* colors, float* rays_o, float* rays_d, float* depth, float* color,
                      eslam_stream_t stream);

/* Backward of K1 w.r.t. the camera matrices (autograd of common.py:92-97): g_c2ws [b,4,4] is OVERWRITTEN
 * with  d/dc2w ( <g_rays_o, rays_o> + <g_rays_d, rays_d> );  rows/cols outside [:3,:4] are zero.     */
int eslam_sample_rays_bwd(const int64_t* indices, int b, int n, int H0, int W0, int W1, float fx, float fy,
                          float cx, float cy, const float* g_rays_o, const float* g_rays_d, float* g_c2ws,
                          eslam_stream_t stream);

/* Whole-image rays.  Replaces src/common.py:183-201 (get_rays).  rays_o/rays_d [H*W,3].             */
int eslam_image_rays(int H, int W, float fx, float fy, float cx, float cy, const float* c2w, float* rays_o,
                     float* rays_d, eslam_stream_t stream);

/* a2 - AABB exit distance  min_axis max_side (bound - o)/d.  Replaces the caller-side pre-filter arithmetic
 * of src/Mapper.py:322-328 and src/Tracker.py:175-181.  t_exit [R].                                  */
int eslam_aabb_exit(const float* rays_o, const float* rays_d, int R, const float* bound6_host, float* t_exit,
                    eslam_stream_t stream);

/* K3 - depth-guided sampler.  Replaces src/utils/Renderer.py:85-105 and perturbation (:46-61) for rays with
 * gt_depth > 0; rows of rays with gt_depth <= 0 are left untouched (K4 fills them).
 * t_free[n_strat], t_surf[n_imp]: the two linspace(0,1,.) vectors (device); t_rand [R,S] uniform numbers or
 * NULL for perturb = False.  z_vals [R,S] out, S = n_strat + n_imp.  Bit-exact vs the reference arithmetic
 * (truncation is a double because Renderer.py:97 forms 1.5*truncation and 3*truncation as Python floats).    */
int eslam_sample_z(const float* gt_depth, int R, int n_strat, int n_imp, double truncation, const float* t_free,
                   const float* t_surf, const float* t_rand, float* z_vals, eslam_stream_t stream);

/* K4 - importance sampler for rays with gt_depth <= 0.  Replaces src/utils/Renderer.py:108-134 and
 * src/common.py:41-77 (sample_pdf, det=False).  Only geometry planes + SDF decoder are evaluated, no grad.
 * t_rand_uni [R,n_strat] or NULL, u [R,n_imp]; rows of rays with gt_depth > 0 are ignored / left untouched. */
int eslam_importance_z(const eslam_plane_t* planes, const eslam_decoders_t* dec, const float* bound6_host,
                       const float* rays_o, const float* rays_d, const float* gt_depth, int R, int n_strat,
                       int n_imp, const float* t_free, const float* t_rand_uni, const float* u, float* z_vals,
                       eslam_stream_t stream);

/* eslam_sample_z + eslam_importance_z in ONE launch: every row of z_vals [R, n_strat+n_imp] - rays with gt_depth > 0 by
 * the depth-guided rule (bit-exact, as eslam_sample_z), the others by the importance sampler.  Needs n_strat >= 3.    */
int eslam_sample_z_all(const eslam_plane_t* planes, const eslam_decoders_t* dec, const float* bound6_host,
                       const float* rays_o, const float* rays_d, const float* gt_depth, int R, int n_strat, int n_imp,
                       double truncation, const float* t_free, const float* t_surf, const float* t_rand,
                       const float* t_rand_uni, const float* u, float* z_vals, eslam_stream_t stream);

/* eslam_sample_z_all with the random numbers drawn INSIDE the kernel instead of read from t_rand / t_rand_uni / u (the
 * reference draws them with torch.rand, Renderer.py:59 and common.py:59; a caller that needs the reference's exact
 * stream injects it through eslam_sample_z_all): U = hash(seed, step, stream, element) / 2^24 in [0,1), with
 * step = *rng_state (device memory; NULL = 0).  perturb = 0 leaves the samples un-jittered (Renderer.perturb False);
 * the importance draw is random either way.  Pass the same rng_state as `rng_bump` to the eslam_render_fwd* call that
 * consumes z_vals: it advances the step, so a replayed hipGraph draws fresh numbers without an extra launch.
 * ray_offset: index of this call's ray 0 in the iteration's whole batch (0 for an unsharded call).  The numbers are keyed
 * on the GLOBAL ray index ray_offset + i, so the ranks of a ray-sharded iteration - same seed, same step - draw exactly what
 * the unsharded batch draws: its z_vals, and with them the loss's set sizes, do not depend on the world size.            */
int eslam_sample_z_all_rng(const eslam_plane_t* planes, const eslam_decoders_t* dec, const float* bound6_host,
                           const float* rays_o, const float* rays_d, const float* gt_depth, int R, int n_strat, int n_imp,
                           double truncation, const float* t_free, const float* t_surf, int perturb, uint64_t seed,
                           const uint32_t* rng_state, int64_t ray_offset, float* z_vals, eslam_stream_t stream);

/* K5-K7 forward.  Replaces src/utils/Renderer.py:136-147 + src/networks/decoders.py:64-146 +
 * src/common.py:204-218:  pts = o + d z -> normalise -> tri-plane bilinear gather (border, align_corners) ->
 * SDF / colour MLPs -> sdf2alpha -> transmittance scan -> composite.
 * Outputs: depth [R], rgb [R,3], sdf [R,S].  For a later backward pass also raw_rgb [R,S,3] (sigmoid outputs)
 * and feat [R*S,128] (geometry 64 || colour 64 features per sample); both may be NULL for inference.  With feat given,
 * R*S is limited to 8 388 607 points (rows of feat, and of the backward pass's feature-gradient buffer, are addressed with
 * 32-bit byte offsets: R*S*512 < 2^32 - 256); larger batches return an error - split them (inference has no such limit).
 * ray_order [ESLAM_RAY_ORDERS][R] (optional, from eslam_ray_order; the forward uses the first): the kernel walks the rays in that order with an XCD-contiguous
 * block mapping; outputs stay in the caller's ray order.  NULL = rays are processed as given, which measured FASTER
 * on MI355X (119 vs 123-125 us at 4096 x 64: neighbouring rays in flight together hit the same L2 channels), so the
 * shipped binding passes NULL here and hands the order to eslam_render_bwd only, whose scatter needs it.
 * rng_bump (optional): a device counter this launch increments by one (see eslam_sample_z_all_rng).            */
int eslam_render_fwd(const eslam_plane_t* planes, const eslam_decoders_t* dec, const float* bound6_host,
                     const float* rays_o, const float* rays_d, const float* z_vals, int R, int S, float* depth,
                     float* rgb, float* sdf, float* raw_rgb, float* feat, const int32_t* ray_order,
                     uint32_t* rng_bump, eslam_stream_t stream);

/* eslam_render_fwd that also forms the sums of the callers' mapping loss (src/Mapper.py:110-144,337-346) in its
 * epilogue, from the depth / rgb / sdf it holds in registers: acc [ESLAM_LOSS_ACC] and loss [1] (may be NULL) exactly as
 * eslam_loss_value produces them, scratch as there, ray_mask as there (optional).  eslam_loss_grad(acc) then gives the
 * upstream gradients for eslam_render_bwd.  Not for the tracker's loss: its outlier mask depends on the rendered depth. */
int eslam_render_fwd_loss(const eslam_plane_t* planes, const eslam_decoders_t* dec, const float* bound6_host,
                          const float* rays_o, const float* rays_d, const float* z_vals, int R, int S, float* depth,
                          float* rgb, float* sdf, float* raw_rgb, float* feat, const int32_t* ray_order,
                          const float* gt_depth, const float* gt_color, double truncation, const float* weights5_host,
                          const uint8_t* ray_mask, float* scratch, float* acc, float* loss, uint32_t* rng_bump,
                          eslam_stream_t stream);

/* Refresh the half copies of all 12 planes from their float32 masters (both channels-last), one launch: what a
 * mixed-precision training loop runs after every optimiser step.                                            */
int eslam_planes_to_half(const eslam_plane_t* planes, eslam_stream_t stream);

/* Layout change of all 12 planes in one launch, for callers that keep the reference's own NCHW-contiguous planes
 * (src/ESLAM.py:199-210: a texel's 32 channels lie h*w floats apart): field 0 copies src[i].data into the memory
 * dst[i].data points at, field 1 copies src[i].grad into dst[i].grad.  One side of every plane must be dense
 * NCHW-contiguous, the other dense channels-last (stride_c 1, stride_x 32, stride_y 32 w), the same way for all 12.
 * The shipped binding runs it on the planes in front of the kernels and on the gradients behind them whenever the caller's
 * planes are not channels-last and the batch is large enough to pay for 2 x (plane bytes) of HBM traffic; nothing is kept
 * across calls (the mapper swaps the plane Parameters every frame, src/Mapper.py:254-266).                          */
int eslam_planes_relayout(const eslam_plane_t* src, const eslam_plane_t* dst, int field, eslam_stream_t stream);

/* Mixed-precision forward for inference (BASELINE.json configs[4], a tolerance study): planes_f16[i].data points to
 * IEEE-half data of a channels-last [1,32,h,w] plane (strides in half elements: stride_c = 1, stride_x = 32), the decoder
 * weights are rounded to bf16 inside the kernel and run on bf16 MFMA with float32 accumulation; everything after the MLPs
 * (activations, alpha, transmittance, composite) is float32.  Same outputs as eslam_render_fwd.  (Round-1 entry point:
 * the general way is a `data_f16` pointer in every eslam_plane_t, which switches eslam_render_fwd / _fwd_loss / _bwd /
 * _bwd_loss to the mixed-precision kernels, backward included.)                                                   */
int eslam_render_fwd_lowp(const eslam_plane_t* planes_f16, const eslam_decoders_t* dec, const float* bound6_host,
                          const float* rays_o, const float* rays_d, const float* z_vals, int R, int S, float* depth,
                          float* rgb, float* sdf, eslam_stream_t stream);

/* Ray orders for eslam_render_bwd's plane-gradient scatter (and, optionally, eslam_render_fwd): perm, a buffer of
 * ESLAM_RAY_ORDER_WORDS(R) int32 words, <- [ESLAM_RAY_ORDERS][R] permutations followed by 3 floats (+ 1 pad): the angular
 * extent of the batch's fan of rays in each plane, which the scatter reads to choose its workgroup order.  Three
 * permutations of the ray ids, one per plane orientation (xy, xz, yz).  Rays that share one origin (one camera's
 * batch) are sorted, for orientation o, by the azimuth of their direction projected into that plane - the rays of a bundle
 * then cover a thin wedge of the plane and share its cells; batches with several origins get the same order three times
 * (a Morton key of the point one metre along each ray).  Single-pass counting sorts, chunks of 8192 rays.  Depends only on the
 * rays, so a caller can run it on a side stream next to the samplers.  (ABI 5: perm was [R].)                           */
int eslam_ray_order(const float* rays_o, const float* rays_d, int R, int32_t* perm, eslam_stream_t stream);

/* Bytes of scratch eslam_render_bwd / eslam_decode_bwd need for n_points = R*S points.               */
int64_t eslam_bwd_workspace_bytes(int64_t n_points);

/* K8 backward of eslam_render_fwd (autograd of the lines above).
 * Upstream: g_depth [R], g_rgb [R,3], g_sdf [R,S] (any may be NULL = zero).
 * Accumulates into planes[i].grad (where non-NULL), OVERWRITES g_dec [ESLAM_N_DEC_PARAMS] (order of
 * eslam_decoders_t: w1,b1,w2,b2,w3,b3,cw1,...,cb3; NULL = decoders frozen, their gradient work is skipped) and
 * g_beta [1] (may be NULL), and when g_rays_o / g_rays_d are
 * non-NULL overwrites them ([R,3] each) with the gradient through pts = o + d z.  z_vals carries no gradient
 * (Renderer.py builds it under no_grad / from gt_depth).  ray_order: the buffer eslam_ray_order filled, or NULL
 * (then the order is computed here).
 * workspace: eslam_bwd_workspace_bytes(R*S) bytes.                                                          */
int eslam_render_bwd(const eslam_plane_t* planes, const eslam_decoders_t* dec, const float* bound6_host,
                     const float* rays_o, const float* rays_d, const float* z_vals, int R, int S,
                     const float* sdf, const float* raw_rgb, const float* feat, const float* g_depth,
                     const float* g_rgb, const float* g_sdf, float* g_dec, float* g_beta, float* g_rays_o,
                     float* g_rays_d, const int32_t* ray_order, void* workspace, eslam_stream_t stream);

/* eslam_render_bwd for an iteration whose loss is the mapping loss (src/Mapper.py:110-144,337-346; with ray_mask the
 * tracker's, src/Tracker.py:114-148,197-204): the upstream gradients d loss / d (depth, rgb, sdf) are formed INSIDE the
 * backward kernel from acc [ESLAM_LOSS_ACC] - the set sizes and sums eslam_render_fwd_loss / eslam_loss_value /
 * eslam_loss_reduce (+ all-reduce) produced - instead of being written by eslam_loss_grad and read back: loss gradient,
 * composite backward and decoder backward are one launch.  depth [R], rgb [R,3]: the forward outputs.  upstream [1] on
 * the device = d L / d loss (NULL = 1).  loss_out [1] (optional): receives the loss value formed from acc (a ray-sharded
 * caller's acc is only complete after its all-reduce).  g_depth / g_rgb / g_sdf (each optional): FURTHER upstream
 * gradients on the rendered outputs, added to the loss's own.  Everything else as eslam_render_bwd.                */
int eslam_render_bwd_loss(const eslam_plane_t* planes, const eslam_decoders_t* dec, const float* bound6_host,
                          const float* rays_o, const float* rays_d, const float* z_vals, int R, int S,
                          const float* sdf, const float* raw_rgb, const float* feat, const float* depth,
                          const float* rgb, const float* gt_depth, const float* gt_color, double truncation,
                          const float* weights5_host, const uint8_t* ray_mask, const float* acc,
                          const float* upstream, float* loss_out, const float* g_depth, const float* g_rgb,
                          const float* g_sdf, float* g_dec, float* g_beta, float* g_rays_o, float* g_rays_d,
                          const int32_t* ray_order, void* workspace, eslam_stream_t stream);

/* Decoder-only query.  Replaces src/networks/decoders.py:127-146 (Decoders.forward), the entry used by
 * src/utils/Mesher.py:151 on up to 500k points.  pts [N,3] world coordinates -> raw [N,4] = (r,g,b,sdf).
 * flags: ESLAM_DECODE_SDF_ONLY evaluates geometry planes + SDF decoder only (decoders.py:87-105) and writes
 * raw [N,1]; ESLAM_DECODE_MASK_OUTSIDE sets the sdf of every point that is not strictly inside the bound to -1
 * (Mesher.eval_points, src/utils/Mesher.py:146-153).  feat [N,128] optional (needed only for eslam_decode_bwd). */
#define ESLAM_DECODE_SDF_ONLY 1
#define ESLAM_DECODE_MASK_OUTSIDE 2
int eslam_decode_fwd(const eslam_plane_t* planes, const eslam_decoders_t* dec, const float* bound6_host,
                     const float* pts, int64_t N, int flags, float* raw, float* feat, eslam_stream_t stream);

/* Backward of eslam_decode_fwd: g_raw [N,4] upstream, raw [N,4] the forward output.  Same gradient outputs as
 * eslam_render_bwd, with g_pts [N,3] (may be NULL) instead of ray gradients.                          */
int eslam_decode_bwd(const eslam_plane_t* planes, const eslam_decoders_t* dec, const float* bound6_host,
                     const float* pts, int64_t N, const float* raw, const float* feat, const float* g_raw,
                     float* g_dec, float* g_pts, void* workspace, eslam_stream_t stream);

/* Caller-side loss of one optimisation iteration, value and upstream gradients in two small launches and
 * without the boolean-mask host syncs of the PyTorch formulation.  Replaces src/Mapper.py:110-144 (sdf_losses)
 * + :337-346, and src/Tracker.py:114-148 + :197-204 when ray_mask is given.
 *   ray_mask == NULL: every ray belongs to the batch.
 *   ray_mask != NULL: uint8 [R], the rays that belong to the batch - the 10x-median outlier mask of
 *                     Tracker.py:193-195, and/or the AABB pre-filter of Mapper.py:322-332 / Tracker.py:175-187 kept
 *                     as a mask instead of a compaction (no host sync, static shapes for hipGraph capture).
 *   Colour term: mean over the batch's rays; SDF and depth terms: mean over the batch's rays with gt_depth > 0
 *   (in tracking the pre-filter already requires depth, so all three run over the same rays, as the reference's do).
 * weights5_host = {w_sdf_fs, w_sdf_center, w_sdf_tail, w_depth, w_color}  (configs/ESLAM.yaml:29-33,53-57).
 * Outputs: loss [1]; g_depth [R], g_rgb [R,3], g_sdf [R,S] = d loss / d (depth, rgb, sdf) (overwritten).
 * Means over empty sets give NaN exactly as torch.mean does.  scratch: 64 bytes, any contents.            */
int eslam_mapping_loss(const float* depth, const float* rgb, const float* sdf, const float* z_vals,
                       const float* gt_depth, const float* gt_color, int R, int S, double truncation,
                       const float* weights5_host, const uint8_t* ray_mask, float* loss,
                       float* g_depth, float* g_rgb, float* g_sdf, void* scratch, eslam_stream_t stream);

/* The two phases of eslam_mapping_loss separately, for ray-sharded data parallelism: every rank runs
 * eslam_loss_reduce on its shard, the ESLAM_LOSS_ACC floats of `acc` (set sizes and squared-error sums) are
 * summed over ranks with one tiny all-reduce, then eslam_loss_grad produces upstream gradients scaled by the GLOBAL
 * set sizes, so the summed gradients equal those of the unsharded batch.  acc must be zeroed by the caller
 * before eslam_loss_reduce (it accumulates).  eslam_loss_grad: loss, g_* may be NULL when not wanted; `upstream`
 * (device scalar, optional) multiplies the gradients - the grad_output autograd hands to the loss's backward.    */
#define ESLAM_LOSS_ACC 16
int eslam_loss_reduce(const float* depth, const float* rgb, const float* sdf, const float* z_vals,
                      const float* gt_depth, const float* gt_color, int R, int S, double truncation,
                      const uint8_t* ray_mask, float* acc, eslam_stream_t stream);
int eslam_loss_grad(const float* depth, const float* rgb, const float* sdf, const float* z_vals,
                    const float* gt_depth, const float* gt_color, int R, int S, double truncation,
                    const float* weights5_host, const uint8_t* ray_mask, const float* acc, float* loss,
                    float* g_depth, float* g_rgb, float* g_sdf, const float* upstream, eslam_stream_t stream);

/* Optimiser step of the callers' loops: torch.optim.Adam exactly as the reference
 * builds it - default betas (0.9, 0.999) and eps 1e-8 unless given, no weight decay, no amsgrad - over all
 * parameter tensors of one step in ONE launch.  Replaces `optimizer.step()` (+ optionally `optimizer.zero_grad()`)
 * of src/Mapper.py:291-303,348-350 (decoders / planes / c_planes / camera-pose groups, one lr each) and of
 * src/Tracker.py:262-266,206-208 (cam_pose translation / rotation groups).
 *   tensors_host: HOST array of n_tensors descriptors; each names four dense device arrays of n float32 in the
 *                 same element order (param, grad, exp_avg, exp_avg_sq) and the lr of the tensor's param group;
 *   step:         1-based step count t of the bias corrections (state['step'] after the increment), used when
 *                 step_dev is NULL;
 *   step_dev:     optional device int32 counter: incremented by one on the stream and then used as t, so that a
 *                 captured hipGraph can be replayed without re-recording (t never appears in the kernel arguments);
 *   zero_grad:    non-zero = clear every consumed gradient in the same pass.
 * Elements whose grad, exp_avg and exp_avg_sq are all zero are skipped - the dense update leaves them unchanged
 * bit for bit.  Tensors whose four pointers are 16-byte aligned take the float4 path.                          */
#define ESLAM_ADAM_MAX_TENSORS 32   /* descriptors per launch; longer tables are split into several launches */
typedef struct {
    float* param;
    float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    int64_t n;
    double lr;
} eslam_adam_tensor_t;
int eslam_adam_step(const eslam_adam_tensor_t* tensors_host, int n_tensors, int step, int32_t* step_dev,
                    double beta1, double beta2, double eps, int zero_grad, eslam_stream_t stream);

/* Keyframe selection by view overlap, the projection test of src/Mapper.py:170-201 for all keyframes in one launch:
 * the current frame's rays (get_samples, Mapper.py:166-168) with depth > 0 are sampled at num_samples depths between
 * 0.8 d and d + 0.5, projected into each keyframe (w2c = inverse(c2ws[k]), x flipped, pinhole K) and counted when they
 * fall inside the image less `edge` pixels with negative camera z.
 *   c2ws [n_keyframes,4,4] row-major camera-to-world (the caller drops the last two keyframes, Mapper.py:183);
 *   counts int32 [n_keyframes + 1]: counts[k] = points inside keyframe k; counts[n_keyframes] = rays with depth > 0,
 *   so percent_inside[k] = counts[k] / (counts[n_keyframes] * num_samples)  (Mapper.py:201).                     */
int eslam_keyframe_overlap(const float* rays_o, const float* rays_d, const float* gt_depth, int n_rays,
                           int num_samples, const float* c2ws, int n_keyframes, int H, int W, float fx, float fy,
                           float cx, float cy, int edge, int32_t* counts, eslam_stream_t stream);

/* Caller-side glue of the optimisation loops, one launch each instead of a chain of small tensor ops.
 *
 * eslam_prefilter: keep[i] = (AABB exit distance of ray i >= gt_depth[i]) [&& gt_depth[i] > 0 when need_depth]
 *   - the pre-filter of src/Mapper.py:322-328 (need_depth = 0) and src/Tracker.py:175-182 (need_depth = 1) as a uint8
 *   mask for the `ray_mask` of the loss entry points, instead of a boolean-index compaction.
 * eslam_pose_to_c2w(_bwd): src/common.py:169-181 cam_pose_to_matrix - poses [b,7] = (quaternion real-first, translation)
 *   -> c2ws [b,4,4] with R = quaternion_to_matrix(q) (q need not be normalised: two_s = 2 / |q|^2) - and its
 *   backward g_c2ws [b,4,4] -> g_poses [b,7].
 * eslam_tracking_mask: src/Tracker.py:192-195.  mask[i] = keep[i] && |gt_depth[i] - depth[i]| < factor * median, the
 *   (lower, as torch.median) median taken over the rays with keep[i] != 0 (all rays when keep is NULL); R <=
 *   ESLAM_TRACKING_MASK_MAX.  A NaN error makes the median NaN and the mask empty, as in the reference.
 * eslam_keep_best: src/Tracker.py:304-307.  if (loss[0] < best[0]) { best[0] = loss[0]; best_pose[0..n) = pose[0..n); } */
#define ESLAM_TRACKING_MASK_MAX 8192
int eslam_prefilter(const float* rays_o, const float* rays_d, const float* gt_depth, int R, const float* bound6_host,
                    int need_depth, uint8_t* keep, eslam_stream_t stream);
int eslam_pose_to_c2w(const float* poses, int b, float* c2ws, eslam_stream_t stream);
int eslam_pose_to_c2w_bwd(const float* poses, const float* g_c2ws, int b, float* g_poses, eslam_stream_t stream);
int eslam_tracking_mask(const float* depth, const float* gt_depth, const uint8_t* keep, int R, float factor,
                        uint8_t* mask, eslam_stream_t stream);
int eslam_keep_best(const float* loss, const float* pose, int n, float* best, float* best_pose,
                    eslam_stream_t stream);

/* Single-GPU forward of the fused loss in ONE launch: the sums of eslam_loss_reduce, acc [ESLAM_LOSS_ACC] (overwritten -
 * no pre-zeroing) and the loss value [1] (may be NULL).  scratch: ESLAM_LOSS_SCRATCH floats that the caller zeroes ONCE
 * when allocating them and then only hands to this function (it holds a self-resetting ticket counter and the
 * running sums, one cache line each; one scratch per concurrently running stream).  eslam_loss_grad(acc) gives the gradients. */
#define ESLAM_LOSS_SCRATCH (32 * 17)
/* Size (floats) of the scratch for a batch of n_rays: ESLAM_LOSS_SCRATCH, or more in deterministic mode.
 * eslam_deterministic(): 1 when the process runs with ESLAM_DETERMINISTIC=1 - the loss's sums are then reduced in a fixed
 * order (per-workgroup slots instead of float atomics) and the plane-gradient scatter accumulates in 64-bit fixed point
 * (integer adds commute), so every output of the path is bitwise reproducible from run to run, at a lower speed.
 * Range of the fixed-point sums: units of 2^-44, so one contribution must stay below 2^18 = 2.6e5 in magnitude and a texel's
 * sum below 5.2e5; a contribution beyond the limit, or a NaN / Inf one, poisons the texel's float gradient with NaN instead
 * of being saturated silently.  One int64 shadow of the planes per device, allocated at the first (eager) use.
 * eslam_loss_scratch_reset: back to the freshly-zeroed state, e.g. after a graph was aborted mid-flight.             */
int eslam_deterministic(void);

/* Host-side helper of the Python layer (no reference counterpart): `waiter` waits for the work enqueued on `signaler` so
 * far - the fork / join of the side stream the ray ordering (eslam_ray_order, what the backward's scatter bundles by) runs
 * on beside the samplers and the forward kernel.  One hipEventRecord + hipStreamWaitEvent; valid inside a stream capture. */
int eslam_stream_wait(eslam_stream_t waiter, eslam_stream_t signaler);

/* Host-side helper: zero `bytes` bytes at `ptr` on `stream`.  The gradient buffer autograd hands to the optimiser has to be
 * zero before the scatter adds into it (the reference's `fill_` of the plane gradients is 15.9 % of its CPU step, SURVEY.md
 * section 8a6); cleared at the head of the side stream it runs beside the samplers. */
int eslam_zero_async(void* ptr, int64_t bytes, eslam_stream_t stream);
int64_t eslam_loss_scratch_floats(int64_t n_rays);
int eslam_loss_scratch_reset(float* scratch, int64_t floats, eslam_stream_t stream);
int eslam_loss_value(const float* depth, const float* rgb, const float* sdf, const float* z_vals,
                     const float* gt_depth, const float* gt_color, int R, int S, double truncation,
                     const float* weights5_host, const uint8_t* ray_mask, float* scratch, float* acc, float* loss,
                     eslam_stream_t stream);

/* Per-kernel device timing for bench.py's roofline line (HIP events recorded on the launch stream around each
 * kernel while enabled; adds nothing to the launch path when disabled).  Usage: enable(1); run one iteration;
 * synchronise the stream; read(ms) -> elapsed milliseconds of the LAST launch of each kernel, -1 if it did not run. */
#define ESLAM_PROF_KERNELS 12
int eslam_profile_enable(int on);
int eslam_profile_read(float* ms_out);
const char* eslam_profile_name(int kernel_id);

/* Ray-sharded mapping iteration WITHOUT a collective between forward and backward (round 3; SURVEY.md section 8(e):
 * "compute them redundantly on every rank").  Every rank holds the iteration's whole batch of rays (same get_samples draw,
 * src/Mapper.py:318-319) and renders its slice; what the backward needs from the other ranks' rays it computes itself:
 *
 * eslam_loss_set_sizes: the five set sizes the mapping loss takes its means over (src/Mapper.py:136-140,343,346) for all R
 *   rays of the batch, without rendering them - they depend on gt_depth, ray_mask and the depth-guided z_vals of the rays
 *   with depth only (src/utils/Renderer.py:85-105), which the kernel replays in LDS with the sampler's own arithmetic and
 *   random numbers (t_rand [R,S] injected, or NULL = the in-kernel numbers of eslam_sample_z_all_rng for seed / rng_state /
 *   global ray index).  acc_out [ESLAM_LOSS_ACC]: the count slots as floats, other slots 0 - hand it to
 *   eslam_render_bwd_loss as `acc`.  scratch: 7 x 32 uint32, zeroed once by the caller, left zeroed.
 * eslam_mark_rays: touched [n_blocks] (cleared here) <- 1 for every texel that CAN receive gradient from the batch's rays,
 *   from ray geometry alone: the samples of a ray with depth d lie in [min(0, d - 1.5 tau), max(1.2 d, d + 1.5 tau)], those of
 *   a depth-less ray in [0, AABB exit + 0.01] (Renderer.py:96-100,114-134); the segment is rasterised conservatively into
 *   each plane.  A superset of the texels the ranks' backward passes add to, identical on every rank, known before anything
 *   is sampled.  channels_last planes only (one block = one texel's 32 channels = 128 bytes of the flat gradient buffer);
 *   block_base_host[12] = index of each plane's first block in that buffer.
 * eslam_blocks_compact: idx [n_blocks capacity] <- ascending indices of the non-zero bytes of touched; meta[0] <- their
 *   number, meta[1] += 1 (a stamp: the host compares it with its own count of launches before it sizes the all-reduce).
 *   host_meta_dev (optional): the device address of two int32 of pinned host memory (eslam_host_meta_alloc) that receive
 *   the same two words straight from the kernel - no copy node.  clear_touched: zero the bytes behind the read, so that the
 *   next iteration's marking needs no memset.  scratch: eslam_blocks_compact_scratch_words(n_blocks) uint32, zeroed once.
 * eslam_shard_prologue: the clear of the previous iteration's gradients (eslam_blocks_zero_dev; clear_flat NULL = none), the
 *   set sizes (eslam_loss_set_sizes, in-kernel random numbers) and the marking (eslam_mark_rays WITHOUT its memset: touched
 *   must be clean - eslam_blocks_compact(clear_touched = 1) leaves it so; mark_planes NULL = none) as ONE launch: a replayed
 *   hipGraph pays ~3 us per node, and this work sits beside the sampler and the forward kernel.
 * eslam_blocks_pack_dev / _unpack_dev / _zero_dev: gather the listed blocks of `flat` and the dense `tail` (decoder / beta /
 *   pose gradients, loss sums) into buf / write the all-reduced buf back / zero them, with the list's length read on the
 *   device (meta[0]) and the dense tail FIRST: buf = [tail_pad floats (n_tail used, rest 0; tail_pad % 4 == 0) | 32 floats per
 *   listed block], so that the all-reduce covers buf[0 : tail_pad + 32 meta[0]] of a fixed-capacity buffer.  step_bump
 *   (pack, optional): a device counter the launch increments - a ray-sharded iteration keeps its random-number step there
 *   (rng_state of eslam_sample_z_all_rng and eslam_shard_prologue, which read it on two streams) and advances it with its
 *   last launch instead of through eslam_render_fwd*'s rng_bump.                                                       */
int eslam_loss_set_sizes(const float* gt_depth, const uint8_t* ray_mask, int R, int n_strat, int n_imp, double truncation,
                         const float* t_free, const float* t_surf, const float* t_rand, int perturb, uint64_t seed,
                         const uint32_t* rng_state, uint32_t* scratch, float* acc_out, eslam_stream_t stream);
int eslam_mark_rays(const eslam_plane_t* planes, const float* bound6_host, const float* rays_o, const float* rays_d,
                    const float* gt_depth, int R, double truncation, const int64_t* block_base_host, int64_t n_blocks,
                    uint8_t* touched, eslam_stream_t stream);
int64_t eslam_blocks_compact_scratch_words(int64_t n_blocks);
int eslam_blocks_compact(uint8_t* touched, int64_t n_blocks, uint32_t* scratch, int32_t* idx, int32_t* meta,
                         int32_t* host_meta_dev, int clear_touched, eslam_stream_t stream);
int eslam_host_meta_alloc(void** host_ptr, void** dev_ptr);
int eslam_host_meta_free(void* host_ptr);
int eslam_shard_prologue(float* clear_flat, const int32_t* clear_idx, const int32_t* clear_meta, int64_t clear_capacity,
                         float* clear_tail, int64_t clear_n_tail, const float* rays_o, const float* rays_d,
                         const float* gt_depth, const uint8_t* ray_mask, int R, int n_strat, int n_imp, double truncation,
                         const float* t_free, const float* t_surf, int perturb, uint64_t seed, const uint32_t* rng_state,
                         uint32_t* sizes_scratch, float* acc_out, const eslam_plane_t* mark_planes, const float* bound6_host,
                         const int64_t* block_base_host, int64_t n_blocks, uint8_t* touched, eslam_stream_t stream);
int eslam_blocks_pack_dev(const float* flat, const int32_t* idx, const int32_t* meta, int64_t capacity, const float* tail,
                          int64_t n_tail, int64_t tail_pad, float* buf, uint32_t* step_bump, eslam_stream_t stream);
int eslam_blocks_unpack_dev(float* flat, const int32_t* idx, const int32_t* meta, int64_t capacity, float* tail,
                            int64_t n_tail, int64_t tail_pad, const float* buf, eslam_stream_t stream);
int eslam_blocks_zero_dev(float* flat, const int32_t* idx, const int32_t* meta, int64_t capacity, float* tail, int64_t n_tail,
                          eslam_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ESLAM_HIP_H */
