// Host glue of one render call as compiled code (optional; myslam_amd/ops.py holds the same logic in Python and stays the
// general path).  What Mapper.py:334-350 pays per iteration when it calls the library unchanged - no hipGraph - is host time:
// the Python layer spent ~0.5 ms per iteration (descriptor building, ~15 tensor constructions, 5 ctypes calls, 4 stream
// fork/joins, autograd glue) for 0.28 ms of GPU work.  Here ONE call does what Renderer.render_batch_ray does on its common
// path - ray order + gradient clear on a side stream, sampler, forward (with the loss's sums when asked), join - and the autograd
// node's backward is ONE call into eslam_render_bwd(_loss).  Everything goes through the C ABI of include/eslam_hip.h; PyTorch
// is plumbing (tensors, the caller's stream, the autograd graph the reference's loops call .backward() on).
//
// Common path = float32 planes (channels-last or NCHW: whatever the kernels accept), random numbers drawn in the sampler kernel,
// n_stratified >= 3, no injected random tensors, no mixed precision, no gradient sink (the ray-sharded mapper keeps the Python
// path).  Built by `make torch_ext` with g++ against the torch headers; absent or ESLAM_TORCH_EXT=0 -> the Python path runs.
#include <torch/extension.h>
#include <c10/hip/HIPGuard.h>
#include <c10/hip/HIPStream.h>
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>

#include "../../include/eslam_hip.h"

namespace {

using torch::Tensor;
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

void check(int rc, const char* what) {
    TORCH_CHECK(rc == 0, what, " failed (rc=", rc, "): ", eslam_last_error());
}

struct Config {                    // what a Renderer holds (src/utils/Renderer.py:34-44) + the decoders' bound
    int64_t n_strat = 0, n_imp = 0;
    bool perturb = true;
    double truncation = 0.0;
    std::vector<float> bound_sample;     // renderer.bound: the importance sampler's AABB (Renderer.py:114-117)
    std::vector<float> bound_decode;     // decoders.bound: what points are normalised with (decoders.py:138)
};

void fill_planes(const std::vector<Tensor>& planes, const std::vector<Tensor>* grads, eslam_plane_t* out) {
    TORCH_CHECK(planes.size() == ESLAM_N_PLANES, "expected 12 planes");
    for (int k = 0; k < ESLAM_N_PLANES; ++k) {
        const Tensor& p = planes[k];
        TORCH_CHECK(p.is_cuda() && p.scalar_type() == at::kFloat, "plane ", k, ": expected a float32 tensor on the GPU; the HIP render path has no CPU fallback");
        TORCH_CHECK(p.dim() == 4 && p.size(0) == 1 && p.size(1) == ESLAM_C_DIM, "plane ", k, ": expected shape [1,32,h,w]");
        eslam_plane_t& d = out[k];
        d.data = p.data_ptr<float>();
        d.grad = nullptr;
        d.h = (int32_t)p.size(2);
        d.w = (int32_t)p.size(3);
        d.stride_c = p.stride(1); d.stride_y = p.stride(2); d.stride_x = p.stride(3);
        d.data_f16 = nullptr;
        if (grads) {
            const Tensor& g = (*grads)[k];
            TORCH_CHECK(g.sizes() == p.sizes() && g.strides() == p.strides(), "plane gradient buffer must have the plane's shape and strides");
            d.grad = g.data_ptr<float>();
        }
    }
}

void fill_decoders(const std::vector<Tensor>& params, const Tensor& beta, eslam_decoders_t* d) {
    static const int64_t shapes[12][2] = {{16, 64}, {16, 0}, {16, 16}, {16, 0}, {1, 16}, {1, 0}, {16, 64}, {16, 0}, {16, 16}, {16, 0}, {3, 16}, {3, 0}};
    TORCH_CHECK(params.size() == 12, "expected 12 decoder tensors");
    const float* ptr[12];
    for (int i = 0; i < 12; ++i) {
        const Tensor& t = params[i];
        TORCH_CHECK(t.is_cuda() && t.scalar_type() == at::kFloat && t.is_contiguous(), "decoder parameter ", i, ": expected a contiguous float32 tensor on the GPU");
        const bool ok = shapes[i][1] ? (t.dim() == 2 && t.size(0) == shapes[i][0] && t.size(1) == shapes[i][1]) : (t.dim() == 1 && t.size(0) == shapes[i][0]);
        TORCH_CHECK(ok, "decoder parameter ", i, ": unexpected shape (c_dim=32, hidden=16, 2 blocks)");
        ptr[i] = t.data_ptr<float>();
    }
    TORCH_CHECK(beta.is_cuda() && beta.scalar_type() == at::kFloat && beta.numel() == 1, "beta: expected a float32 device tensor [1]");
    d->w1 = ptr[0]; d->b1 = ptr[1]; d->w2 = ptr[2]; d->b2 = ptr[3]; d->w3 = ptr[4]; d->b3 = ptr[5];
    d->cw1 = ptr[6]; d->cb1 = ptr[7]; d->cw2 = ptr[8]; d->cb2 = ptr[9]; d->cw3 = ptr[10]; d->cb3 = ptr[11];
    d->beta = beta.data_ptr<float>();
}

// one side stream per device, owned by this module: the ray ordering and the gradient clear run there beside the sampler and
// the forward kernel, forked and joined with events (valid inside a stream capture: they become graph edges)
hipStream_t side_stream(int dev) {
    static std::mutex mu;
    static hipStream_t streams[64];
    TORCH_CHECK(dev >= 0 && dev < 64, "device index out of range");
    std::lock_guard<std::mutex> lk(mu);
    if (!streams[dev]) TORCH_CHECK(hipStreamCreateWithFlags(&streams[dev], hipStreamNonBlocking) == hipSuccess, "hipStreamCreate failed");
    return streams[dev];
}

struct LossArgs {                  // the fused mapping loss (ops.fused_loss); gt_color undefined = no loss
    Tensor gt_color, ray_mask, scratch, acc_out;
    std::vector<float> weights5;
};

// Flat buffer (uninitialised) + 12 views of the planes' shapes: with the planes' own strides (autograd adopts such views as
// gradients without a copy), or dense channels-last / NCHW-contiguous whatever the planes are
enum class Layout { same, channels_last, nchw };
std::pair<Tensor, std::vector<Tensor>> alloc_plane_grads(const std::vector<Tensor>& planes, Layout layout = Layout::same) {
    int64_t total = 0;
    for (auto& p : planes) total += p.numel();
    Tensor flat = at::empty({total}, planes[0].options());
    std::vector<Tensor> views;
    views.reserve(12);
    int64_t off = 0;
    for (auto& p : planes) {
        TORCH_CHECK(p.is_contiguous() || p.is_contiguous(at::MemoryFormat::ChannelsLast), "planes must be dense (contiguous or channels_last)");
        const int64_t c = p.size(1), h = p.size(2), w = p.size(3);
        if (layout == Layout::same) views.push_back(at::as_strided(flat, p.sizes(), p.strides(), off));
        else if (layout == Layout::channels_last) views.push_back(at::as_strided(flat, p.sizes(), {c * h * w, 1, c * w, c}, off));
        else views.push_back(at::as_strided(flat, p.sizes(), {c * h * w, h * w, w, 1}, off));
        off += p.numel();
    }
    return {flat, views};
}

// eslam_planes_relayout between two lists of 12 tensors (field 0: as values, 1: as gradients)
void relayout(const std::vector<Tensor>& src, const std::vector<Tensor>& dst, int field, hipStream_t st) {
    eslam_plane_t a[ESLAM_N_PLANES], b[ESLAM_N_PLANES];
    fill_planes(src, field ? &src : nullptr, a);
    fill_planes(dst, field ? &dst : nullptr, b);
    check(eslam_planes_relayout(a, b, field, st), "eslam_planes_relayout");
}

class RenderNode : public torch::autograd::Function<RenderNode> {
public:
    // inputs: rays_o, rays_d, beta, 12 planes, 12 decoder tensors  (27 tensors; everything else rides in `io`)
    struct IO {
        Config cfg;
        Tensor gt_depth, t_free, t_surf, rng_state;
        uint64_t seed = 0;
        int64_t ray_offset = 0;
        LossArgs loss;
        bool needs = false, planes_grad = false;      // anything / the planes want a gradient (under the CALLER's grad mode)
        // planes in the reference's NCHW layout (src/ESLAM.py:199-210) and a batch large enough to pay for two copies of them: the
        // kernels run on channels-last scratch filled here, and the backward hands the gradients back in the planes' own strides
        // (what ops.ChannelsLastFn does around the Python path; nothing is kept across calls: src/Mapper.py:254-266)
        bool relayout = false;
    };

    static variable_list forward(AutogradContext* ctx, at::TensorList in, const IO& io) {
        const Tensor& rays_o_in = in[0];
        const Tensor& rays_d_in = in[1];
        const Tensor& beta = in[2];
        std::vector<Tensor> planes(in.begin() + 3, in.begin() + 15), params(in.begin() + 15, in.begin() + 27);
        TORCH_CHECK(rays_o_in.is_cuda() && rays_o_in.scalar_type() == at::kFloat, "rays_o: expected a float32 tensor on the GPU; the HIP render path has no CPU fallback");
        TORCH_CHECK(rays_d_in.is_cuda() && rays_d_in.scalar_type() == at::kFloat, "rays_d: expected a float32 tensor on the GPU");
        TORCH_CHECK(io.gt_depth.is_cuda() && io.gt_depth.scalar_type() == at::kFloat, "gt_depth: expected a float32 tensor on the GPU");
        const Tensor rays_o = rays_o_in.detach().contiguous(), rays_d = rays_d_in.detach().contiguous();
        const Tensor gd = io.gt_depth.detach().reshape({-1}).contiguous();
        const int64_t R = gd.size(0), S = io.cfg.n_strat + io.cfg.n_imp;
        TORCH_CHECK(rays_o.size(0) == R && rays_d.size(0) == R, "rays and gt_depth disagree on the number of rays");
        const int dev = rays_o.get_device();
        c10::hip::HIPGuard guard(dev);
        hipStream_t st = c10::hip::getCurrentHIPStream(dev).stream();
        const bool with_loss = io.loss.gt_color.defined();

        // (grad mode is OFF inside a C++ Function's forward: what the caller's mode and the inputs ask for was read in render())
        const bool needs = io.needs, planes_grad = io.planes_grad;
        auto opt = rays_o.options();
        Tensor z = at::empty({R, S}, opt), depth = at::empty({R}, opt), rgb = at::empty({R, 3}, opt), sdf = at::empty({R, S}, opt);
        Tensor raw_rgb, feat, perm, acc, value;
        if (needs) {
            raw_rgb = at::empty({R, S, 3}, opt);
            feat = at::empty({R * S, 128}, opt);
            perm = at::empty({planes_grad ? ESLAM_RAY_ORDER_WORDS(R) : 0}, opt.dtype(at::kInt));     // the scatter's bundling orders: only planes with a gradient need them
        }
        if (io.relayout && R > 0) {
            auto cl = alloc_plane_grads(planes, Layout::channels_last);
            relayout(planes, cl.second, 0, st);
            planes = std::move(cl.second);
        }
        eslam_plane_t pd[ESLAM_N_PLANES];
        fill_planes(planes, nullptr, pd);
        eslam_decoders_t dd;
        fill_decoders(params, beta, &dd);

        // side stream: gradient clear (27-70 MB) + ray order, beside the sampler and the forward kernel
        std::vector<Tensor> grad_views;
        Tensor grad_flat;
        hipStream_t side = nullptr;
        if (planes_grad && R > 0) {       // (tracking - planes without a gradient - has nothing for the side stream to do)
            side = side_stream(dev);
            auto gv = alloc_plane_grads(planes);
            grad_flat = gv.first;
            grad_views = std::move(gv.second);
            check(eslam_stream_wait(side, st), "eslam_stream_wait");
            check(eslam_zero_async(grad_flat.data_ptr<float>(), grad_flat.numel() * 4, side), "eslam_zero_async");
            check(eslam_ray_order(rays_o.data_ptr<float>(), rays_d.data_ptr<float>(), (int)R, perm.data_ptr<int32_t>(), side), "eslam_ray_order");
        }
        if (R > 0) {
            check(eslam_sample_z_all_rng(pd, &dd, io.cfg.bound_sample.data(), rays_o.data_ptr<float>(), rays_d.data_ptr<float>(),
                                         gd.data_ptr<float>(), (int)R, (int)io.cfg.n_strat, (int)io.cfg.n_imp, io.cfg.truncation,
                                         io.t_free.data_ptr<float>(), io.t_surf.data_ptr<float>(), io.cfg.perturb ? 1 : 0, io.seed,
                                         (const uint32_t*)io.rng_state.data_ptr<int32_t>(), io.ray_offset, z.data_ptr<float>(), st),
                  "eslam_sample_z_all_rng");
        }
        uint32_t* bump = (uint32_t*)io.rng_state.data_ptr<int32_t>();
        if (with_loss) {
            TORCH_CHECK(R > 0, "eslam_render_fwd_loss: empty batch");
            TORCH_CHECK(io.loss.gt_color.is_cuda() && io.loss.gt_color.scalar_type() == at::kFloat, "gt_color: expected a float32 tensor on the GPU");
            acc = io.loss.acc_out.defined() ? io.loss.acc_out : at::empty({16}, opt);
            value = at::empty({}, opt);
            const Tensor gc = io.loss.gt_color.contiguous();
            const uint8_t* mask = io.loss.ray_mask.defined() ? (const uint8_t*)io.loss.ray_mask.data_ptr() : nullptr;
            check(eslam_render_fwd_loss(pd, &dd, io.cfg.bound_decode.data(), rays_o.data_ptr<float>(), rays_d.data_ptr<float>(),
                                        z.data_ptr<float>(), (int)R, (int)S, depth.data_ptr<float>(), rgb.data_ptr<float>(),
                                        sdf.data_ptr<float>(), needs ? raw_rgb.data_ptr<float>() : nullptr,
                                        needs ? feat.data_ptr<float>() : nullptr, nullptr, gd.data_ptr<float>(), gc.data_ptr<float>(),
                                        io.cfg.truncation, io.loss.weights5.data(), mask, io.loss.scratch.data_ptr<float>(),
                                        acc.data_ptr<float>(), value.data_ptr<float>(), bump, st), "eslam_render_fwd_loss");
            ctx->saved_data["gt_color"] = gc;
            if (io.loss.ray_mask.defined()) ctx->saved_data["ray_mask"] = io.loss.ray_mask;
            ctx->saved_data["acc"] = acc;
            ctx->saved_data["w5"] = std::vector<double>(io.loss.weights5.begin(), io.loss.weights5.end());
        } else if (R > 0) {
            check(eslam_render_fwd(pd, &dd, io.cfg.bound_decode.data(), rays_o.data_ptr<float>(), rays_d.data_ptr<float>(), z.data_ptr<float>(),
                                   (int)R, (int)S, depth.data_ptr<float>(), rgb.data_ptr<float>(), sdf.data_ptr<float>(),
                                   needs ? raw_rgb.data_ptr<float>() : nullptr, needs ? feat.data_ptr<float>() : nullptr, nullptr, bump, st),
                  "eslam_render_fwd");
        }
        if (side) check(eslam_stream_wait(st, side), "eslam_stream_wait");      // join behind the forward kernel
        if (needs) {
            variable_list saved = {rays_o, rays_d, z, sdf, raw_rgb, feat, perm, beta, depth, rgb, gd};
            saved.insert(saved.end(), planes.begin(), planes.end());
            saved.insert(saved.end(), params.begin(), params.end());
            ctx->save_for_backward(saved);
            ctx->saved_data["with_loss"] = with_loss;
            ctx->saved_data["relayout"] = io.relayout && R > 0;
            ctx->saved_data["truncation"] = io.cfg.truncation;
            ctx->saved_data["bound"] = std::vector<double>(io.cfg.bound_decode.begin(), io.cfg.bound_decode.end());
            if (planes_grad) ctx->saved_data["grad_views"] = grad_views;
        }
        ctx->set_materialize_grads(false);
        ctx->mark_non_differentiable({z});
        variable_list out = {depth, rgb, sdf, z};
        if (with_loss) {
            out.push_back(value);
            out.push_back(acc);
            ctx->mark_non_differentiable({acc});
        }
        return out;
    }

    static variable_list backward(AutogradContext* ctx, variable_list g) {
        auto saved = ctx->get_saved_variables();
        TORCH_CHECK(saved.size() == 35, "eslam_torch_ext: expected 35 saved tensors, got ", saved.size());
        const Tensor &rays_o = saved[0], &rays_d = saved[1], &z = saved[2], &sdf = saved[3], &raw_rgb = saved[4], &feat = saved[5],
                     &perm = saved[6], &beta = saved[7], &depth = saved[8], &rgb = saved[9], &gd = saved[10];
        std::vector<Tensor> planes(saved.begin() + 11, saved.begin() + 23), params(saved.begin() + 23, saved.begin() + 35);
        const int64_t R = z.size(0), S = z.size(1);
        const int dev = rays_o.get_device();
        c10::hip::HIPGuard guard(dev);
        hipStream_t st = c10::hip::getCurrentHIPStream(dev).stream();
        const bool with_loss = ctx->saved_data["with_loss"].toBool();
        const bool need_ro = ctx->needs_input_grad(0), need_rd = ctx->needs_input_grad(1), need_beta = ctx->needs_input_grad(2);
        bool need_planes = false, need_dec = false;
        for (int i = 0; i < 12; ++i) need_planes = need_planes || ctx->needs_input_grad(3 + i);
        for (int i = 0; i < 12; ++i) need_dec = need_dec || ctx->needs_input_grad(15 + i);
        const bool need_rays = need_ro || need_rd;
        auto opt = rays_o.options();
        std::vector<Tensor> grad_views;
        if (need_planes) {
            if (ctx->saved_data.count("grad_views")) {
                grad_views = ctx->saved_data["grad_views"].toTensorVector();
                ctx->saved_data.erase("grad_views");
            } else {
                auto gv = alloc_plane_grads(planes);
                gv.first.zero_();
                grad_views = std::move(gv.second);
            }
        }
        eslam_plane_t pd[ESLAM_N_PLANES];
        fill_planes(planes, need_planes ? &grad_views : nullptr, pd);
        eslam_decoders_t dd;
        fill_decoders(params, beta, &dd);
        Tensor g_dec = need_dec ? at::empty({ESLAM_N_DEC_PARAMS}, opt) : Tensor();
        Tensor g_beta = need_beta ? at::empty({1}, opt) : Tensor();
        Tensor g_ro = need_rays ? at::empty({R, 3}, opt) : Tensor(), g_rd = need_rays ? at::empty({R, 3}, opt) : Tensor();
        if (R == 0) {
            if (need_dec) g_dec.zero_();
            if (need_beta) g_beta.zero_();
        }
        Tensor ws = at::empty({eslam_bwd_workspace_bytes(R * S)}, opt.dtype(at::kByte));
        // keep contiguous copies alive across the launch
        Tensor gdep = g[0].defined() ? g[0].contiguous() : Tensor(), grgb = g[1].defined() ? g[1].contiguous() : Tensor(),
               gsdf = g[2].defined() ? g[2].contiguous() : Tensor();
        auto bound_d = ctx->saved_data["bound"].toDoubleVector();
        float bound[6];
        for (int i = 0; i < 6; ++i) bound[i] = (float)bound_d[i];
        const double truncation = ctx->saved_data["truncation"].toDouble();
        const bool loss_grad = with_loss && g.size() > 4 && g[4].defined();
        if (loss_grad) {
            const Tensor up = g[4].detach().reshape({1}).to(at::kFloat).contiguous();
            const Tensor gc = ctx->saved_data["gt_color"].toTensor();
            const Tensor acc = ctx->saved_data["acc"].toTensor();
            const uint8_t* mask = ctx->saved_data.count("ray_mask") ? (const uint8_t*)ctx->saved_data["ray_mask"].toTensor().data_ptr() : nullptr;
            auto w5d = ctx->saved_data["w5"].toDoubleVector();
            float w5[5];
            for (int i = 0; i < 5; ++i) w5[i] = (float)w5d[i];
            check(eslam_render_bwd_loss(pd, &dd, bound, rays_o.data_ptr<float>(), rays_d.data_ptr<float>(), z.data_ptr<float>(), (int)R, (int)S,
                                        sdf.data_ptr<float>(), raw_rgb.data_ptr<float>(), feat.data_ptr<float>(), depth.data_ptr<float>(),
                                        rgb.data_ptr<float>(), gd.data_ptr<float>(), gc.data_ptr<float>(), truncation, w5, mask,
                                        acc.data_ptr<float>(), up.data_ptr<float>(), nullptr, gdep.defined() ? gdep.data_ptr<float>() : nullptr,
                                        grgb.defined() ? grgb.data_ptr<float>() : nullptr, gsdf.defined() ? gsdf.data_ptr<float>() : nullptr,
                                        need_dec ? g_dec.data_ptr<float>() : nullptr, need_beta ? g_beta.data_ptr<float>() : nullptr,
                                        need_rays ? g_ro.data_ptr<float>() : nullptr, need_rays ? g_rd.data_ptr<float>() : nullptr,
                                        perm.numel() ? perm.data_ptr<int32_t>() : nullptr, ws.data_ptr(), st), "eslam_render_bwd_loss");
        } else {
            check(eslam_render_bwd(pd, &dd, bound, rays_o.data_ptr<float>(), rays_d.data_ptr<float>(), z.data_ptr<float>(), (int)R, (int)S,
                                   sdf.data_ptr<float>(), raw_rgb.data_ptr<float>(), feat.data_ptr<float>(), gdep.defined() ? gdep.data_ptr<float>() : nullptr,
                                   grgb.defined() ? grgb.data_ptr<float>() : nullptr, gsdf.defined() ? gsdf.data_ptr<float>() : nullptr,
                                   need_dec ? g_dec.data_ptr<float>() : nullptr, need_beta ? g_beta.data_ptr<float>() : nullptr,
                                   need_rays ? g_ro.data_ptr<float>() : nullptr, need_rays ? g_rd.data_ptr<float>() : nullptr,
                                   perm.numel() ? perm.data_ptr<int32_t>() : nullptr, ws.data_ptr(), st), "eslam_render_bwd");
        }
        variable_list out(28);             // 27 tensor inputs + the IO argument's (undefined) slot
        if (need_ro) out[0] = g_ro;
        if (need_rd) out[1] = g_rd;
        if (need_beta) out[2] = g_beta;
        if (need_planes && ctx->saved_data["relayout"].toBool()) {       // channels-last scratch gradients -> the planes' NCHW strides
            auto nchw = alloc_plane_grads(grad_views, Layout::nchw);
            relayout(grad_views, nchw.second, 1, st);
            grad_views = std::move(nchw.second);
        }
        for (int i = 0; i < 12; ++i)
            if (need_planes && ctx->needs_input_grad(3 + i)) out[3 + i] = grad_views[i];
        if (need_dec) {
            static const int64_t sizes[12] = {1024, 16, 256, 16, 16, 1, 1024, 16, 256, 16, 48, 3};
            int64_t off = 0;
            for (int i = 0; i < 12; ++i) {
                if (ctx->needs_input_grad(15 + i)) out[15 + i] = g_dec.narrow(0, off, sizes[i]).view(params[i].sizes());
                off += sizes[i];
            }
        }
        return out;
    }
};

// The callers' loss as its own autograd node (src/Mapper.py:110-144,337-346; with ray_mask src/Tracker.py:114-148,197-204): forward =
// eslam_loss_value (sums, set sizes and the value in one launch), backward = eslam_loss_grad scaled by the incoming gradient.
class LossNode : public torch::autograd::Function<LossNode> {
public:
    struct IO {
        Tensor z, gt_depth, gt_color, ray_mask, scratch;
        double truncation = 0.0;
        std::vector<float> w5;
    };
    static Tensor forward(AutogradContext* ctx, const Tensor& depth_in, const Tensor& rgb_in, const Tensor& sdf_in, const IO& io) {
        for (const Tensor* t : {&depth_in, &rgb_in, &sdf_in, &io.z, &io.gt_depth, &io.gt_color})
            TORCH_CHECK(t->is_cuda() && t->scalar_type() == at::kFloat, "loss: expected float32 tensors on the GPU; the HIP render path has no CPU fallback");
        const Tensor depth = depth_in.detach().contiguous(), rgb = rgb_in.detach().contiguous(), sdf = sdf_in.detach().contiguous();
        const Tensor z = io.z.detach().contiguous(), gd = io.gt_depth.detach().contiguous(), gc = io.gt_color.detach().contiguous();
        const int64_t R = sdf.size(0), S = sdf.size(1);
        const int dev = sdf.get_device();
        c10::hip::HIPGuard guard(dev);
        hipStream_t st = c10::hip::getCurrentHIPStream(dev).stream();
        Tensor acc = at::empty({16}, sdf.options()), loss = at::empty({}, sdf.options());
        const uint8_t* mask = io.ray_mask.defined() ? (const uint8_t*)io.ray_mask.data_ptr() : nullptr;
        check(eslam_loss_value(depth.data_ptr<float>(), rgb.data_ptr<float>(), sdf.data_ptr<float>(), z.data_ptr<float>(), gd.data_ptr<float>(),
                               gc.data_ptr<float>(), (int)R, (int)S, io.truncation, io.w5.data(), mask, io.scratch.data_ptr<float>(),
                               acc.data_ptr<float>(), loss.data_ptr<float>(), st), "eslam_loss_value");
        ctx->save_for_backward({depth, rgb, sdf, z, gd, gc, acc});
        if (io.ray_mask.defined()) ctx->saved_data["ray_mask"] = io.ray_mask;
        ctx->saved_data["truncation"] = io.truncation;
        ctx->saved_data["w5"] = std::vector<double>(io.w5.begin(), io.w5.end());
        return loss;
    }
    static variable_list backward(AutogradContext* ctx, variable_list g) {
        auto sv = ctx->get_saved_variables();
        TORCH_CHECK(sv.size() == 7, "eslam_torch_ext: loss node lost its saved tensors");
        const Tensor &depth = sv[0], &rgb = sv[1], &sdf = sv[2], &z = sv[3], &gd = sv[4], &gc = sv[5], &acc = sv[6];
        const int64_t R = sdf.size(0), S = sdf.size(1);
        const int dev = sdf.get_device();
        c10::hip::HIPGuard guard(dev);
        hipStream_t st = c10::hip::getCurrentHIPStream(dev).stream();
        auto opt = sdf.options();
        Tensor g_depth = at::empty({R}, opt), g_rgb = at::empty({R, 3}, opt), g_sdf = at::empty({R, S}, opt);
        const Tensor up = g[0].detach().reshape({1}).to(at::kFloat).contiguous();
        auto w5d = ctx->saved_data["w5"].toDoubleVector();
        float w5[5];
        for (int i = 0; i < 5; ++i) w5[i] = (float)w5d[i];
        const uint8_t* mask = ctx->saved_data.count("ray_mask") ? (const uint8_t*)ctx->saved_data["ray_mask"].toTensor().data_ptr() : nullptr;
        check(eslam_loss_grad(depth.data_ptr<float>(), rgb.data_ptr<float>(), sdf.data_ptr<float>(), z.data_ptr<float>(), gd.data_ptr<float>(),
                              gc.data_ptr<float>(), (int)R, (int)S, ctx->saved_data["truncation"].toDouble(), w5, mask, acc.data_ptr<float>(),
                              nullptr, g_depth.data_ptr<float>(), g_rgb.data_ptr<float>(), g_sdf.data_ptr<float>(), up.data_ptr<float>(), st),
              "eslam_loss_grad");
        return {g_depth, g_rgb, g_sdf, Tensor()};
    }
};

Tensor mapping_loss(const Tensor& depth, const Tensor& rgb, const Tensor& sdf, const Tensor& z, const Tensor& gt_depth, const Tensor& gt_color,
                    double truncation, const std::vector<double>& weights5, const c10::optional<Tensor>& ray_mask, const Tensor& scratch) {
    TORCH_CHECK(weights5.size() == 5 && sdf.dim() == 2, "mapping_loss: 5 weights and sdf [R,S] expected");
    LossNode::IO io;
    io.z = z; io.gt_depth = gt_depth; io.gt_color = gt_color; io.scratch = scratch; io.truncation = truncation;
    io.w5.assign(weights5.begin(), weights5.end());
    if (ray_mask.has_value()) io.ray_mask = ray_mask->contiguous();
    return LossNode::apply(depth, rgb, sdf, io);
}

// depth, rgb, sdf, z_vals[, loss value, acc] = render(...)
std::vector<Tensor> render(const Config& cfg, const Tensor& rays_o, const Tensor& rays_d, const Tensor& gt_depth, const Tensor& beta,
                           const std::vector<Tensor>& planes, const std::vector<Tensor>& params, const Tensor& t_free, const Tensor& t_surf,
                           const Tensor& rng_state, uint64_t seed, int64_t ray_offset, const c10::optional<Tensor>& gt_color,
                           const c10::optional<Tensor>& ray_mask, const c10::optional<Tensor>& scratch, const c10::optional<Tensor>& acc_out,
                           const std::vector<double>& weights5, bool relayout) {
    TORCH_CHECK(cfg.n_strat >= 3 && cfg.bound_sample.size() == 6 && cfg.bound_decode.size() == 6, "Config not initialised");
    TORCH_CHECK(planes.size() == 12 && params.size() == 12, "expected 12 planes and 12 decoder tensors");
    RenderNode::IO io;
    io.cfg = cfg;
    io.gt_depth = gt_depth; io.t_free = t_free; io.t_surf = t_surf; io.rng_state = rng_state;
    io.seed = seed; io.ray_offset = ray_offset; io.relayout = relayout;
    TORCH_CHECK(t_free.numel() == cfg.n_strat && t_surf.numel() == cfg.n_imp && rng_state.scalar_type() == at::kInt, "bad sampler tensors");
    if (gt_color.has_value()) {
        TORCH_CHECK(scratch.has_value() && weights5.size() == 5, "the fused loss needs its scratch and 5 weights");
        io.loss.gt_color = *gt_color;
        if (ray_mask.has_value()) io.loss.ray_mask = ray_mask->contiguous();
        io.loss.scratch = *scratch;
        if (acc_out.has_value()) io.loss.acc_out = *acc_out;
        io.loss.weights5.assign(weights5.begin(), weights5.end());
    }
    if (torch::GradMode::is_enabled()) {
        bool dec_grad = false;
        for (auto& p : planes) io.planes_grad = io.planes_grad || p.requires_grad();
        for (auto& p : params) dec_grad = dec_grad || p.requires_grad();
        io.needs = io.planes_grad || dec_grad || beta.requires_grad() || rays_o.requires_grad() || rays_d.requires_grad();
    }
    variable_list in;
    in.reserve(27);
    in.push_back(rays_o); in.push_back(rays_d); in.push_back(beta);
    in.insert(in.end(), planes.begin(), planes.end());
    in.insert(in.end(), params.begin(), params.end());
    return RenderNode::apply(at::TensorList(in), io);
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.doc() = "compiled host glue of Renderer.render_batch_ray (see the header comment of eslam_torch_ext.cpp)";
    py::class_<Config>(m, "Config")
        .def(py::init([](int64_t n_strat, int64_t n_imp, bool perturb, double truncation, std::vector<float> bound_sample,
                         std::vector<float> bound_decode) {
            Config c;
            c.n_strat = n_strat; c.n_imp = n_imp; c.perturb = perturb; c.truncation = truncation;
            c.bound_sample = std::move(bound_sample); c.bound_decode = std::move(bound_decode);
            return c;
        }));
    m.def("render", &render, py::arg("cfg"), py::arg("rays_o"), py::arg("rays_d"), py::arg("gt_depth"), py::arg("beta"), py::arg("planes"),
          py::arg("params"), py::arg("t_free"), py::arg("t_surf"), py::arg("rng_state"), py::arg("seed"), py::arg("ray_offset"),
          py::arg("gt_color") = py::none(), py::arg("ray_mask") = py::none(), py::arg("scratch") = py::none(), py::arg("acc_out") = py::none(),
          py::arg("weights5") = std::vector<double>(), py::arg("relayout") = false);
    m.def("mapping_loss", &mapping_loss, py::arg("depth"), py::arg("rgb"), py::arg("sdf"), py::arg("z_vals"), py::arg("gt_depth"),
          py::arg("gt_color"), py::arg("truncation"), py::arg("weights5"), py::arg("ray_mask"), py::arg("scratch"));
    m.def("abi_version", []() { return eslam_abi_version(); });
}
