"""torch.autograd glue around the C-ABI kernels.  PyTorch is plumbing here (device memory, streams, the
autograd graph the reference's Mapper/Tracker call .backward() on); all arithmetic is in the HIP library.

Nothing in this file computes on the CPU: tensors that are not on a GPU raise.
"""
import ctypes
import os
import sys

import torch

from . import _hip

_const_cache = {}


def _cached(key, make):
    t = _const_cache.get(key)
    if t is None:
        t = make()
        _const_cache[key] = t
    return t


def beta_tensor(beta, device):
    """decoders.beta is an nn.Parameter or the Python int 10 (reference decoders.py:59-62)."""
    if torch.is_tensor(beta):
        return beta
    dev = torch.device(device)
    return _cached(("beta", float(beta), dev.index), lambda: torch.full((1,), float(beta), device=dev))


def linspace01(n, device):
    dev = torch.device(device)
    return _cached(("lin", n, dev.index), lambda: torch.linspace(0.0, 1.0, steps=n, device=dev))


def decoder_params(decoders):
    """The 12 tensors in C-ABI order from our Decoders or the reference's (same attribute names).  Walking the
    ModuleLists costs ~12 us; the list is remembered on the module and checked by identity of its first and last
    entries (nn.Module.to() / _apply replace the Parameter objects, load_state_dict copies in place)."""
    hit = decoders.__dict__.get("_eslam_param_list")
    if hit is not None and hit[0] is decoders.linears[0].weight and hit[11] is decoders.c_output_linear.bias:
        return hit
    lst = _decoder_params_slow(decoders)
    decoders.__dict__["_eslam_param_list"] = lst
    return lst


def _decoder_params_slow(decoders):
    return [decoders.linears[0].weight, decoders.linears[0].bias, decoders.linears[1].weight,
            decoders.linears[1].bias, decoders.output_linear.weight, decoders.output_linear.bias,
            decoders.c_linears[0].weight, decoders.c_linears[0].bias, decoders.c_linears[1].weight,
            decoders.c_linears[1].bias, decoders.c_output_linear.weight, decoders.c_output_linear.bias]


_bound_last = None


def bound_to_host(bound):
    """[3,2] tensor (the reference keeps decoders.bound on the CPU, ESLAM.py:173) -> 6 floats.  The last tensor's values are
    remembered (same object, same version counter: a render call asks twice per iteration)."""
    global _bound_last
    if torch.is_tensor(bound):
        last = _bound_last
        if last is not None and last[0] is bound and last[1] == bound._version:
            return last[2]
        vals = tuple(float(v) for v in bound.detach().reshape(-1).tolist())
        _bound_last = (bound, bound._version, vals)
        return vals
    return tuple(float(v) for v in bound)


def _c(t):
    return t if t is None or t.is_contiguous() else t.contiguous()


def _split_planes(flat12):
    return tuple([flat12[2 * g], flat12[2 * g + 1]] for g in range(6))


_grad_sink = None


class grad_sink:
    """Context manager: RenderFn.backward writes plane / decoder / beta gradients into the views of a
    parallel.FlatGrads (params = 12 planes, 12 decoder tensors, [beta]) instead of fresh buffers, so a data-parallel
    caller can all-reduce one flat buffer with no copy in between.  Enter it around the backward pass; the FORWARD pass of
    that call must have run inside this context or inside ops.keep_layout() (either keeps the render call on the Python
    glue, whose backward looks the sink up - the compiled glue, eslam_torch_ext, hands its gradients to autograd)."""

    def __init__(self, flat_grads):
        self.fg = flat_grads

    def __enter__(self):
        global _grad_sink
        self._prev, _grad_sink = _grad_sink, self.fg
        return self.fg

    def __exit__(self, *a):
        global _grad_sink
        _grad_sink = self._prev


_grad_layout_cache = {}
_grad_reuse = {}


def _alloc_plane_grads(planes, zero=True):
    """One flat zero buffer, 12 views with the planes' own strides (so autograd can adopt them without a copy
    and a multi-GPU caller can all-reduce the flat buffer).  zero=False: uninitialised (the caller clears it)."""
    key = tuple((tuple(p.shape), p.stride()) for p in planes)
    layout = _grad_layout_cache.get(key)
    if layout is None:
        off, layout = 0, []
        for p in planes:
            if not (p.is_contiguous() or p.is_contiguous(memory_format=torch.channels_last)):
                raise RuntimeError("planes must be dense (contiguous or channels_last)")
            layout.append((tuple(p.shape), p.stride(), off))
            off += p.numel()
        layout = _grad_layout_cache[key] = (layout, off)
    entries, total = layout
    dev = planes[0].device
    # The previous call's buffer and its 12 views are handed out again when NOBODY else holds them any more (the caller dropped
    # the gradients: optimizer.zero_grad(set_to_none=True), p.grad = None): 13 tensor constructions (~30 us of host time per
    # iteration) become 12 reference-count reads.  A view that is still somebody's .grad keeps its count up, and a fresh
    # buffer is built.
    ck = (key, dev.index)
    old = _grad_reuse.get(ck)
    if old is not None and not torch.cuda.is_current_stream_capturing():
        flat, views = old
        # the LIST is what an autograd node of a forward pass without its backward yet holds on to (ctx.pregrads); the ELEMENTS
        # are what became somebody's .grad.  Counts: cache tuple + local + argument (list); list + loop variable + argument (views)
        if sys.getrefcount(views) == 3 and all(sys.getrefcount(v) == 3 for v in views):
            if zero:
                flat.zero_()
            return flat, views
    flat = (torch.zeros if zero else torch.empty)(total, device=dev, dtype=torch.float32)
    views = [torch.as_strided(flat, shp, st, off) for shp, st, off in entries]
    if not torch.cuda.is_current_stream_capturing():
        _grad_reuse[ck] = (flat, views)
    return flat, views


def _flat_views(planes, memory_format):
    """One uninitialised flat buffer + 12 views of the planes' shapes in `memory_format` (dense)."""
    dev = planes[0].device
    total = sum(p.numel() for p in planes)
    flat = torch.empty(total, device=dev, dtype=torch.float32)
    views, off = [], 0
    for p in planes:
        n, c, h, w = p.shape
        st = (c * h * w, 1, c * w, c) if memory_format == torch.channels_last else (c * h * w, h * w, w, 1)
        views.append(torch.as_strided(flat, tuple(p.shape), st, off))
        off += p.numel()
    return flat, views


def _relayout(src, dst, field):
    """eslam_planes_relayout between two lists of 12 tensors (field 0: values; field 1: the tensors are gradients)."""
    dev = src[0].device
    a, _ = _hip.make_planes(_split_planes(src), src if field else None, remember=False)
    b, _ = _hip.make_planes(_split_planes(dst), dst if field else None, remember=False)
    with _hip.on_device(dev):
        _hip.check(_hip.lib().eslam_planes_relayout(a, b, int(field), _hip.stream_handle(dev)), "eslam_planes_relayout")


class ChannelsLastFn(torch.autograd.Function):
    """12 channels-last scratch copies = ChannelsLastFn.apply(*12 planes in the reference's NCHW layout, ESLAM.py:199-210).

    Forward: one launch (eslam_planes_relayout).  Backward: the gradients the render kernels scattered into channels-last
    buffers go back to the planes' own layout in one launch, so `p.grad.stride() == p.stride()` for the caller's optimiser.
    Nothing is kept across calls: the mapper swaps the plane Parameters every frame (Mapper.py:254-266)."""

    @staticmethod
    def forward(ctx, *planes):
        for k, p in enumerate(planes):
            _hip.require_gpu_f32(f"plane {k}", p)
            if not p.is_contiguous():
                raise RuntimeError("planes must be dense (contiguous or channels_last)")
        src = [p.detach() for p in planes]
        _, views = _flat_views(src, torch.channels_last)
        _relayout(src, views, 0)
        ctx.shapes = [tuple(p.shape) for p in planes]
        ctx.device = planes[0].device
        return tuple(views)

    @staticmethod
    def backward(ctx, *grads):
        if all(g is None for g in grads):
            return (None,) * 12
        if any(g is None for g in grads):           # (the render backward produces all 12 or none)
            grads = [g if g is not None else torch.zeros(shp, device=ctx.device).contiguous(memory_format=torch.channels_last)
                     for g, shp in zip(grads, ctx.shapes)]
        grads = [g if g.is_contiguous(memory_format=torch.channels_last) else g.contiguous(memory_format=torch.channels_last)
                 for g in grads]
        _, out = _flat_views(grads, torch.contiguous_format)
        _relayout(list(grads), out, 1)
        return tuple(out)


# Planes in the reference's NCHW layout are copied to channels-last scratch per call when the batch has at least this many
# points (below, the direct strided kernels are cheaper than moving 2 x 27-70 MB); ESLAM_NCHW_RELAYOUT_MIN=-1 disables it
_RELAYOUT_MIN_POINTS = int(os.environ.get("ESLAM_NCHW_RELAYOUT_MIN", "4096"))


_keep_layout = 0


class keep_layout:
    """Context manager: no per-call layout change inside (a caller that hands the kernels gradient buffers in the planes' own
    strides - parallel.ShardedMapper's flat exchange buffer - needs the kernels to see the planes as they are)."""

    def __enter__(self):
        global _keep_layout
        _keep_layout += 1

    def __exit__(self, *a):
        global _keep_layout
        _keep_layout -= 1


def wants_relayout(all_planes, n_points):
    """Whether this call should run on per-call channels-last copies of the planes: they are in the reference's NCHW layout
    and the batch is large enough to pay for the copies."""
    flat = [p for grp in all_planes for p in grp]
    if _keep_layout or _RELAYOUT_MIN_POINTS < 0 or n_points < _RELAYOUT_MIN_POINTS or len(flat) != 12:
        return False
    if all(p.dim() == 4 and p.shape[2] * p.shape[3] > 1 and p.is_contiguous(memory_format=torch.channels_last) for p in flat):
        return False
    # (anything else: let the kernels' own validation speak)
    return all(p.is_cuda and p.dtype == torch.float32 and p.dim() == 4 and p.is_contiguous() for p in flat)


def planes_for_kernels(all_planes, n_points):
    """all_planes as the kernels should see them: unchanged when channels-last (or the batch is small), else per-call
    channels-last scratch copies that carry the gradient back to the caller's planes (ChannelsLastFn)."""
    if not wants_relayout(all_planes, n_points):
        return all_planes
    return _split_planes(ChannelsLastFn.apply(*[p for grp in all_planes for p in grp]))


_DEC_SIZES = [16 * 64, 16, 16 * 16, 16, 16, 1, 16 * 64, 16, 16 * 16, 16, 48, 3]


def _split_dec_grads(g_dec):
    parts = g_dec.split(_DEC_SIZES)
    return [t.view(shape) if len(shape) > 1 else t for t, (_, _, shape) in zip(parts, _hip.DEC_FIELDS)]


_dec_param_cache = {}


_side_streams = {}


# ESLAM_PRECLEAR=0: the plane-gradient buffer is a torch.zeros in the backward pass instead of a memset on the side stream
_PRECLEAR = os.environ.get("ESLAM_PRECLEAR", "1") == "1"


def ray_order_async(rays_o, rays_d, grad_planes=None):
    """Launch eslam_ray_order on a side stream (it depends only on the rays, so it overlaps the samplers).
    Returns (perm int32 [3 R + 4]: one order per plane orientation + the fan's extent, stream to join before the order is used[, gradient views]).
    grad_planes: the 12 planes when the coming backward pass will need their gradient buffer: it is allocated here and
    cleared at the HEAD of the side stream, beside the samplers (27 - 70 MB: 6 - 15 us in front of the backward pass
    otherwise; beside the forward kernel the memset's writes slowed the forward down by more than that)."""
    _hip.require_gpu_f32("rays_o", rays_o)
    _hip.require_gpu_f32("rays_d", rays_d)
    dev = rays_o.device
    ro, rd = _c(rays_o.detach()), _c(rays_d.detach())
    R = ro.shape[0]
    side = _side_streams.get(dev.index)
    if side is None:
        side = _side_streams[dev.index] = torch.cuda.Stream(device=dev)
    perm = torch.empty(_hip.ray_order_words(R), dtype=torch.int32, device=dev)      # one order per plane orientation (+ the fan's extent)
    pre = None
    if grad_planes is not None and _PRECLEAR and _grad_sink is None and not _keep_layout:      # (keep_layout: a sink will take the gradients)
        pre = _alloc_plane_grads(grad_planes, zero=False)      # on the caller's stream: its allocator's memory
    _hip.stream_wait(dev, side, None)           # the rays - and any earlier use of that memory - come from work on the caller's stream
    with _hip.on_device(dev):
        if pre is not None:
            _hip.check(_hip.lib().eslam_zero_async(_hip.ptr(pre[0]), pre[0].numel() * 4, ctypes.c_void_p(side.cuda_stream)),
                       "eslam_zero_async")
        _hip.check(_hip.lib().eslam_ray_order(_hip.ptr(ro), _hip.ptr(rd), R, _hip.ptr(perm),
                                              ctypes.c_void_p(side.cuda_stream)), "eslam_ray_order")
    perm.record_stream(side)
    for t, src in ((ro, rays_o), (rd, rays_d)):
        if t.data_ptr() != src.data_ptr():
            t.record_stream(side)        # a contiguous copy made on the current stream and read on the side stream
    return (perm, side) if pre is None else (perm, side, pre[1])


# The forward kernel can process rays in the bundling order of the backward (ESLAM_FWD_ORDER=1).  Off: measured on
# MI355X it is faster on the rays as given (119 vs 123-125 us at 4096x64 - sorted neighbours hit the same L2 channels).
_FWD_USES_ORDER = os.environ.get("ESLAM_FWD_ORDER", "0") == "1"


_fused_loss = None

# The compiled host glue of a render call (myslam_amd/csrc/eslam_torch_ext.cpp, `make torch_ext`): optional - the Python code below
# is the general path and does the same through ctypes.  ESLAM_TORCH_EXT=0 switches it off.
_EXT_WANTED = os.environ.get("ESLAM_TORCH_EXT", "1") != "0"
_ext_mod = False


def torch_ext():
    """The eslam_torch_ext module, or None when it is switched off or not built."""
    global _ext_mod
    if _ext_mod is False:
        _ext_mod = None
        path = os.path.join(os.path.dirname(_hip.LIB_PATH), "eslam_torch_ext.so")
        if _EXT_WANTED and os.path.exists(path) and not os.environ.get("ESLAM_HIP_LIB"):
            import importlib.util
            _hip.load_library()                      # (the module links against libeslam_hip.so: resolve it from the same file)
            spec = importlib.util.spec_from_file_location("eslam_torch_ext", path)
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            if mod.abi_version() != _hip.ABI_VERSION:
                raise RuntimeError(f"eslam_torch_ext was built against ABI {mod.abi_version()}, the binding expects {_hip.ABI_VERSION}: "
                                   "run `make -C myslam_amd/csrc torch_ext`")
            _ext_mod = mod
    return _ext_mod


def ext_render_ok(rays_o, n_strat, rand):
    """Whether a render call may take the compiled path: the common case only (see eslam_torch_ext.cpp's header)."""
    return (rand is None and _USE_KERNEL_RNG and n_strat >= 3 and _half_planes is None and _grad_sink is None and not _keep_layout and
            _rng_override is None and not _FWD_USES_ORDER and rays_o.is_cuda and torch_ext() is not None)


def ext_render(cfg, rays_o, rays_d, gt_depth, beta, flat_planes, dec_params, n_strat, n_imp, fl, relayout=False):
    """One compiled call: [planes -> channels-last scratch when `relayout`,] ray order + gradient clear (side stream), sampler,
    forward (+ the loss's sums), join.  fl: the active ops.fused_loss context or None.  Returns depth, rgb, sdf, z_vals."""
    dev = rays_o.device
    seed_v = _rng_seed(dev)
    state = _rng_state(dev)
    if _rng_pending.get(dev.index, False):
        state[0] += 1                  # samples drawn earlier were never rendered
        _rng_pending[dev.index] = False
    ext = torch_ext()
    with _hip.on_device(dev):
        if fl is None:
            outs = ext.render(cfg, rays_o, rays_d, gt_depth, beta, flat_planes, dec_params, linspace01(n_strat, dev), linspace01(n_imp, dev),
                              state, seed_v, _ray_offset, relayout=relayout)
        else:
            mask = fl.ray_mask
            if mask is not None:
                mask = _c(mask.view(torch.uint8) if mask.dtype == torch.bool else mask.to(torch.uint8))
            outs = ext.render(cfg, rays_o, rays_d, gt_depth, beta, flat_planes, dec_params, linspace01(n_strat, dev), linspace01(n_imp, dev),
                              state, seed_v, _ray_offset, fl.gt_color, mask, _loss_scratch(dev, rays_o.shape[0]), fl.acc_out,
                              list(fl.state.weights5), relayout)
            fl.loss, fl.acc = outs[4], outs[5]
            fl.value = outs[4].detach()
    return outs[0], outs[1], outs[2], outs[3]


class _LossState:
    __slots__ = ("truncation", "weights5", "rewrite_value", "gt_depth", "gt_color", "ray_mask", "acc", "value", "acc_global")

    def __init__(self, truncation, weights5, rewrite_value):
        self.truncation, self.weights5, self.rewrite_value = float(truncation), tuple(float(v) for v in weights5), rewrite_value
        self.gt_depth = self.gt_color = self.ray_mask = self.acc = self.value = self.acc_global = None


class fused_loss:
    """Context manager: the next Renderer.render_batch_ray renders WITH the callers' loss: the forward kernel forms the
    loss's sums in its epilogue (eslam_render_fwd_loss) and the backward kernel forms the loss's gradients itself
    (eslam_render_bwd_loss) - no loss launches at all, and depth / rgb / sdf gradients never touch memory.
    After the call `.acc` [16], `.value` [1] and `.loss` (a 0-d tensor connected to the autograd graph: call
    .backward() on it, or pass the context as losses.mapping_loss(..., precomputed=ctx), which returns it) are set.
    A ray-sharded caller all-reduces `.acc` in place before the backward; the backward then also rewrites `.value` with
    the global loss.  Mapping-style loss only (the tracker's outlier mask depends on the rendered depth itself)."""

    def __init__(self, gt_depth, gt_color, truncation, weights5, ray_mask=None, rewrite_value=False, acc_out=None):
        self.gt_depth, self.gt_color, self.truncation, self.weights5, self.ray_mask = gt_depth, gt_color, truncation, weights5, ray_mask
        self.acc = self.value = self.loss = None
        self.acc_out = acc_out          # optional float32 [16] the forward kernel writes the sums and set sizes into
        # what the backward needs, in an object of its own: the autograd node keeps THIS alive, not the context, which
        # holds .loss and would otherwise close a reference cycle around the saved activations (134 MB at 4096 x 64)
        self.state = _LossState(truncation, weights5, rewrite_value)

    @property
    def acc_global(self):
        return self.state.acc_global

    @acc_global.setter
    def acc_global(self, t):
        self.state.acc_global = t

    def __enter__(self):
        global _fused_loss
        self._prev, _fused_loss = _fused_loss, self
        return self

    def __exit__(self, *a):
        global _fused_loss
        _fused_loss = self._prev


def current_fused_loss():
    return _fused_loss


_half_planes = None


class mixed_precision:
    """Context manager: Renderer.render_batch_ray calls issued inside run the mixed-precision kernels (BASELINE.json
    configs[4]: texels gathered from float16 copies of the planes, decoders on bf16 MFMA forward and backward, float32
    accumulation everywhere, plane gradients accumulated in float32 for the float32 masters).  half: a lowp.HalfPlanes (or any
    object with `.flat`, the 12 float16 channels_last copies in all_planes order) - refresh it after every optimiser step."""

    def __init__(self, half):
        self.half = half

    def __enter__(self):
        global _half_planes
        self._prev, _half_planes = _half_planes, self.half
        return self.half

    def __exit__(self, *a):
        global _half_planes
        _half_planes = self._prev


def join_ray_order(device):
    """Make the current stream wait for the ray-ordering side stream.  RenderFn joins it in its backward; a caller that
    captures forward and backward into SEPARATE hipGraphs must join inside the forward's capture (parallel.py)."""
    side = _side_streams.get(torch.device(device).index)
    if side is not None:
        _hip.stream_wait(torch.device(device), None, side)


# In-kernel random numbers of the samplers (eslam_sample_z_all_rng): a step counter per device, read by the sampler and
# advanced by the forward kernel that consumes its samples.  ESLAM_TORCH_RAND=1 restores the torch.rand pool.
_USE_KERNEL_RNG = os.environ.get("ESLAM_TORCH_RAND", "0") != "1"
_rng_pending = {}


def _rng_state(dev):
    return _cached(("rng_state", dev.index), lambda: torch.zeros(4, dtype=torch.int32, device=dev))


# Reproducibility contract of the in-kernel random numbers (the jitter of Renderer.py:59 and the importance draw of
# common.py:59 when nobody injects them): U = hash(key, step, stream, GLOBAL ray index, element).
#   key   ops.seed(s) if called, else torch's seeds: the CPU generator's (torch.manual_seed) folded with the device
#         generator's (torch.cuda.manual_seed) - re-seeding either gives a new key;
#   step  a device counter the forward kernel advances; it restarts at 0 whenever the key changes, so seeding twice with
#         the same value reproduces the same samples (a captured graph bakes the key in: re-seed before capturing);
#   ray   ops.ray_offset(lo) + local index: the ranks of a ray-sharded iteration draw what the unsharded batch draws.
_explicit_seed = None
_rng_key = {}
_ray_offset = 0


def seed(value):
    """Key the in-kernel random numbers explicitly (None: back to torch's seeds).  The step counter restarts."""
    global _explicit_seed
    _explicit_seed = None if value is None else int(value) & 0xFFFFFFFFFFFFFFFF
    _rng_key.clear()


def _rng_seed(dev):
    if _explicit_seed is not None:
        key = _explicit_seed
    else:
        gen = torch.cuda.default_generators[dev.index]
        key = (torch.initial_seed() * 0x9E3779B97F4A7C15 + gen.initial_seed()) & 0xFFFFFFFFFFFFFFFF
    if _rng_key.get(dev.index) != key:
        if not torch.cuda.is_current_stream_capturing():      # (a fill captured into a graph would restart the counter per replay)
            _rng_key[dev.index] = key
            _rng_state(dev).zero_()
            _rng_pending.pop(dev.index, None)
    return key


_rng_override = None


class rng_step:
    """Context manager: sampler calls issued inside read the random-number step from `state` (int32 [4], element 0) instead of
    the device's own counter, and the forward kernel does NOT advance it: the caller does (parallel.ShardedMapper advances it
    with the last launch of its iteration - its set-size replay reads the same step on another stream)."""

    def __init__(self, state):
        self.state = state

    def __enter__(self):
        global _rng_override
        self._prev, _rng_override = _rng_override, self.state

    def __exit__(self, *a):
        global _rng_override
        _rng_override = self._prev


class ray_offset:
    """Context manager: the rays of render / sampler calls issued inside are rays lo, lo + 1, ... of the iteration's whole
    batch (a ray-sharded rank rendering its slice): the in-kernel random numbers are keyed on the global index."""

    def __init__(self, lo):
        self.lo = int(lo)

    def __enter__(self):
        global _ray_offset
        self._prev, _ray_offset = _ray_offset, self.lo

    def __exit__(self, *a):
        global _ray_offset
        _ray_offset = self._prev


def rng_step_snapshot(dev, out=None):
    """A copy of the device step counter the NEXT sampler call will draw with (int32 [4]): lets loss_set_sizes run on another
    stream beside the forward kernel, which advances the live counter."""
    state = _rng_state(dev)
    if _rng_pending.get(dev.index, False):
        state[0] += 1                  # samples drawn earlier were never rendered: what the next sampler call would do
        _rng_pending[dev.index] = False
    _rng_seed(dev)                     # (a new seed restarts the counter: settle that before the copy)
    if out is None:
        return state.clone()
    out.copy_(state)
    return out


def loss_set_sizes(gt_depth, ray_mask, n_strat, n_imp, truncation, perturb, t_rand=None, out=None, state=None):
    """acc [16] with the mapping loss's five set sizes over ALL rays given (eslam_loss_set_sizes): what a ray-sharded rank
    computes for the iteration's whole batch instead of all-reducing its shard's counts.  t_rand [R,S]: injected jitter
    numbers (tests); None = the in-kernel numbers the sampler will draw for the same seed / step / global ray index - call
    it BEFORE the forward kernel that consumes the samples (that kernel advances the step), or hand it a rng_step_snapshot
    taken before as `state`."""
    _hip.require_gpu_f32("gt_depth", gt_depth)
    dev = gt_depth.device
    gd = _c(gt_depth.detach().reshape(-1))
    R = gd.shape[0]
    if ray_mask is not None:
        ray_mask = _c(ray_mask.view(torch.uint8) if ray_mask.dtype == torch.bool else ray_mask.to(torch.uint8))
    # ticket + five counters, zeroed once and left zeroed by the kernel: a slot per (device, stream) out of the pre-zeroed pool
    # the fused loss's scratch comes from (a buffer created inside a graph capture would be re-zeroed by a captured fill)
    scratch = _pooled_scratch(dev, 7 * 32, "set_sizes")
    acc = torch.empty(16, device=dev) if out is None else out
    seed_v = 0 if t_rand is not None else _rng_seed(dev)
    if state is None:
        state = _rng_state(dev)
        if t_rand is None and _rng_pending.get(dev.index, False):
            state[0] += 1              # samples drawn earlier were never rendered: the sampler below will advance the step, too
            _rng_pending[dev.index] = False
    with _hip.on_device(dev):
        _hip.check(_hip.lib().eslam_loss_set_sizes(_hip.ptr(gd), _hip.ptr(ray_mask), R, n_strat, n_imp, float(truncation),
                                                  _hip.ptr(linspace01(n_strat, dev)), _hip.ptr(linspace01(n_imp, dev)),
                                                  _hip.ptr(None if t_rand is None else _c(t_rand)), 1 if perturb else 0, seed_v,
                                                  _hip.ptr(state), _hip.ptr(scratch), _hip.ptr(acc), _hip.stream_handle(dev)),
                   "eslam_loss_set_sizes")
    return acc


def _take_rng_bump(dev):
    """Pointer for the forward kernel's rng_bump if the samples it renders came from the in-kernel generator."""
    if _rng_pending.pop(dev.index, False):
        return _hip.ptr(_rng_state(dev))
    return None


class RenderFn(torch.autograd.Function):
    """depth, rgb, sdf[, loss] = RenderFn.apply(rays_o, rays_d, z_vals, bound6, beta, order, lossctx, *12 planes,
                                                *12 decoder params)

    Forward = eslam_render_fwd, backward = eslam_render_bwd (reference: Renderer.py:136-147 and its autograd).
    order = (perm, stream) from ray_order_async, or None.
    lossctx = an ops.fused_loss context or None.  With it the forward is eslam_render_fwd_loss, a fourth output `loss`
    (0-d) is returned, and the gradient arriving on it is turned into the rendered outputs' gradients inside the backward
    kernel (eslam_render_bwd_loss); gradients arriving on depth / rgb / sdf themselves are added as usual."""

    N_LEAD = 7           # inputs before the planes

    @staticmethod
    def forward(ctx, rays_o, rays_d, z_vals, bound6, beta, order_in, lossctx, *tensors):
        planes, params = tensors[:12], tensors[12:24]
        for n, t in (("rays_o", rays_o), ("rays_d", rays_d), ("z_vals", z_vals)):
            _hip.require_gpu_f32(n, t)
        rays_o, rays_d, z_vals = _c(rays_o.detach()), _c(rays_d.detach()), _c(z_vals.detach())
        R, S = z_vals.shape
        dev = rays_o.device
        lib = _hip.lib()
        half = None if _half_planes is None else list(_half_planes.flat)
        ctx.half = half
        arr, _ = _hip.make_planes(planes, half=half)
        dec, keep = _hip.make_decoders(params, beta)
        needs = any(ctx.needs_input_grad)
        if half is not None and (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]):
            raise RuntimeError("mixed precision: gradients with respect to the rays (pose) are not built; detach the rays")
        depth = torch.empty(R, device=dev)
        rgb = torch.empty(R, 3, device=dev)
        sdf = torch.empty(R, S, device=dev)
        raw_rgb = torch.empty(R, S, 3, device=dev) if needs else None
        # saved features: float32, or (mixed precision) the bf16 values the decoders consumed - half the bytes
        feat = torch.empty(R * S, 128, device=dev, dtype=torch.float32 if half is None else torch.bfloat16) if needs else None
        order = None
        ctx.pregrads = None
        if order_in is not None:
            order, side = order_in[0], order_in[1]
            if len(order_in) > 2 and needs:
                ctx.pregrads = order_in[2]          # cleared on the side stream (ray_order_async), joined below
            if _FWD_USES_ORDER:
                _hip.stream_wait(dev, None, side)
                side = None
            # otherwise only the backward needs the order: the ordering kernel (side stream) is joined there, and the
            # forward kernel starts as soon as the samplers are done
        fl = lossctx
        ctx.set_materialize_grads(False)
        ctx.lossctx = None
        with _hip.on_device(dev):
            if fl is None:
                _hip.check(lib.eslam_render_fwd(arr, ctypes.byref(dec), _hip.make_bound(bound6), _hip.ptr(rays_o),
                                                _hip.ptr(rays_d), _hip.ptr(z_vals), R, S, _hip.ptr(depth), _hip.ptr(rgb),
                                                _hip.ptr(sdf), _hip.ptr(raw_rgb), _hip.ptr(feat),
                                                _hip.ptr(order) if _FWD_USES_ORDER else None, _take_rng_bump(dev),
                                                _hip.stream_handle(dev)), "eslam_render_fwd")
            else:
                _hip.require_gpu_f32("gt_depth", fl.gt_depth)
                _hip.require_gpu_f32("gt_color", fl.gt_color)
                mask = fl.ray_mask
                if mask is not None:
                    mask = _c(mask.view(torch.uint8) if mask.dtype == torch.bool else mask.to(torch.uint8))
                fl.acc = torch.empty(16, device=dev) if getattr(fl, "acc_out", None) is None else fl.acc_out
                fl.value = torch.empty((), device=dev)
                st = fl.state
                st.gt_depth, st.gt_color, st.ray_mask, st.acc, st.value = _c(fl.gt_depth), _c(fl.gt_color), mask, fl.acc, fl.value
                w5 = (ctypes.c_float * 5)(*st.weights5)
                _hip.check(lib.eslam_render_fwd_loss(arr, ctypes.byref(dec), _hip.make_bound(bound6), _hip.ptr(rays_o),
                                                     _hip.ptr(rays_d), _hip.ptr(z_vals), R, S, _hip.ptr(depth),
                                                     _hip.ptr(rgb), _hip.ptr(sdf), _hip.ptr(raw_rgb), _hip.ptr(feat),
                                                     _hip.ptr(order) if _FWD_USES_ORDER else None,
                                                     _hip.ptr(st.gt_depth), _hip.ptr(st.gt_color),
                                                     st.truncation, w5, _hip.ptr(mask),
                                                     _hip.ptr(_loss_scratch(dev, R)), _hip.ptr(fl.acc), _hip.ptr(fl.value),
                                                     _take_rng_bump(dev), _hip.stream_handle(dev)), "eslam_render_fwd_loss")
                ctx.lossctx = st
        if order_in is not None and side is not None:
            # join the side stream (ray order, gradient clear) BEHIND the forward kernel: the work beside it has overlapped,
            # and no fork is left dangling if this forward is never followed by a backward (or sits in a graph of its own)
            _hip.stream_wait(dev, None, side)
            side = None
        if needs:
            ctx.bound6 = bound6
            ctx.order_stream = None
            ctx.save_for_backward(rays_o, rays_d, z_vals, sdf, raw_rgb, feat, order, beta, depth, rgb, *planes, *params)
        if fl is not None:
            return depth, rgb, sdf, fl.value.detach()     # an alias: the state must not hold the output object itself
        return depth, rgb, sdf

    @staticmethod
    def backward(ctx, g_depth, g_rgb, g_sdf, g_loss=None):
        saved = ctx.saved_tensors
        rays_o, rays_d, z_vals, sdf, raw_rgb, feat, order, beta, depth, rgb = saved[:10]
        planes, params = saved[10:22], saved[22:34]
        R, S = z_vals.shape
        dev = rays_o.device
        lib = _hip.lib()
        if getattr(ctx, "order_stream", None) is not None:
            _hip.stream_wait(dev, None, ctx.order_stream)      # join the ordering kernel
            ctx.order_stream = None
        need = ctx.needs_input_grad
        L = RenderFn.N_LEAD
        need_planes = any(need[L:L + 12])
        need_rays = need[0] or need[1]
        grads = None
        sink = _grad_sink
        if sink is not None:
            sink.zero_()
            grads = sink.views[:12]
        elif need_planes:
            grads = getattr(ctx, "pregrads", None)
            ctx.pregrads = None
            if grads is None:
                _, grads = _alloc_plane_grads(planes)
        arr, _ = _hip.make_planes(planes, grads, half=ctx.half)
        dec, keep = _hip.make_decoders(params, beta)
        if sink is not None:
            # the 12 decoder tensors follow the planes in the flat buffer in C-ABI order = the layout of g_dec
            o = sink.offsets[12]
            g_dec = sink.flat[o:o + _hip.N_DEC_PARAMS]
            g_beta = sink.flat[sink.offsets[24]:sink.offsets[24] + 1] if len(sink.views) > 24 else \
                torch.empty(1, device=dev)
        else:
            need_dec = any(need[L + 12:L + 24])
            g_dec = torch.empty(_hip.N_DEC_PARAMS, device=dev) if need_dec else None
            g_beta = torch.empty(1, device=dev) if need[4] else None
        g_ro = torch.empty(R, 3, device=dev) if need_rays else None
        g_rd = torch.empty(R, 3, device=dev) if need_rays else None
        if R == 0:        # the C entry returns at once for an empty batch: the gradient of nothing is zero, not uninitialised
            for t in (g_dec, g_beta):
                if t is not None:
                    t.zero_()
        ws = torch.empty(lib.eslam_bwd_workspace_bytes(R * S), dtype=torch.uint8, device=dev)
        g_depth, g_rgb, g_sdf = _c(g_depth), _c(g_rgb), _c(g_sdf)
        fl = ctx.lossctx                 # a _LossState
        with _hip.on_device(dev):
            if fl is not None and g_loss is not None:
                # the loss's gradients are formed inside the backward kernel from the set sizes in acc (a ray-sharded caller
                # has put the global ones into acc_global by now)
                up = _c(g_loss.detach().reshape(1).to(torch.float32))
                w5 = (ctypes.c_float * 5)(*fl.weights5)
                _hip.check(lib.eslam_render_bwd_loss(
                    arr, ctypes.byref(dec), _hip.make_bound(ctx.bound6), _hip.ptr(rays_o), _hip.ptr(rays_d),
                    _hip.ptr(z_vals), R, S, _hip.ptr(sdf), _hip.ptr(raw_rgb), _hip.ptr(feat), _hip.ptr(depth), _hip.ptr(rgb),
                    _hip.ptr(fl.gt_depth), _hip.ptr(fl.gt_color), fl.truncation, w5, _hip.ptr(fl.ray_mask),
                    _hip.ptr(fl.acc if fl.acc_global is None else fl.acc_global), _hip.ptr(up), _hip.ptr(fl.value) if fl.rewrite_value else None, _hip.ptr(g_depth),
                    _hip.ptr(g_rgb), _hip.ptr(g_sdf), _hip.ptr(g_dec), _hip.ptr(g_beta), _hip.ptr(g_ro), _hip.ptr(g_rd),
                    _hip.ptr(order), _hip.ptr(ws), _hip.stream_handle(dev)), "eslam_render_bwd_loss")
            else:
                _hip.check(lib.eslam_render_bwd(arr, ctypes.byref(dec), _hip.make_bound(ctx.bound6), _hip.ptr(rays_o),
                                                _hip.ptr(rays_d), _hip.ptr(z_vals), R, S, _hip.ptr(sdf), _hip.ptr(raw_rgb),
                                                _hip.ptr(feat), _hip.ptr(g_depth), _hip.ptr(g_rgb), _hip.ptr(g_sdf),
                                                _hip.ptr(g_dec), _hip.ptr(g_beta), _hip.ptr(g_ro), _hip.ptr(g_rd),
                                                _hip.ptr(order), _hip.ptr(ws), _hip.stream_handle(dev)),
                           "eslam_render_bwd")
        if sink is not None:
            # the data-parallel caller owns .grad assignment (FlatGrads.assign): hand autograd nothing to accumulate
            return (g_ro if need[0] else None, g_rd if need[1] else None) + (None,) * (L - 2 + 24)
        dec_grads = _split_dec_grads(g_dec) if g_dec is not None else [None] * 12
        out = [g_ro if need[0] else None, g_rd if need[1] else None, None, None, g_beta if need[4] else None, None, None]
        out += [grads[i] if (need_planes and need[L + i]) else None for i in range(12)]
        out += [dec_grads[i] if need[L + 12 + i] else None for i in range(12)]
        return tuple(out)


class DecodeFn(torch.autograd.Function):
    """raw[N,4] = DecodeFn.apply(pts[N,3], bound6, *12 planes, *12 decoder params)   (decoders.py:127-146)"""

    @staticmethod
    def forward(ctx, pts, bound6, dummy_beta, *tensors):
        planes, params = tensors[:12], tensors[12:24]
        _hip.require_gpu_f32("p", pts)
        pts = _c(pts.detach())
        N = pts.shape[0]
        dev = pts.device
        lib = _hip.lib()
        arr, _ = _hip.make_planes(planes)
        dec, keep = _hip.make_decoders(params, dummy_beta)
        needs = any(ctx.needs_input_grad)
        raw = torch.empty(N, 4, device=dev)
        feat = torch.empty(N, 128, device=dev) if needs else None
        with _hip.on_device(dev):
            _hip.check(lib.eslam_decode_fwd(arr, ctypes.byref(dec), _hip.make_bound(bound6), _hip.ptr(pts), N, 0,
                                            _hip.ptr(raw), _hip.ptr(feat), _hip.stream_handle(dev)), "eslam_decode_fwd")
        if needs:
            ctx.bound6 = bound6
            ctx.save_for_backward(pts, raw, feat, dummy_beta, *planes, *params)
        return raw

    @staticmethod
    def backward(ctx, g_raw):
        saved = ctx.saved_tensors
        pts, raw, feat, dummy_beta = saved[:4]
        planes, params = saved[4:16], saved[16:28]
        N = pts.shape[0]
        dev = pts.device
        lib = _hip.lib()
        need = ctx.needs_input_grad
        need_planes = any(need[3:15])
        grads = None
        if need_planes:
            _, grads = _alloc_plane_grads(planes)
        arr, _ = _hip.make_planes(planes, grads)
        dec, keep = _hip.make_decoders(params, dummy_beta)
        g_dec = torch.empty(_hip.N_DEC_PARAMS, device=dev)
        g_pts = torch.empty(N, 3, device=dev) if need[0] else None
        ws = torch.empty(lib.eslam_bwd_workspace_bytes(N), dtype=torch.uint8, device=dev)
        g_raw = _c(g_raw)
        with _hip.on_device(dev):
            _hip.check(lib.eslam_decode_bwd(arr, ctypes.byref(dec), _hip.make_bound(ctx.bound6), _hip.ptr(pts), N,
                                            _hip.ptr(raw), _hip.ptr(feat), _hip.ptr(g_raw), _hip.ptr(g_dec),
                                            _hip.ptr(g_pts), _hip.ptr(ws), _hip.stream_handle(dev)), "eslam_decode_bwd")
        dec_grads = _split_dec_grads(g_dec)
        out = [g_pts, None, None]
        out += [grads[i] if (need_planes and need[3 + i]) else None for i in range(12)]
        out += [dec_grads[i] if need[15 + i] else None for i in range(12)]
        return tuple(out)


def decode_sdf_only(pts, bound6, all_planes, decoders):
    """Geometry planes + SDF decoder only, no autograd (reference decoders.py:87-105 under no_grad)."""
    _hip.require_gpu_f32("p", pts)
    pts = _c(pts.detach())
    N = pts.shape[0]
    dev = pts.device
    lib = _hip.lib()
    geo = tuple(all_planes[:3]) + tuple(all_planes[:3])
    arr, _ = _hip.make_planes(geo)
    dec, keep = _hip.make_decoders(decoder_params(decoders), beta_tensor(10, dev))
    out = torch.empty(N, device=dev)
    with _hip.on_device(dev):
        _hip.check(lib.eslam_decode_fwd(arr, ctypes.byref(dec), _hip.make_bound(bound6), _hip.ptr(pts), N, 1,
                                        _hip.ptr(out), None, _hip.stream_handle(dev)), "eslam_decode_fwd(sdf)")
    return out


class SampleRaysFn(torch.autograd.Function):
    """rays_o, rays_d, depth, color = SampleRaysFn.apply(c2ws, indices, depths, colors, geom)   (common.py:87-153)"""

    @staticmethod
    def forward(ctx, c2ws, indices, depths, colors, geom):
        H0, H1, W0, W1, n, H, W, fx, fy, cx, cy = geom
        _hip.require_gpu_f32("c2ws", c2ws)
        _hip.require_gpu_f32("depths", depths)
        _hip.require_gpu_f32("colors", colors)
        c2 = _c(c2ws.detach())
        b = c2.shape[0]
        dev = c2.device
        depths, colors, indices = _c(depths), _c(colors), _c(indices)
        if depths.shape != (b, H, W) or colors.shape != (b, H, W, 3):
            raise RuntimeError(f"get_samples: depths {tuple(depths.shape)} / colors {tuple(colors.shape)} do not match "
                               f"b={b}, H={H}, W={W}")
        if indices.dtype != torch.int64 or indices.numel() != b * n:
            raise RuntimeError("get_samples: indices must be int64 [b*n]")
        ro = torch.empty(b * n, 3, device=dev)
        rd = torch.empty(b * n, 3, device=dev)
        d = torch.empty(b * n, device=dev)
        c = torch.empty(b * n, 3, device=dev)
        with _hip.on_device(dev):
            _hip.check(_hip.lib().eslam_sample_rays(_hip.ptr(indices), b, n, H0, H1, W0, W1, H, W, fx, fy, cx, cy,
                                                    _hip.ptr(c2), _hip.ptr(depths), _hip.ptr(colors), _hip.ptr(ro),
                                                    _hip.ptr(rd), _hip.ptr(d), _hip.ptr(c), _hip.stream_handle(dev)),
                       "eslam_sample_rays")
        ctx.geom = geom
        ctx.b = b
        ctx.save_for_backward(indices)
        ctx.mark_non_differentiable(d, c)
        return ro, rd, d, c

    @staticmethod
    def backward(ctx, g_ro, g_rd, _gd, _gc):
        (indices,) = ctx.saved_tensors
        H0, H1, W0, W1, n, H, W, fx, fy, cx, cy = ctx.geom
        dev = indices.device
        g = torch.empty(ctx.b, 4, 4, device=dev)
        g_ro, g_rd = _c(g_ro), _c(g_rd)
        with _hip.on_device(dev):
            _hip.check(_hip.lib().eslam_sample_rays_bwd(_hip.ptr(indices), ctx.b, n, H0, W0, W1, fx, fy, cx, cy,
                                                        _hip.ptr(g_ro), _hip.ptr(g_rd), _hip.ptr(g),
                                                        _hip.stream_handle(dev)), "eslam_sample_rays_bwd")
        return g, None, None, None, None


def image_rays(H, W, fx, fy, cx, cy, c2w):
    _hip.require_gpu_f32("c2w", c2w)
    c2 = _c(c2w.detach())
    dev = c2.device
    ro = torch.empty(H * W, 3, device=dev)
    rd = torch.empty(H * W, 3, device=dev)
    with _hip.on_device(dev):
        _hip.check(_hip.lib().eslam_image_rays(H, W, fx, fy, cx, cy, _hip.ptr(c2), _hip.ptr(ro), _hip.ptr(rd),
                                               _hip.stream_handle(dev)), "eslam_image_rays")
    return ro, rd


def aabb_exit(rays_o, rays_d, bound6):
    _hip.require_gpu_f32("rays_o", rays_o)
    ro, rd = _c(rays_o.detach()), _c(rays_d.detach())
    R = ro.shape[0]
    t = torch.empty(R, device=ro.device)
    with _hip.on_device(ro.device):
        _hip.check(_hip.lib().eslam_aabb_exit(_hip.ptr(ro), _hip.ptr(rd), R, _hip.make_bound(bound6), _hip.ptr(t),
                                              _hip.stream_handle(ro.device)), "eslam_aabb_exit")
    return t


def prefilter(rays_o, rays_d, gt_depth, bound6, need_depth):
    """bool [R]: the callers' AABB (+ depth) pre-filter as a mask (Mapper.py:322-328 / Tracker.py:175-182), one launch."""
    _hip.require_gpu_f32("rays_o", rays_o)
    ro, rd, gd = _c(rays_o.detach()), _c(rays_d.detach()), _c(gt_depth)
    R = ro.shape[0]
    keep = torch.empty(R, dtype=torch.uint8, device=ro.device)
    with _hip.on_device(ro.device):
        _hip.check(_hip.lib().eslam_prefilter(_hip.ptr(ro), _hip.ptr(rd), _hip.ptr(gd), R, _hip.make_bound(bound6),
                                              1 if need_depth else 0, _hip.ptr(keep), _hip.stream_handle(ro.device)),
                   "eslam_prefilter")
    return keep.view(torch.bool)


TRACKING_MASK_MAX = 8192        # ESLAM_TRACKING_MASK_MAX


def tracking_mask(depth, gt_depth, keep=None, factor=10.0):
    """bool [R]: Tracker.py:192-195, |gt_depth - depth| < factor * median over the kept rays, one launch."""
    _hip.require_gpu_f32("depth", depth)
    d, gd = _c(depth.detach()), _c(gt_depth)
    R = d.shape[0]
    if keep is not None:
        keep = _c(keep.view(torch.uint8) if keep.dtype == torch.bool else keep.to(torch.uint8))
    mask = torch.empty(R, dtype=torch.uint8, device=d.device)
    with _hip.on_device(d.device):
        _hip.check(_hip.lib().eslam_tracking_mask(_hip.ptr(d), _hip.ptr(gd), _hip.ptr(keep), R, float(factor),
                                                  _hip.ptr(mask), _hip.stream_handle(d.device)), "eslam_tracking_mask")
    return mask.view(torch.bool)


def keep_best(loss, pose, best, best_pose):
    """In place: if loss < best, best = loss and best_pose = pose (Tracker.py:304-307), without a host round trip."""
    _hip.require_gpu_f32("loss", loss)
    pose = _c(pose.detach())
    with _hip.on_device(loss.device):
        _hip.check(_hip.lib().eslam_keep_best(_hip.ptr(loss.detach()), _hip.ptr(pose), pose.numel(), _hip.ptr(best),
                                              _hip.ptr(best_pose), _hip.stream_handle(loss.device)), "eslam_keep_best")


class PoseToMatrixFn(torch.autograd.Function):
    """c2ws [b,4,4] = PoseToMatrixFn.apply(poses [b,7])   (common.py:169-181, quaternion real-first then translation)"""

    @staticmethod
    def forward(ctx, poses):
        _hip.require_gpu_f32("batch_poses", poses)
        p = _c(poses.detach())
        b = p.shape[0]
        out = torch.empty(b, 4, 4, device=p.device)
        with _hip.on_device(p.device):
            _hip.check(_hip.lib().eslam_pose_to_c2w(_hip.ptr(p), b, _hip.ptr(out), _hip.stream_handle(p.device)),
                       "eslam_pose_to_c2w")
        ctx.save_for_backward(p)
        return out

    @staticmethod
    def backward(ctx, g):
        (p,) = ctx.saved_tensors
        g = _c(g)
        gp = torch.empty_like(p)
        with _hip.on_device(p.device):
            _hip.check(_hip.lib().eslam_pose_to_c2w_bwd(_hip.ptr(p), _hip.ptr(g), p.shape[0], _hip.ptr(gp),
                                                        _hip.stream_handle(p.device)), "eslam_pose_to_c2w_bwd")
        return gp


def sample_z(rays_o, rays_d, gt_depth, all_planes, decoders, bound6, truncation, n_strat, n_imp, perturb,
             rand=None):
    """z_vals [R,S] of reference Renderer.py:85-134, sync-free: both samplers run over all rays and each skips
    the rays that belong to the other.  rand = (t_rand [R,S], t_rand_uni [R,n_strat], u [R,n_imp]) or None to
    draw them with torch.rand."""
    _hip.require_gpu_f32("gt_depth", gt_depth)
    dev = rays_o.device
    gd = _c(gt_depth.detach().reshape(-1))
    R = gd.shape[0]
    S = n_strat + n_imp
    lib = _hip.lib()
    if rand is None and _USE_KERNEL_RNG and n_strat >= 3:
        # the uniform numbers are drawn inside the sampler kernel, keyed on torch's seed and a device step counter
        z = torch.empty(R, S, device=dev)
        t_free, t_surf = linspace01(n_strat, dev), linspace01(n_imp, dev)
        seed_v = _rng_seed(dev)
        state = _rng_state(dev) if _rng_override is None else _rng_override
        if _rng_override is None and _rng_pending.get(dev.index, False):
            state[0] += 1              # the previous samples were never rendered: advance the step here
        ro, rd = _c(rays_o.detach()), _c(rays_d.detach())
        arr, _ = _hip.make_planes(all_planes)
        dec, keep = _hip.make_decoders(decoder_params(decoders), beta_tensor(decoders.beta, dev))
        with _hip.on_device(dev):
            _hip.check(lib.eslam_sample_z_all_rng(arr, ctypes.byref(dec), _hip.make_bound(bound6), _hip.ptr(ro), _hip.ptr(rd),
                                                  _hip.ptr(gd), R, n_strat, n_imp, float(truncation), _hip.ptr(t_free),
                                                  _hip.ptr(t_surf), 1 if perturb else 0, seed_v, _hip.ptr(state),
                                                  _ray_offset, _hip.ptr(z), _hip.stream_handle(dev)), "eslam_sample_z_all_rng")
        _rng_pending[dev.index] = _rng_override is None
        return z
    if rand is None:
        # one draw, three row-major blocks (the reference draws them in three calls, Renderer.py:59, common.py:59)
        n1, n2 = (R * S, R * n_strat) if perturb else (0, 0)
        pool = torch.rand(n1 + n2 + R * n_imp, device=dev)
        t_rand = pool[:n1].view(R, S) if perturb else None
        t_uni = pool[n1:n1 + n2].view(R, n_strat) if perturb else None
        u = pool[n1 + n2:].view(R, n_imp)
    else:
        t_rand, t_uni, u = (None if t is None else _c(t) for t in rand)
    z = torch.empty(R, S, device=dev)
    t_free, t_surf = linspace01(n_strat, dev), linspace01(n_imp, dev)
    with _hip.on_device(dev):
        st = _hip.stream_handle(dev)
        if u is not None and n_strat >= 3:
            # both samplers in one launch (each wave takes the rule its ray needs)
            ro, rd = _c(rays_o.detach()), _c(rays_d.detach())
            arr, _ = _hip.make_planes(all_planes)
            dec, keep = _hip.make_decoders(decoder_params(decoders), beta_tensor(decoders.beta, dev))
            _hip.check(lib.eslam_sample_z_all(arr, ctypes.byref(dec), _hip.make_bound(bound6), _hip.ptr(ro), _hip.ptr(rd),
                                              _hip.ptr(gd), R, n_strat, n_imp, float(truncation), _hip.ptr(t_free),
                                              _hip.ptr(t_surf), _hip.ptr(t_rand), _hip.ptr(t_uni), _hip.ptr(u),
                                              _hip.ptr(z), st), "eslam_sample_z_all")
        else:
            _hip.check(lib.eslam_sample_z(_hip.ptr(gd), R, n_strat, n_imp, float(truncation), _hip.ptr(t_free),
                                          _hip.ptr(t_surf), _hip.ptr(t_rand), _hip.ptr(z), st), "eslam_sample_z")
            if u is not None:        # fewer than 3 stratified samples: the importance sampler reports it as unsupported
                ro, rd = _c(rays_o.detach()), _c(rays_d.detach())
                arr, _ = _hip.make_planes(all_planes)
                dec, keep = _hip.make_decoders(decoder_params(decoders), beta_tensor(decoders.beta, dev))
                _hip.check(lib.eslam_importance_z(arr, ctypes.byref(dec), _hip.make_bound(bound6), _hip.ptr(ro),
                                                  _hip.ptr(rd), _hip.ptr(gd), R, n_strat, n_imp, _hip.ptr(t_free),
                                                  _hip.ptr(t_uni), _hip.ptr(u), _hip.ptr(z), st), "eslam_importance_z")
    return z


def loss_reduce(depth, rgb, sdf, z_vals, gt_depth, gt_color, truncation, ray_mask=None, acc=None):
    """Phase 1 of the fused loss (eslam_loss_reduce): set sizes and squared-error sums of this rank's rays accumulated
    into acc [16] (zeroed here unless given)."""
    for n, t in (("depth", depth), ("rgb", rgb), ("sdf", sdf), ("z_vals", z_vals), ("gt_depth", gt_depth),
                 ("gt_color", gt_color)):
        _hip.require_gpu_f32(n, t)
    dev = depth.device
    R, S = sdf.shape
    args = [_c(t.detach()) for t in (depth, rgb, sdf, z_vals, gt_depth, gt_color)]
    if acc is None:
        acc = torch.zeros(16, device=dev)
    else:
        acc.zero_()
    with _hip.on_device(dev):
        _hip.check(_hip.lib().eslam_loss_reduce(*[_hip.ptr(t) for t in args], R, S, float(truncation), _hip.ptr(ray_mask),
                                                _hip.ptr(acc), _hip.stream_handle(dev)), "eslam_loss_reduce")
    return acc


_scratch_slots = {}           # (device index, stream handle) -> scratch tensor
_scratch_pool = {}            # device index -> [pre-zeroed pool tensor, floats handed out]
_DEBUG_SCRATCH = os.environ.get("ESLAM_DEBUG_SCRATCH", "0") == "1"
_SCRATCH_POOL_FLOATS = 1 << 20        # 4 MB per device: 8 streams of 32 768-ray deterministic scratch, or ~1900 plain ones


def _loss_scratch(dev, n_rays=0):
    """The scratch of the fused loss's ticket scheme (eslam_loss_value / eslam_render_fwd_loss): zeroed once, then owned by
    the kernels, which leave it zeroed again.  ONE PER (device, stream): two streams of a process evaluating losses at the
    same time must not share the running sums.  Slots come out of a pool that is zeroed when it is created - outside any
    graph capture, by the warm-up iterations - because a buffer created inside a capture would be re-zeroed by a captured
    fill on every replay.  (torch captures every graph on one shared capture stream: graphs that are to REPLAY concurrently
    on different streams need their own slot - capture them under ops.fresh_loss_scratch().)
    ESLAM_DEBUG_SCRATCH=1 checks on the host that the ticket counter is 0 before every use (a graph aborted between the
    adds and the last ticket leaves it non-zero): one sync per loss, for debugging only."""
    t = _pooled_scratch(dev, int(_hip.lib().eslam_loss_scratch_floats(int(n_rays))), "loss")
    if _DEBUG_SCRATCH and not torch.cuda.is_current_stream_capturing():
        if int(t[:1].view(torch.int32).item()) != 0:
            reset_loss_scratch(dev)
            raise RuntimeError("loss scratch: the ticket counter was not 0 on entry - a previous loss evaluation on this stream "
                               "did not finish (aborted graph?) or two streams shared a scratch; it has been reset")
    return t


def _pooled_scratch(dev, need, tag):
    """`need` floats of scratch that were zero when handed out for the first time, one slot per (device, stream, tag)."""
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream, _scratch_epoch, tag)
    t = _scratch_slots.get(key)
    if t is None or t.numel() < need:
        pool = _scratch_pool.get(dev.index)
        if pool is None or pool[1] + need > pool[0].numel():
            pool = _scratch_pool[dev.index] = [torch.zeros(max(_SCRATCH_POOL_FLOATS, need), device=dev), 0]
        t = pool[0][pool[1]:pool[1] + need]
        pool[1] += (need + 31) // 32 * 32
        _scratch_slots[key] = t
    return t


_scratch_epoch = 0


class fresh_loss_scratch:
    """Context manager: loss evaluations issued inside get scratch slots of their own (for a graph that will replay
    concurrently with other graphs captured on torch's shared capture stream)."""

    def __enter__(self):
        global _scratch_epoch
        self._prev = _scratch_epoch
        _scratch_epoch = fresh_loss_scratch._next = getattr(fresh_loss_scratch, "_next", 0) + 1
        return self

    def __exit__(self, *a):
        global _scratch_epoch
        _scratch_epoch = self._prev


def reset_loss_scratch(device=None):
    """eslam_loss_scratch_reset on every scratch slot (of one device): back to the freshly zeroed state."""
    for (di, _, _, tag), t in _scratch_slots.items():
        if tag == "loss" and (device is None or torch.device(device).index == di):
            with _hip.on_device(t.device):
                _hip.check(_hip.lib().eslam_loss_scratch_reset(_hip.ptr(t), t.numel(), _hip.stream_handle(t.device)),
                           "eslam_loss_scratch_reset")


class MappingLossFn(torch.autograd.Function):
    """loss = MappingLossFn.apply(depth, rgb, sdf, z_vals, gt_depth, gt_color, truncation, weights5, ray_mask, group, acc)

    Fused restatement of Mapper.py:110-144,337-346 (ray_mask None) / Tracker.py:114-148,197-204 (ray_mask given).
    forward = eslam_loss_reduce (+ the loss value); backward = eslam_loss_grad scaled by the upstream gradient inside the
    kernel.  `group`: a torch.distributed process group (or True for the default group) makes the set sizes and error
    sums global with one 16-float all-reduce, for ray-sharded data parallelism (myslam_amd/parallel.py).
    `acc`: accumulators that are already reduced (loss_reduce + the caller's own all-reduce); skips phase 1."""

    @staticmethod
    def forward(ctx, depth, rgb, sdf, z_vals, gt_depth, gt_color, truncation, weights5, ray_mask, group, acc):
        if ray_mask is not None:
            ray_mask = _c(ray_mask.view(torch.uint8) if ray_mask.dtype == torch.bool else ray_mask.to(torch.uint8))
        for n, t in (("depth", depth), ("rgb", rgb), ("sdf", sdf), ("z_vals", z_vals), ("gt_depth", gt_depth),
                     ("gt_color", gt_color)):
            _hip.require_gpu_f32(n, t)
        dev = depth.device
        R, S = sdf.shape
        args = [_c(t.detach()) for t in (depth, rgb, sdf, z_vals, gt_depth, gt_color)]
        loss = torch.empty(1, device=dev)
        w = (ctypes.c_float * 5)(*[float(v) for v in weights5])
        if isinstance(acc, tuple):
            acc, value = acc                     # both already formed by the forward kernel (ops.fused_loss)
            loss = value
        elif acc is None and group is None:
            # single GPU: sums, set sizes and the value in one launch, no pre-zeroed accumulator
            acc = torch.empty(16, device=dev)
            with _hip.on_device(dev):
                _hip.check(_hip.lib().eslam_loss_value(*[_hip.ptr(t) for t in args], R, S, float(truncation), w,
                                                       _hip.ptr(ray_mask), _hip.ptr(_loss_scratch(dev, R)), _hip.ptr(acc),
                                                       _hip.ptr(loss), _hip.stream_handle(dev)), "eslam_loss_value")
        else:
            if acc is None:
                acc = loss_reduce(depth, rgb, sdf, z_vals, gt_depth, gt_color, truncation, ray_mask)
                import torch.distributed as dist
                dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=None if group is True else group)
            with _hip.on_device(dev):
                _hip.check(_hip.lib().eslam_loss_grad(*[_hip.ptr(t) for t in args], R, S, float(truncation), w,
                                                      _hip.ptr(ray_mask), _hip.ptr(acc), _hip.ptr(loss), None, None, None,
                                                      None, _hip.stream_handle(dev)), "eslam_loss_grad(value)")
        ctx.save_for_backward(*args, acc)
        ctx.ray_mask = ray_mask
        ctx.consts = (float(truncation), tuple(float(v) for v in weights5))
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        *args, acc = ctx.saved_tensors
        dev = acc.device
        R, S = args[2].shape
        truncation, weights5 = ctx.consts
        g_depth = torch.empty(R, device=dev)
        g_rgb = torch.empty(R, 3, device=dev)
        g_sdf = torch.empty(R, S, device=dev)
        w = (ctypes.c_float * 5)(*weights5)
        g = _c(g.detach().reshape(1).to(torch.float32))
        with _hip.on_device(dev):
            _hip.check(_hip.lib().eslam_loss_grad(*[_hip.ptr(t) for t in args], R, S, truncation, w,
                                                  _hip.ptr(ctx.ray_mask), _hip.ptr(acc), None, _hip.ptr(g_depth),
                                                  _hip.ptr(g_rgb), _hip.ptr(g_sdf), _hip.ptr(g),
                                                  _hip.stream_handle(dev)), "eslam_loss_grad")
        return g_depth, g_rgb, g_sdf, None, None, None, None, None, None, None, None
