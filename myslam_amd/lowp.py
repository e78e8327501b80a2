"""Mixed-precision inference path (BASELINE.json configs[4]: fp16 planes + bf16 MFMA decoders, a tolerance study).

The planes are kept as float32 masters (the optimiser's copy, exactly as in the reference) and converted to
channels-last float16 copies for rendering; the decoders' float32 weights are rounded to bf16 inside the kernel.
Sampling (z_vals) is the float32 path.  Forward only: there is no low-precision backward.
"""
import ctypes

import torch

from . import _hip, ops


def half_planes(all_planes):
    """float16, channels-last copies of the 12 planes (one texel = 64 contiguous bytes)."""
    return tuple([p.detach().to(torch.float16).contiguous(memory_format=torch.channels_last) for p in grp]
                 for grp in all_planes)


def render_batch_ray_lowp(renderer, all_planes, planes_f16, decoders, rays_d, rays_o, truncation, gt_depth, _rand=None):
    """Same contract as Renderer.render_batch_ray (reference src/utils/Renderer.py:63-147) without autograd:
    returns depth [R], rgb [R,3], sdf [R,S], z_vals [R,S]."""
    _hip.require_gpu_f32("rays_o", rays_o)
    dev = rays_o.device
    with torch.no_grad():
        z_vals = ops.sample_z(rays_o, rays_d, gt_depth, all_planes, decoders, renderer._bound6, truncation,
                              renderer.n_stratified, renderer.n_importance, renderer.perturb, _rand)
        R, S = z_vals.shape
        arr, keep = _hip.make_planes(planes_f16, dtype=torch.float16)
        dec, keep2 = _hip.make_decoders([p.detach() for p in ops.decoder_params(decoders)],
                                        ops.beta_tensor(decoders.beta, dev).detach())
        depth = torch.empty(R, device=dev)
        rgb = torch.empty(R, 3, device=dev)
        sdf = torch.empty(R, S, device=dev)
        ro, rd = rays_o.detach().contiguous(), rays_d.detach().contiguous()
        with _hip.on_device(dev):
            _hip.check(_hip.lib().eslam_render_fwd_lowp(arr, ctypes.byref(dec), _hip.make_bound(ops.bound_to_host(decoders.bound)),
                                                        _hip.ptr(ro), _hip.ptr(rd), _hip.ptr(z_vals), R, S, _hip.ptr(depth),
                                                        _hip.ptr(rgb), _hip.ptr(sdf), _hip.stream_handle(dev)),
                       "eslam_render_fwd_lowp")
    return depth, rgb, sdf, z_vals
