"""Mixed precision (BASELINE.json configs[4]: fp16 planes + bf16 MFMA decoders), forward AND backward.

The planes stay float32 masters (the optimiser's copy, exactly as in the reference); HalfPlanes holds their channels-last
float16 copies and refreshes all 12 in one launch (eslam_planes_to_half) after an optimiser step.  Under
`ops.mixed_precision(half)` the renderer's kernels gather texels from the copies (float32 accumulation), run both decoders on
bf16 MFMA with float32 accumulation - in the backward pass too - and accumulate plane gradients in float32 for the masters.
Sampling (z_vals), the loss, the composite and its backward, the scatter and the optimiser are the float32 path.
"""
import ctypes

import torch

from . import _hip, ops


class HalfPlanes:
    def __init__(self, all_planes):
        for grp in all_planes:
            for p in grp:
                if not (p.is_cuda and p.dtype == torch.float32 and p.dim() == 4 and
                        p.is_contiguous(memory_format=torch.channels_last)):
                    raise RuntimeError("mixed precision needs float32 channels_last planes on the GPU")
        self.planes = tuple([torch.empty_like(p.detach(), dtype=torch.float16) for p in grp] for grp in all_planes)
        self.flat = [p for grp in self.planes for p in grp]
        self.refresh(all_planes)

    def refresh(self, all_planes):
        """float32 masters -> float16 copies, all 12 planes in one launch."""
        arr, _ = _hip.make_planes(tuple([p.detach() for p in grp] for grp in all_planes), half=self.flat)
        dev = self.flat[0].device
        with _hip.on_device(dev):
            _hip.check(_hip.lib().eslam_planes_to_half(arr, _hip.stream_handle(dev)), "eslam_planes_to_half")
        return self


def half_planes(all_planes):
    """float16, channels-last copies of the 12 planes (one texel = 64 contiguous bytes), as a HalfPlanes."""
    return HalfPlanes(all_planes)


def render_batch_ray_lowp(renderer, all_planes, planes_f16, decoders, rays_d, rays_o, truncation, gt_depth, _rand=None):
    """Renderer.render_batch_ray (reference src/utils/Renderer.py:63-147) on the mixed-precision kernels, no autograd:
    returns depth [R], rgb [R,3], sdf [R,S], z_vals [R,S]."""
    with torch.no_grad(), ops.mixed_precision(planes_f16):
        return renderer.render_batch_ray(all_planes, decoders, rays_d, rays_o, rays_o.device, truncation, gt_depth=gt_depth,
                                         _rand=_rand)
