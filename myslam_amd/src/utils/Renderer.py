"""Renderer: same constructor, attributes and methods as reference src/utils/Renderer.py:26-204, running on the
fused HIP kernels.  Holds only Python scalars and the bound tensor, so it pickles into the reference's tracker
and mapper processes (src/ESLAM.py:246-260); the native library is loaded lazily in each process.
"""
import torch

from ... import ops
from ..common import get_rays


class Renderer(object):
    def __init__(self, cfg, eslam, ray_batch_size=10000):
        # Renderer.py:34-44
        self.ray_batch_size = ray_batch_size
        self.perturb = cfg['rendering']['perturb']
        self.n_stratified = cfg['rendering']['n_stratified']
        self.n_importance = cfg['rendering']['n_importance']
        self.scale = cfg['scale']
        self._bound6 = ops.bound_to_host(eslam.bound)        # host copy: no device sync per call
        self.bound = eslam.bound.to(eslam.device, non_blocking=True)
        self.H, self.W, self.fx, self.fy, self.cx, self.cy = eslam.H, eslam.W, eslam.fx, eslam.fy, eslam.cx, eslam.cy

    def render_batch_ray(self, all_planes, decoders, rays_d, rays_o, device, truncation, gt_depth=None, _rand=None):
        """Renderer.py:63-147.  NB argument order (rays_d, rays_o).  Returns depth [R], rgb [R,3], sdf [R,S],
        z_vals [R,S].  `_rand` (tests only) injects the three uniform tensors instead of drawing them."""
        n_rays = rays_o.shape[0]
        S = self.n_stratified + self.n_importance
        if gt_depth is None:
            # the reference dereferences gt_depth unconditionally (Renderer.py:91)
            raise AttributeError("'NoneType' object has no attribute 'reshape'")
        if n_rays == 0:
            e = rays_o.new_empty
            return e(0), e(0, 3), e(0, S), e(0, S)
        beta = ops.beta_tensor(decoders.beta, rays_o.device)
        if ops.ext_render_ok(rays_o, self.n_stratified, _rand):
            # the common case as ONE compiled call (eslam_torch_ext.cpp); everything below is the same sequence through ctypes
            return ops.ext_render(self._ext_cfg(truncation, decoders), rays_o, rays_d, gt_depth, beta,
                                  [p for grp in all_planes for p in grp], ops.decoder_params(decoders), self.n_stratified,
                                  self.n_importance, ops.current_fused_loss(), ops.wants_relayout(all_planes, n_rays * S))
        # planes in the reference's NCHW layout: per-call channels-last scratch copies (gradients flow back through them)
        all_planes = ops.planes_for_kernels(all_planes, n_rays * S)
        flat_planes = [p for grp in all_planes for p in grp]
        # training calls get a direction-sorted ray order (better L2 locality forward, bundling for the scatter); it only
        # depends on the rays, so it runs on a side stream next to the samplers
        grad_on = torch.is_grad_enabled()
        planes_grad = grad_on and any(p.requires_grad for p in flat_planes)
        wants_grad = planes_grad or (grad_on and (rays_o.requires_grad or rays_d.requires_grad or
                                                  any(p.requires_grad for p in ops.decoder_params(decoders)) or
                                                  (torch.is_tensor(decoders.beta) and decoders.beta.requires_grad)))
        # (the order bundles rays for the plane-gradient scatter; a call whose planes take no gradient - tracking - has no use for it)
        order = ops.ray_order_async(rays_o, rays_d, flat_planes) if planes_grad else None
        z_vals = ops.sample_z(rays_o, rays_d, gt_depth, all_planes, decoders, self._bound6, truncation,
                              self.n_stratified, self.n_importance, self.perturb, _rand)
        # pts are normalised with decoders.bound (decoders.py:138), the importance sampler uses renderer.bound
        bound6 = ops.bound_to_host(decoders.bound)
        fl = ops.current_fused_loss()          # set by render_batch_ray_with_loss
        outs = ops.RenderFn.apply(rays_o, rays_d, z_vals, bound6, beta, order, fl, *flat_planes,
                                  *ops.decoder_params(decoders))
        if fl is not None:
            fl.loss = outs[3]
        return outs[0], outs[1], outs[2], z_vals

    def _ext_cfg(self, truncation, decoders):
        """eslam_torch_ext.Config for this renderer and these decoders' bound, rebuilt when anything it holds changes.  Kept out
        of __getstate__: the Renderer pickles into the reference's tracker / mapper processes (ESLAM.py:246-260)."""
        b = decoders.bound
        key = (float(truncation), self.n_stratified, self.n_importance, bool(self.perturb), self._bound6, id(b),
               b._version if torch.is_tensor(b) else None)
        hit = self.__dict__.get("_ext_cfg_cache")
        if hit is None or hit[0] != key:
            cfg = ops.torch_ext().Config(self.n_stratified, self.n_importance, bool(self.perturb), float(truncation),
                                         list(self._bound6), list(ops.bound_to_host(b)))
            hit = self.__dict__["_ext_cfg_cache"] = (key, cfg)
        return hit[1]

    def __getstate__(self):
        d = dict(self.__dict__)
        d.pop("_ext_cfg_cache", None)
        return d

    def render_batch_ray_with_loss(self, all_planes, decoders, rays_d, rays_o, device, truncation, gt_depth, gt_color,
                                   weights, ray_mask=None, _rand=None, acc_out=None):
        """render_batch_ray + the mapping loss (src/Mapper.py:337-346): its sums are formed in the forward kernel's
        epilogue and its gradients inside the backward kernel.  Returns (depth, rgb, sdf, z_vals, pre); pre.loss is the
        loss (losses.mapping_loss(..., precomputed=pre) returns it): call .backward() on it."""
        with ops.fused_loss(gt_depth, gt_color, truncation, weights, ray_mask, acc_out=acc_out) as pre:
            depth, rgb, sdf, z_vals = self.render_batch_ray(all_planes, decoders, rays_d, rays_o, device, truncation,
                                                            gt_depth=gt_depth, _rand=_rand)
        return depth, rgb, sdf, z_vals, pre

    def sdf2alpha(self, sdf, beta=10):
        """Renderer.py:149-153 (kept for callers; the kernels fuse it)."""
        return 1. - torch.exp(-beta * torch.sigmoid(-sdf * beta))

    def render_img(self, all_planes, decoders, c2w, truncation, device, gt_depth=None, _rand_chunks=None):
        """Renderer.py:155-204: all H*W rays in chunks of ray_batch_size, no grad; depth returned as float64.
        `_rand_chunks` (tests only): per chunk the three uniform tensors to inject instead of drawing them."""
        with torch.no_grad():
            H, W = self.H, self.W
            rays_o, rays_d = get_rays(H, W, self.fx, self.fy, self.cx, self.cy, c2w, device)
            rays_o = rays_o.reshape(-1, 3)
            rays_d = rays_d.reshape(-1, 3)
            gt_depth = gt_depth.reshape(-1)
            depth_list, color_list = [], []
            for k, i in enumerate(range(0, rays_d.shape[0], self.ray_batch_size)):
                ret = self.render_batch_ray(all_planes, decoders, rays_d[i:i + self.ray_batch_size],
                                            rays_o[i:i + self.ray_batch_size], device, truncation,
                                            gt_depth=gt_depth[i:i + self.ray_batch_size],
                                            _rand=None if _rand_chunks is None else _rand_chunks[k])
                depth_list.append(ret[0].double())
                color_list.append(ret[1])
            return torch.cat(depth_list, 0).reshape(H, W), torch.cat(color_list, 0).reshape(H, W, 3)
