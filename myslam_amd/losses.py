"""Caller-side losses of one optimisation iteration on the fused HIP loss kernels.

The reference computes them with boolean-mask indexing in its loops (src/Mapper.py:110-144,337-346 and
src/Tracker.py:114-148,192-204); these functions return the same scalar with the same gradients, in two kernel
launches and without host synchronisation.  Weights are the reference's config values (configs/ESLAM.yaml:29-33,53-57).
"""
import torch

from . import ops

MAPPING_W = (5.0, 200.0, 10.0, 0.1, 5.0)       # w_sdf_fs, w_sdf_center, w_sdf_tail, w_depth, w_color
TRACKING_W = (10.0, 200.0, 50.0, 1.0, 5.0)


def mapping_loss(depth, color, sdf, z_vals, gt_depth, gt_color, truncation, weights=MAPPING_W, ray_mask=None,
                 precomputed=None):
    """Mapper.py:337-346: SDF + depth terms over rays with gt_depth > 0, colour over all rays.
    ray_mask (bool [R], optional): restrict every term to these rays - the AABB pre-filter of Mapper.py:322-332 kept
    as a mask, so the batch keeps a static shape and no boolean index forces a host sync.
    precomputed: the ops.fused_loss context the forward pass ran under (Renderer.render_batch_ray_with_loss): the value
    comes from the forward kernel and the gradients are formed inside the backward kernel - nothing is launched here."""
    if precomputed is not None:
        if precomputed.loss is None:
            raise RuntimeError("mapping_loss: the fused_loss context has not seen a forward pass")
        return precomputed.loss              # an output of the render's own autograd node (ops.RenderFn)
    return _loss(depth, color, sdf, z_vals, gt_depth, gt_color, truncation, weights, ray_mask)


def _loss(depth, color, sdf, z_vals, gt_depth, gt_color, truncation, weights, mask):
    """The loss node: compiled (eslam_torch_ext.mapping_loss) when the extension is there, else the Python Function - the same
    two launches either way."""
    ext = ops.torch_ext()
    if ext is not None and sdf.is_cuda and sdf.dim() == 2:
        if mask is not None:
            mask = mask.view(torch.uint8) if mask.dtype == torch.bool else mask.to(torch.uint8)
        return ext.mapping_loss(depth, color, sdf, z_vals, gt_depth, gt_color, float(truncation), [float(v) for v in weights], mask,
                                ops._loss_scratch(sdf.device, sdf.shape[0]))
    return ops.MappingLossFn.apply(depth, color, sdf, z_vals, gt_depth, gt_color, truncation, weights, mask, None, None)


def tracking_loss(depth, color, sdf, z_vals, gt_depth, gt_color, truncation, weights=TRACKING_W, ray_mask=None):
    """Tracker.py:192-204: all terms over the rays whose depth error is below 10x the median error.
    ray_mask (bool [R], optional): the pre-filter of Tracker.py:175-187 (inside the bound and depth > 0) as a mask;
    the median is then taken over the masked rays only, without a host sync (lower median, as torch.median)."""
    if depth.is_cuda and depth.shape[0] <= ops.TRACKING_MASK_MAX:
        mask = ops.tracking_mask(depth, gt_depth, ray_mask)                 # one launch, no host sync
    else:
        err = (gt_depth - depth.detach()).abs()
        if ray_mask is None:
            mask = err < 10 * err.median()
        else:
            srt = torch.where(ray_mask, err, torch.full_like(err, float("inf"))).sort().values
            k = ((ray_mask.sum() - 1).clamp(min=0) // 2).reshape(1)
            mask = ray_mask & (err < 10 * srt.gather(0, k))
    return _loss(depth, color, sdf, z_vals, gt_depth, gt_color, truncation, weights, mask)
