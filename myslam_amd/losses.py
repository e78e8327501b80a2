"""Caller-side losses of one optimisation iteration on the fused HIP loss kernels.

The reference computes them with boolean-mask indexing in its loops (src/Mapper.py:110-144,337-346 and
src/Tracker.py:114-148,192-204); these functions return the same scalar with the same gradients, in two kernel
launches and without host synchronisation.  Weights are the reference's config values (configs/ESLAM.yaml:29-33,53-57).
"""
import torch

from . import ops

MAPPING_W = (5.0, 200.0, 10.0, 0.1, 5.0)       # w_sdf_fs, w_sdf_center, w_sdf_tail, w_depth, w_color
TRACKING_W = (10.0, 200.0, 50.0, 1.0, 5.0)


def mapping_loss(depth, color, sdf, z_vals, gt_depth, gt_color, truncation, weights=MAPPING_W):
    """Mapper.py:337-346: SDF + depth terms over rays with gt_depth > 0, colour over all rays."""
    return ops.MappingLossFn.apply(depth, color, sdf, z_vals, gt_depth, gt_color, truncation, weights, None, None, None)


def tracking_loss(depth, color, sdf, z_vals, gt_depth, gt_color, truncation, weights=TRACKING_W):
    """Tracker.py:192-204: all terms over the rays whose depth error is below 10x the median error."""
    err = (gt_depth - depth.detach()).abs()
    mask = err < 10 * err.median()
    return ops.MappingLossFn.apply(depth, color, sdf, z_vals, gt_depth, gt_color, truncation, weights, mask, None, None)
