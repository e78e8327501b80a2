"""Keyframe selection by view overlap on the HIP path.

`keyframe_selection_overlap` has the signature of the reference's `Mapper.keyframe_selection_overlap`
(src/Mapper.py:146-209) with the mapper itself as first argument, so a maintainer binds it with

    from myslam_amd.keyframes import keyframe_selection_overlap
    Mapper.keyframe_selection_overlap = keyframe_selection_overlap

It reads the same attributes (`device, H, W, fx, fy, cx, cy, estimate_c2w_list, keyframe_list`), draws the same random
numbers in the same order (one `randint` inside get_samples, one `randperm`), and returns the same list.  The
projection test for all keyframes is one `eslam_keyframe_overlap` launch instead of a batched inverse and ~30 small
ops; the only host synchronisation is the copy of K+1 counters, which the reference needs as well (`nonzero`).
"""
import torch

from . import _hip
from .src.common import get_samples


def overlap_counts(rays_o, rays_d, gt_depth, keyframes_c2ws, H, W, fx, fy, cx, cy, num_samples=8, edge=20):
    """int32 [K+1] on the device: points inside each keyframe's image, then the number of rays with depth > 0."""
    for name, t in (("rays_o", rays_o), ("rays_d", rays_d), ("gt_depth", gt_depth), ("keyframes_c2ws", keyframes_c2ws)):
        _hip.require_gpu_f32(name, t)
    n = int(gt_depth.numel())
    K = int(keyframes_c2ws.shape[0])
    if keyframes_c2ws.shape[1:] != (4, 4) or rays_o.shape != (n, 3) or rays_d.shape != (n, 3):
        raise RuntimeError("overlap_counts: expected rays [n,3], gt_depth [n], keyframes_c2ws [K,4,4]")
    dev = rays_o.device
    counts = torch.empty(K + 1, dtype=torch.int32, device=dev)
    with _hip.on_device(dev):
        _hip.check(_hip.lib().eslam_keyframe_overlap(
            _hip.ptr(rays_o.detach().contiguous()), _hip.ptr(rays_d.detach().contiguous()),
            _hip.ptr(gt_depth.contiguous()), n, int(num_samples), _hip.ptr(keyframes_c2ws.detach().contiguous()), K,
            int(H), int(W), float(fx), float(fy), float(cx), float(cy), int(edge), _hip.ptr(counts),
            _hip.stream_handle(dev)), "eslam_keyframe_overlap")
    return counts


def percent_inside(rays_o, rays_d, gt_depth, keyframes_c2ws, H, W, fx, fy, cx, cy, num_samples=8, edge=20):
    """Mapper.py:201: fraction of the sample points that each keyframe sees (float32 [K], on the device)."""
    c = overlap_counts(rays_o, rays_d, gt_depth, keyframes_c2ws, H, W, fx, fy, cx, cy, num_samples, edge)
    return c[:-1] / (c[-1] * num_samples)


def keyframe_selection_overlap(self, gt_color, gt_depth, c2w, num_keyframes, num_samples=8, num_rays=50):
    device = self.device
    H, W, fx, fy, cx, cy = self.H, self.W, self.fx, self.fy, self.cx, self.cy
    rays_o, rays_d, depth, _ = get_samples(0, H, 0, W, num_rays, H, W, fx, fy, cx, cy, c2w.unsqueeze(0),
                                           gt_depth.unsqueeze(0), gt_color.unsqueeze(0), device)
    keyframes_c2ws = torch.stack([self.estimate_c2w_list[idx] for idx in self.keyframe_list], dim=0)
    pct = percent_inside(rays_o, rays_d, depth, keyframes_c2ws[:-2].to(rays_o.device, torch.float32), H, W, fx, fy, cx,
                         cy, num_samples)
    selected = torch.nonzero(pct).squeeze(-1)                       # Mapper.py:203-207
    rnd_inds = torch.randperm(selected.shape[0])
    selected = selected[rnd_inds[:num_keyframes].to(selected.device)]
    return list(selected.cpu().numpy())
