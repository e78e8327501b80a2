"""The field query of the reference's Mesher on the HIP path.

Only `eval_points` (reference src/utils/Mesher.py:130-157) lives here - the caller that pushes the 500k-point batches
of the marching-cubes grid through the decoders; grid construction, marching cubes and mesh clean-up stay with the
reference (CPU, skimage / open3d / trimesh).  Bind it with

    from myslam_amd.src.utils.Mesher import eval_points
    Mesher.eval_points = eval_points

Same arguments, same [N,4] result (rgb, sdf with -1 outside the bound).  The bound test is folded into the decode
kernel (ESLAM_DECODE_MASK_OUTSIDE), so a batch is one launch instead of a decode plus 8 mask / index ops, and all
batches write into one output tensor (no torch.cat).
"""
import ctypes

import torch

from ... import _hip, ops


def eval_points(self, p, all_planes, decoders):
    _hip.require_gpu_f32("p", p)
    p = ops._c(p.detach().reshape(-1, 3))
    N = p.shape[0]
    dev = p.device
    bound6 = ops.bound_to_host(decoders.bound)
    same_bound = bound6 == ops.bound_to_host(self.bound)
    arr, _ = _hip.make_planes(tuple([t.detach() for t in grp] for grp in all_planes))
    dec, keep = _hip.make_decoders([t.detach() for t in ops.decoder_params(decoders)], ops.beta_tensor(10, dev))
    out = torch.empty(N, 4, device=dev)
    lib = _hip.lib()
    step = int(self.points_batch_size)
    flags = 2 if same_bound else 0                       # ESLAM_DECODE_MASK_OUTSIDE
    with _hip.on_device(dev):
        for lo in range(0, N, step):                     # Mesher.py:141: torch.split(p, points_batch_size)
            n = min(step, N - lo)
            _hip.check(lib.eslam_decode_fwd(arr, ctypes.byref(dec), _hip.make_bound(bound6),
                                            ctypes.c_void_p(p.data_ptr() + lo * 12), n, flags,
                                            ctypes.c_void_p(out.data_ptr() + lo * 16), None, _hip.stream_handle(dev)),
                       "eslam_decode_fwd")
    if not same_bound:      # a mesher bound that differs from the decoders' normalisation bound: mask with its own
        b = self.bound.to(dev)
        inside = ((p < b[:, 1]) & (p > b[:, 0])).all(dim=1)
        out[~inside, -1] = -1
    return out
