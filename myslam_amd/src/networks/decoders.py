"""Decoders: same parameters, attribute names and call signatures as reference src/networks/decoders.py:28-146,
evaluated by the fused HIP kernels (tri-plane gather + both MLPs in one launch).

state_dict keys (the checkpoint compatibility surface, reference src/utils/Logger.py:41-47):
  beta, linears.{0,1}.{weight,bias}, c_linears.{0,1}.{weight,bias}, output_linear.*, c_output_linear.*
"""
import torch
import torch.nn as nn

from ... import ops

_UNIT_BOUND = (-1.0, 1.0, -1.0, 1.0, -1.0, 1.0)


class Decoders(nn.Module):
    def __init__(self, c_dim=32, hidden_size=16, truncation=0.08, n_blocks=2, learnable_beta=True):
        super().__init__()
        if c_dim != 32 or hidden_size != 16 or n_blocks != 2:
            raise NotImplementedError("the HIP kernels are specialised for c_dim=32, hidden_size=16, n_blocks=2 "
                                      "(the only configuration the reference ships, configs/ESLAM.yaml:76-78)")
        self.c_dim = c_dim
        self.truncation = truncation
        self.n_blocks = n_blocks
        # decoders.py:47-57
        self.linears = nn.ModuleList([nn.Linear(2 * c_dim, hidden_size)] +
                                     [nn.Linear(hidden_size, hidden_size) for _ in range(n_blocks - 1)])
        self.c_linears = nn.ModuleList([nn.Linear(2 * c_dim, hidden_size)] +
                                       [nn.Linear(hidden_size, hidden_size) for _ in range(n_blocks - 1)])
        self.output_linear = nn.Linear(hidden_size, 1)
        self.c_output_linear = nn.Linear(hidden_size, 3)
        # decoders.py:59-62
        if learnable_beta:
            self.beta = nn.Parameter(10 * torch.ones(1))
        else:
            self.beta = 10
        # self.bound is set from outside (reference ESLAM.py:173), a CPU [3,2] tensor

    def _decode(self, p_flat, bound6, all_planes):
        all_planes = ops.planes_for_kernels(all_planes, p_flat.shape[0])
        flat_planes = [p for grp in all_planes for p in grp]
        dummy_beta = ops.beta_tensor(10, p_flat.device)
        return ops.DecodeFn.apply(p_flat, bound6, dummy_beta, *flat_planes, *ops.decoder_params(self))

    def get_raw_sdf(self, p_nor, all_planes):
        """decoders.py:87-105.  p_nor are already-normalised coordinates, so the kernel's normalisation is made the
        identity by passing the unit cube as bound."""
        p = p_nor.reshape(-1, 3)
        # the no-autograd kernel only when NOTHING that feeds the sdf wants a gradient: points, decoders or the geometry planes
        # (frozen decoders with trainable planes still back-propagate through grid_sample in the reference)
        wants = p.requires_grad or any(t.requires_grad for t in self.parameters()) or \
            any(t.requires_grad for grp in all_planes[:3] for t in grp)
        if not (torch.is_grad_enabled() and wants):
            return ops.decode_sdf_only(p, _UNIT_BOUND, ops.planes_for_kernels(all_planes, p.shape[0]), self)
        return self._decode(p, _UNIT_BOUND, all_planes)[:, 3]

    def get_raw_rgb(self, p_nor, all_planes):
        """decoders.py:107-125."""
        return self._decode(p_nor.reshape(-1, 3), _UNIT_BOUND, all_planes)[:, :3]

    def forward(self, p, all_planes):
        """decoders.py:127-146: p [...,3] world coordinates -> raw [...,4] = (rgb, sdf)."""
        p_shape = p.shape
        raw = self._decode(p.reshape(-1, 3), ops.bound_to_host(self.bound), all_planes)
        return raw.reshape(*p_shape[:-1], -1)
