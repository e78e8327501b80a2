"""Synthetic mapping workload shared by bench.py, __graft_entry__.smoke() and the tests.

Scene = bound / intrinsics / plane shapes of a reference scene; planes ~ N(0, 0.01^2) (reference
src/ESLAM.py:201-210), decoders with default nn.Linear init, beta = 10, camera at the AABB centre with identity
rotation, depth image ~ U(0.5, 2.5) m, colour ~ U(0,1).  Rays come from get_samples on the whole image followed
by the caller-side AABB pre-filter of reference src/Mapper.py:322-332, exactly as a mapping iteration does.
"""
import os
from types import SimpleNamespace

import torch

from . import losses, ops, scene as scn, synth
from .src.common import get_samples_at
from .src.networks.decoders import Decoders
from .src.utils.Renderer import Renderer


_SEPARATE_LOSS = os.environ.get("ESLAM_SEPARATE_LOSS", "0") == "1"


class Workload:
    def __init__(self, scene_name, R, n_strat, n_imp, device, zero_frac=0.0, seed=0, channels_last=True,
                 rays_grad=False, planes="normal", model_seed=0, shard=None, state="initial", cams=1, focal_scale=1.0):
        """seed: image / pixel choice of this rank's rays; model_seed: planes and decoders (same on every rank of a
        data-parallel job, whose replicas must be identical).  shard = (rank, world): build the WHOLE batch (give every
        rank the same seed) and keep this rank's contiguous slice of its rays (parallel.shard_slice) - the ray-sharded
        mapping iteration of SURVEY.md section 8(e); R_total is the batch's ray count.
        state: "initial" = the reference's initial state (planes ~ 0.01, sdf ~ 0: the FIRST sample of a ray takes 99.9 % of
        the compositing weight); "trained" = planes x 60 and the SDF head's bias + 0.55, which spreads the weights over ~20
        samples per ray - the state parity tests need to see the colour features of later samples at all.
        cams > 1: the batch of a keyframe WINDOW (src/Mapper.py:308-319): R // cams pixels from each of `cams` cameras standing
        on a ring round the AABB centre and looking outwards in different directions (the bench workload is ONE camera)."""
        dev = torch.device(device)
        self.device = dev
        sc = scn.make_scene(scene_name)
        self.scene = sc
        self.truncation = sc.truncation
        gen = torch.Generator(device=dev)
        gen.manual_seed(model_seed)
        if planes == "normal":
            pl = scn.random_planes(sc, dev, generator=gen, channels_last=channels_last)
        else:
            pl = scn.synth_planes(sc, device=dev, channels_last=channels_last)
        self.planes = tuple([torch.nn.Parameter(p) for p in grp] for grp in pl)      # as Mapper.py:254-266
        torch.manual_seed(model_seed)
        self.decoders = Decoders(learnable_beta=sc.learnable_beta).to(dev)
        self.decoders.bound = sc.bound
        if state == "trained":
            with torch.no_grad():
                for grp in self.planes:
                    for p in grp:
                        p.mul_(60.0)
                self.decoders.output_linear.bias += 0.55
        elif state != "initial":
            raise ValueError(state)
        cfg = sc.cfg(perturb=True)
        cfg["rendering"]["n_stratified"] = n_strat
        cfg["rendering"]["n_importance"] = n_imp
        self.n_strat, self.n_imp, self.S = n_strat, n_imp, n_strat + n_imp
        eslam = SimpleNamespace(bound=sc.bound, device=dev, H=sc.H, W=sc.W, fx=sc.fx, fy=sc.fy, cx=sc.cx, cy=sc.cy)
        self.renderer = Renderer(cfg, eslam)
        self.c2w = scn.center_pose(sc)
        cams = int(cams)
        depth_img = torch.from_numpy(synth.depth_image(sc.H, sc.W, 10 + seed, zero_frac)).to(dev)[None].repeat(cams, 1, 1)
        color_img = torch.from_numpy(synth.color_image(sc.H, sc.W, 12 + seed)).to(dev)[None].repeat(cams, 1, 1, 1)
        R = (R // cams) * cams
        idx = torch.from_numpy(synth.hash_randint(sc.H * sc.W, (R,), 50_000 + seed)).to(dev)
        c2ws = self.c2w[None].repeat(cams, 1, 1)
        if cams > 1:
            import math
            half = 0.15 * (sc.bound[:, 1] - sc.bound[:, 0])
            for k in range(cams):
                a = 2.0 * math.pi * k / cams
                c2ws[k, :3, :3] = torch.tensor([[math.cos(a), 0.0, math.sin(a)], [0.0, 1.0, 0.0], [-math.sin(a), 0.0, math.cos(a)]])
                c2ws[k, :3, 3] += torch.tensor([math.sin(a), 0.0, math.cos(a)]) * half
        c2ws = c2ws.to(dev)
        with torch.no_grad():
            # (focal_scale: profiling knob - the same pixels through a longer lens, i.e. a narrower fan of rays)
            ro, rd, gd, gc = get_samples_at(idx, 0, sc.H, 0, sc.W, R // cams, sc.H, sc.W, sc.fx * focal_scale, sc.fy * focal_scale, sc.cx, sc.cy, c2ws,
                                            depth_img, color_img)
            inside = ops.aabb_exit(ro, rd, ops.bound_to_host(sc.bound)) >= gd           # Mapper.py:322-332
        ro, rd, gd, gc = ro[inside], rd[inside], gd[inside], gc[inside]
        self.R_total = int(ro.shape[0])
        # the WHOLE batch stays on every rank of a ray-sharded job (parallel.ShardedMapper forms the loss's global set sizes
        # and the union of texels the batch can touch from it, instead of exchanging them)
        self.all_rays_o, self.all_rays_d, self.all_gt_depth = ro.contiguous(), rd.contiguous(), gd.contiguous()
        self.ray_lo = 0
        if shard is not None:
            from .parallel import shard_slice
            lo, hi = shard_slice(self.R_total, shard[0], shard[1])
            ro, rd, gd, gc = ro[lo:hi], rd[lo:hi], gd[lo:hi], gc[lo:hi]
            self.ray_lo = lo
        self.rays_o = ro.contiguous().requires_grad_(rays_grad)
        self.rays_d = rd.contiguous().requires_grad_(rays_grad)
        self.gt_depth = gd.contiguous()
        self.gt_color = gc.contiguous()
        self.R = int(self.rays_o.shape[0])
        self._one = torch.ones((), device=dev)
        # fixed random numbers / cotangents for the reproducible (test) paths
        self._rand = (torch.from_numpy(synth.hash_uniform((self.R, self.S), 90_000)).to(dev),
                      torch.from_numpy(synth.hash_uniform((self.R, n_strat), 90_001)).to(dev),
                      torch.from_numpy(synth.hash_uniform((self.R, n_imp), 90_002)).to(dev))
        self._cot = (torch.from_numpy(synth.hash_uniform((self.R,), 91_000)).to(dev) - 0.5,
                     torch.from_numpy(synth.hash_uniform((self.R, 3), 91_001)).to(dev) - 0.5,
                     (torch.from_numpy(synth.hash_uniform((self.R, self.S), 91_002)).to(dev) - 0.5) * 0.1)

    @property
    def plane_list(self):
        pl = getattr(self, "_plane_list", None)
        if pl is None or pl[0] is not self.planes[0][0]:
            pl = self._plane_list = [p for grp in self.planes for p in grp]
        return pl

    def params(self):
        ps = getattr(self, "_params", None)
        if ps is None:          # (walking the module tree costs ~12 us a call, and a step asks twice)
            ps = self._params = self.plane_list + list(self.decoders.parameters())
        return ps

    def _slice(self, shard):
        if shard is None:
            return slice(0, self.R)
        k, n = shard
        per = (self.R + n - 1) // n
        return slice(k * per, min(self.R, (k + 1) * per))

    def forward(self, shard=None, fixed_rand=True):
        sl = self._slice(shard)
        rand = tuple(t[sl] for t in self._rand) if fixed_rand else None
        return self.renderer.render_batch_ray(self.planes, self.decoders, self.rays_d[sl], self.rays_o[sl],
                                              self.device, self.truncation, gt_depth=self.gt_depth[sl], _rand=rand)

    def backward_with(self, out, scale=1.0, shard=None):
        """Back-propagate fixed pseudo-random cotangents (a loss that is linear in the outputs); returns every
        parameter gradient as a list of tensors."""
        sl = self._slice(shard)
        depth, color, sdf, _ = out
        a, b, c = (t[sl] for t in self._cot)
        for p in self.params():
            p.grad = None
        (scale * ((depth * a).sum() + (color * b).sum() + (sdf * c).sum())).backward()
        return [p.grad.detach().clone() for p in self.params()]

    def step(self):
        """One mapping iteration minus the optimiser step: render, loss, backward."""
        for p in self.params():
            p.grad = None
        # the loss sums ride in the forward kernel's epilogue (eslam_render_fwd_loss): one launch less than a separate
        # eslam_loss_value; ESLAM_SEPARATE_LOSS=1 restores the two-call form the reference's loop has
        if _SEPARATE_LOSS:
            depth, color, sdf, z = self.renderer.render_batch_ray(self.planes, self.decoders, self.rays_d, self.rays_o,
                                                                  self.device, self.truncation, gt_depth=self.gt_depth)
            loss = losses.mapping_loss(depth, color, sdf, z, self.gt_depth, self.gt_color, self.truncation)
        else:
            depth, color, sdf, z, pre = self.renderer.render_batch_ray_with_loss(
                self.planes, self.decoders, self.rays_d, self.rays_o, self.device, self.truncation, self.gt_depth,
                self.gt_color, losses.MAPPING_W)
            loss = losses.mapping_loss(depth, color, sdf, z, self.gt_depth, self.gt_color, self.truncation,
                                       precomputed=pre)
        loss.backward(gradient=self._one)          # a cached 1.0 instead of autograd's ones_like(loss) fill per step
        return loss


class GraphedStep:
    """One mapping iteration captured into a hipGraph (torch.cuda.CUDAGraph) and replayed.

    A step is ~14 kernel launches of 5-250 us each; issued eagerly from Python they cost more host time than the GPU
    needs to run them, so the iteration is captured once (our C-ABI calls only enqueue on the current stream: no
    allocation, no sync) and replayed.  Every replay does the full work - new random jitter (the Philox offset of
    torch's generator advances per replay), forward, loss, backward - into static gradient buffers (p.grad)."""

    def __init__(self, step_fn, params, warmup=3):
        self.params = list(params)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                step_fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        for p in self.params:
            p.grad = None
        self.graph = torch.cuda.CUDAGraph()
        # thread_local: other threads (e.g. the RCCL watchdog of a process group) may touch the runtime while we capture
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.loss = step_fn()
        torch.cuda.synchronize()

    def __call__(self):
        self.graph.replay()
        return self.loss


def make_workload(scene_name, R, n_strat, n_imp, device, **kw):
    return Workload(scene_name, R, n_strat, n_imp, device, **kw)
