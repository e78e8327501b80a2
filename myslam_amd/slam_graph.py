"""The tracking + mapping loop of slam.py with every optimisation iteration captured into a hipGraph.

Issued eagerly from Python, one tracking iteration costs ~3 ms of host time (about 60 launches, three boolean-index
compactions that each synchronise, a .item()) for 0.26 ms of GPU work.  Here the iteration is made shape-static and
sync-free and is replayed as a graph:

  * the AABB / depth pre-filter (Tracker.py:175-187, Mapper.py:322-332) stays a MASK: filtered rays are still
    rendered (a few per cent of the batch) but excluded from every mean of the loss (eslam_loss_* `ray_mask`), instead
    of being compacted away with a boolean index;
  * the tracker's 10x-median outlier test takes the median over the masked rays on the device (losses.tracking_loss);
  * "keep the pose with the smallest loss" (Tracker.py:304-307) is a device-side select, not a .item() comparison;
  * the optimiser is the fused Adam with its step counter in device memory (optim.Adam(capturable=True)), and
    "a new optimiser for every frame" (Mapper.py:291, Tracker.py:279) becomes zeroing its state in place.

One graph per tracker and one per (window size, joint_opt, lr_factor) of the mapper, built on first use; images,
poses and camera parameters live in static buffers that are overwritten between replays.  Random pixels and jitter
still change every replay (torch's graph-safe Philox generator).  Same arithmetic per kept ray as slam.Slam; the
random streams differ (all rays draw jitter, not only the kept ones), so results agree statistically, not bitwise.
"""
import time
from types import SimpleNamespace

import torch

from . import losses, ops, optim
from .slam import HipBackend, Slam


class _EagerReplay:
    """replay() = call the iteration directly: the sync-free formulation without hipGraphs (use_graphs=False)."""

    def __init__(self, fn):
        self.replay = fn


class GraphedSlam(Slam):
    def __init__(self, sc, cfg=None, device="cuda:0", seed=0, warmup=2, use_graphs=True):
        """use_graphs=False keeps the sync-free iterations (masks, device-side median / best pose, in-place optimiser
        reset) but issues them eagerly - for hosts that cannot capture graphs; about 2x the plain eager loop."""
        super().__init__(sc, cfg, device, backend=HipBackend(sc, device), seed=seed)
        self.warmup = warmup
        self.use_graphs = bool(use_graphs)
        self._bound6 = ops.bound_to_host(sc.bound)
        # persistent Parameters: the graphs hold their addresses (the reference re-wraps the same storages per frame)
        for grp in self.all_planes:
            for i, p in enumerate(grp):
                grp[i] = torch.nn.Parameter(p.detach())
        self._plane_groups = ([p for grp in self.all_planes[:3] for p in grp], [p for grp in self.all_planes[3:] for p in grp])
        b_max = self.cfg.mapping_window_size + 2
        self._depths = torch.zeros(b_max, sc.H, sc.W, device=self.device)
        self._colors = torch.zeros(b_max, sc.H, sc.W, 3, device=self.device)
        self._trk = None
        self._map = {}

    # ------------------------------------------------------------------------------------------------------------
    def _capture(self, fn, reset):
        """Warm up `fn` on a side stream (creates optimiser state outside the graph), reset, capture."""
        if not self.use_graphs:
            return _EagerReplay(fn)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(self.warmup):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        # a private memory pool per graph: the graphs are replayed in data-dependent order (window sizes come and go)
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            fn()
        torch.cuda.synchronize()
        reset()
        self.stats["graphs"] = self.stats.get("graphs", 0) + 1
        self.stats["capture_seconds"] = self.stats.get("capture_seconds", 0.0) + time.perf_counter() - t0
        return g

    @staticmethod
    def _reset_adam(opt):
        for st in opt.state.values():
            st["exp_avg"].zero_()
            st["exp_avg_sq"].zero_()
        if opt._step_dev is not None:
            opt._step_dev.zero_()

    # ------------------------------------------------------------------------------------------------------------
    def _build_tracker(self, cam_pose, gt_color, gt_depth):
        """Static buffers + one captured tracking iteration (Tracker.py:150-210).  The buffers are filled with the first
        frame before the warm-up iterations run, so that they see real data."""
        be, cfg, sc, dev = self.be, self.cfg, self.sc, self.device
        st = SimpleNamespace()
        st.depth = gt_depth[None].clone()
        st.color = gt_color[None].clone()
        st.init = cam_pose.detach().clone()
        st.T = torch.nn.Parameter(st.init[:, 4:].clone())
        st.R = torch.nn.Parameter(st.init[:, :4].clone())
        st.best = torch.full((1,), float("inf"), device=dev)
        st.best_pose = st.init.clone()
        st.opt = optim.Adam([{"params": [st.T], "lr": cfg.lr_T, "betas": (0.5, 0.999)},
                             {"params": [st.R], "lr": cfg.lr_R, "betas": (0.5, 0.999)}], capturable=True)
        planes = tuple([p.detach() for p in grp] for grp in self.all_planes)          # Tracker.py:222-232

        def iteration():
            pose = torch.cat([st.R, st.T], -1)
            c2w = be.cam_pose_to_matrix(pose)
            ro, rd, gd, gc = be.get_samples(cfg.ignore_edge_H, sc.H - cfg.ignore_edge_H, cfg.ignore_edge_W,
                                            sc.W - cfg.ignore_edge_W, cfg.tracking_pixels, sc.H, sc.W, sc.fx, sc.fy,
                                            sc.cx, sc.cy, c2w, st.depth, st.color, dev)
            keep = ops.prefilter(ro, rd, gd, self._bound6, True)                      # Tracker.py:175-182, as a mask
            depth, color, sdf, z = be.render_batch_ray(planes, self.decoders, rd, ro, self.truncation, gd)
            loss = losses.tracking_loss(depth, color, sdf, z, gd, gc, self.truncation, cfg.tracking_w, ray_mask=keep)
            st.opt.zero_grad()
            loss.backward()
            st.opt.step()
            ops.keep_best(loss, pose, st.best, st.best_pose)                          # Tracker.py:304-307 on the device

        def reset():
            with torch.no_grad():
                st.R.copy_(st.init[:, :4])
                st.T.copy_(st.init[:, 4:])
                st.best.fill_(float("inf"))
            self._reset_adam(st.opt)

        st.reset = reset
        for p in self.decoders.parameters():
            p.requires_grad_(False)                                                   # Tracker.py:111-112
        try:
            st.graph = self._capture(iteration, reset)
        finally:
            for p in self.decoders.parameters():
                p.requires_grad_(True)
        return st

    def track(self, idx, gt_color, gt_depth):
        be, cfg = self.be, self.cfg
        pre = self.estimate_c2w_list[idx - 1][None]
        if cfg.const_speed_assumption and idx - 2 >= 0:                               # Tracker.py:270-274
            pp = be.matrix_to_cam_pose(torch.stack([self.estimate_c2w_list[idx - 2], pre[0]], 0))
            cam_pose = 2 * pp[1:] - pp[0:1]
        else:
            cam_pose = be.matrix_to_cam_pose(pre)
        st = self._trk
        if st is None:
            st = self._trk = self._build_tracker(cam_pose, gt_color, gt_depth)
        st.depth[0].copy_(gt_depth)
        st.color[0].copy_(gt_color)
        st.init.copy_(cam_pose)
        st.reset()
        for _ in range(cfg.tracking_iters):
            st.graph.replay()
        self.stats["tracking_iters"] += cfg.tracking_iters
        self.stats["tracking_rays"] += cfg.tracking_iters * cfg.tracking_pixels
        return be.cam_pose_to_matrix(st.best_pose)[0]

    # ------------------------------------------------------------------------------------------------------------
    def _build_mapper(self, b, joint, lr_factor, c2ws):
        """One captured mapping iteration (Mapper.py:308-350) for a window of b frames; self._depths[:b] / _colors[:b]
        and c2ws already hold the window that triggered the build."""
        be, cfg, sc, dev = self.be, self.cfg, self.sc, self.device
        st = SimpleNamespace()
        st.c2ws = c2ws.detach().clone()
        depths, colors = self._depths[:b], self._colors[:b]
        pixs = cfg.mapping_pixels // b
        st.rays = pixs * b
        params = list(self.decoders.parameters())
        groups = [{"params": params, "lr": cfg.decoders_lr * lr_factor},
                  {"params": self._plane_groups[0], "lr": cfg.planes_lr * lr_factor},
                  {"params": self._plane_groups[1], "lr": cfg.c_planes_lr * lr_factor}]
        st.cam_poses = st.cam_init = None
        if joint:                                                                     # Mapper.py:288-306
            st.cam_init = be.matrix_to_cam_pose(st.c2ws[1:]).detach().clone()
            st.cam_poses = torch.nn.Parameter(st.cam_init.clone())
            groups.append({"params": [st.cam_poses], "lr": cfg.joint_opt_cam_lr})
        st.opt = optim.Adam(groups, capturable=True)

        def iteration():
            c2ws_ = torch.cat([st.c2ws[0:1], be.cam_pose_to_matrix(st.cam_poses)], 0) if joint else st.c2ws
            ro, rd, gd, gc = be.get_samples(0, sc.H, 0, sc.W, pixs, sc.H, sc.W, sc.fx, sc.fy, sc.cx, sc.cy, c2ws_, depths,
                                            colors, dev)
            keep = ops.prefilter(ro, rd, gd, self._bound6, False)                     # Mapper.py:322-328, as a mask
            depth, color, sdf, z, pre = be.renderer.render_batch_ray_with_loss(
                self.all_planes, self.decoders, rd, ro, dev, self.truncation, gd, gc, cfg.mapping_w, ray_mask=keep)
            loss = losses.mapping_loss(depth, color, sdf, z, gd, gc, self.truncation, cfg.mapping_w, precomputed=pre)
            st.opt.zero_grad()
            loss.backward()
            st.opt.step()

        def reset():
            with torch.no_grad():
                if joint:
                    st.cam_poses.copy_(st.cam_init)
            self._reset_adam(st.opt)

        # the warm-up iterations really update planes and decoders: snapshot them and restore after the capture
        learned = params + self._plane_groups[0] + self._plane_groups[1]
        snap = [p.detach().clone() for p in learned]

        def restore_and_reset():
            with torch.no_grad():
                for p, s in zip(learned, snap):
                    p.copy_(s)
            reset()

        st.reset = reset
        st.graph = self._capture(iteration, restore_and_reset)
        return st

    def map(self, idx, gt_color, gt_depth, gt_c2w, cur_c2w, first):
        be, cfg, sc = self.be, self.cfg, self.sc
        iters = cfg.iters_first if first else cfg.iters
        lr_factor = cfg.lr_first_factor if first else cfg.lr_factor
        kd, kl = self.keyframe_dict, self.keyframe_list
        frames = []
        if len(kl) > 2:
            ns = SimpleNamespace(device=self.device, H=sc.H, W=sc.W, fx=sc.fx, fy=sc.fy, cx=sc.cx, cy=sc.cy,
                                 estimate_c2w_list=self.estimate_c2w_list, keyframe_list=kl)
            frames = [int(i) for i in be.keyframe_selection_overlap(ns, gt_color, gt_depth, cur_c2w,
                                                                    cfg.mapping_window_size - 1)]
        if len(kl) > 1:
            frames = sorted(frames + [len(kl) - 1, len(kl) - 2])
        frames += [-1]
        b = len(frames)
        joint = cfg.joint_opt and len(kl) > 4                                         # Mapper.py:416
        for k, f in enumerate(frames):
            self._depths[k].copy_(kd[f]["depth"] if f != -1 else gt_depth)
            self._colors[k].copy_(kd[f]["color"] if f != -1 else gt_color)
        c2ws = torch.stack([kd[f]["est_c2w"] if f != -1 else cur_c2w for f in frames], 0)
        key = (b, joint, float(lr_factor))
        st = self._map.get(key)
        if st is None:
            st = self._map[key] = self._build_mapper(b, joint, lr_factor, c2ws)
        st.c2ws.copy_(c2ws)
        if joint:
            st.cam_init.copy_(be.matrix_to_cam_pose(c2ws[1:]))
        st.reset()
        for _ in range(iters):
            st.graph.replay()
        self.stats["mapping_iters"] += iters
        self.stats["mapping_rays"] += iters * st.rays
        if joint:                                                                     # Mapper.py:352-363
            new = be.cam_pose_to_matrix(st.cam_poses.detach())
            k = 0
            for f in frames[1:]:
                if f != -1:
                    kd[f]["est_c2w"] = new[k].clone()
                    k += 1
                else:
                    cur_c2w = new[-1].clone()
        return cur_c2w
