"""A single-process tracking + mapping loop over the HIP path - the callers either side of the hot path, restated
compactly so that BASELINE.json configs[2] ("full tracking+mapping loop, ATE") can be run and measured end to end.

This is NOT the reference's orchestration (two spawned processes, shared-memory planes, DataLoader, Logger, Mesher,
visualiser - all out of scope).  It is the arithmetic of its two per-frame routines in the
lock-step order the reference's processes synchronise to (Tracker.py:253-256 waits for the mapping of frame idx-1,
Mapper.py:388-396 waits for the tracking of frame idx):

  frame 0          map with the ground-truth pose, `iters_first` iterations, lr x `lr_first_factor`  (Mapper.py:409-414)
  frame idx > 0    track: constant-speed initial pose (Tracker.py:270-274), `tracking_iters` Adam steps on (T, R) with
                   the tracking loss, keep the pose of the smallest loss                        (Tracker.py:279-307)
  idx % every == 0 map: window = overlap-selected keyframes + the last two + the current frame (Mapper.py:236-247),
                   `iters` Adam steps on decoders + planes (+ the window's poses when joint_opt, Mapper.py:288-306,
                   first pose fixed), then the frame becomes a keyframe                          (Mapper.py:419-424)

Every heavy step goes through a `backend` whose default is the product path: get_samples, the AABB pre-filter,
Renderer.render_batch_ray, the fused losses, the fused Adam, keyframe overlap selection - all HIP kernels behind the
C ABI.  The tests plug in a CPU backend built from the oracle to run the SAME loop and compare trajectory error and
render quality at equal iterations.
"""
from dataclasses import dataclass
from types import SimpleNamespace

import torch


@dataclass
class SlamConfig:
    """configs/ESLAM.yaml:17-61 with configs/Replica/replica.yaml overrides."""
    tracking_pixels: int = 2000
    tracking_iters: int = 8
    ignore_edge_H: int = 75
    ignore_edge_W: int = 75
    lr_T: float = 0.002
    lr_R: float = 0.001
    const_speed_assumption: bool = True
    tracking_w: tuple = (10.0, 200.0, 50.0, 1.0, 5.0)
    mapping_pixels: int = 4000
    iters_first: int = 1000
    iters: int = 15
    every_frame: int = 4
    keyframe_every: int = 4
    mapping_window_size: int = 20
    joint_opt: bool = True
    joint_opt_cam_lr: float = 0.001
    lr_first_factor: float = 5.0
    lr_factor: float = 1.0
    decoders_lr: float = 0.001
    planes_lr: float = 0.005
    c_planes_lr: float = 0.005
    mapping_w: tuple = (5.0, 200.0, 10.0, 0.1, 5.0)


class HipBackend:
    """The product path.  (tests/ provides the same interface over the CPU oracle.)"""

    def __init__(self, sc, device):
        from . import keyframes, losses, ops, optim
        from .src import common
        from .src.networks.decoders import Decoders
        from .src.utils.Renderer import Renderer
        self.device = torch.device(device)
        self.sc = sc
        eslam = SimpleNamespace(bound=sc.bound, device=self.device, H=sc.H, W=sc.W, fx=sc.fx, fy=sc.fy, cx=sc.cx, cy=sc.cy)
        self.renderer = Renderer(sc.cfg(perturb=True), eslam)
        self.Decoders = Decoders
        self.Adam = optim.Adam
        self.get_samples = common.get_samples
        self.matrix_to_cam_pose = common.matrix_to_cam_pose
        self.cam_pose_to_matrix = common.cam_pose_to_matrix
        self.mapping_loss = losses.mapping_loss
        self.tracking_loss = losses.tracking_loss
        self._bound6 = ops.bound_to_host(sc.bound)
        self._aabb_exit = ops.aabb_exit
        self._kf_select = keyframes.keyframe_selection_overlap

    def aabb_exit(self, rays_o, rays_d):
        return self._aabb_exit(rays_o.detach(), rays_d.detach(), self._bound6)

    def render_batch_ray(self, all_planes, decoders, rays_d, rays_o, truncation, gt_depth):
        return self.renderer.render_batch_ray(all_planes, decoders, rays_d, rays_o, self.device, truncation,
                                              gt_depth=gt_depth)

    def render_img(self, all_planes, decoders, c2w, truncation, gt_depth):
        return self.renderer.render_img(all_planes, decoders, c2w, truncation, self.device, gt_depth=gt_depth)

    def keyframe_selection_overlap(self, ns, gt_color, gt_depth, c2w, num):
        return self._kf_select(ns, gt_color, gt_depth, c2w, num)


class Slam:
    def __init__(self, sc, cfg=None, device="cuda:0", backend=None, seed=0):
        self.sc = sc
        self.cfg = cfg or SlamConfig()
        self.device = torch.device(device)
        self.be = backend or HipBackend(sc, device)
        self.truncation = sc.truncation
        gen = torch.Generator().manual_seed(seed)
        # ESLAM.py:201-210: planes ~ N(0, 0.01^2); same values for every backend (drawn on the CPU)
        planes = []
        for grp in sc.plane_shapes:
            lvl = []
            for shp in grp:
                t = torch.empty(shp).normal_(mean=0, std=0.01, generator=gen)
                lvl.append(t.to(self.device).contiguous(memory_format=torch.channels_last))
            planes.append(lvl)
        self.all_planes = tuple(planes)
        torch.manual_seed(seed)
        self.decoders = self.be.Decoders(learnable_beta=sc.learnable_beta).to(self.device)
        self.decoders.bound = sc.bound
        self.estimate_c2w_list = []
        self.gt_c2w_list = []
        self.keyframe_list = []
        self.keyframe_dict = []
        self.stats = dict(tracking_iters=0, mapping_iters=0, tracking_rays=0, mapping_rays=0)

    # ------------------------------------------------------------------------------------------------------------
    def _prefilter(self, ro, rd, gd, gc, need_depth):
        """Mapper.py:322-332 / Tracker.py:175-187: drop rays whose depth lies beyond the scene bound."""
        with torch.no_grad():
            inside = self.be.aabb_exit(ro, rd) >= gd
            if need_depth:
                inside = inside & (gd > 0)
        return ro[inside], rd[inside], gd[inside], gc[inside]

    def track(self, idx, gt_color, gt_depth):
        """Tracker.py:262-309 for one frame; returns the estimated c2w [4,4]."""
        be, cfg, sc = self.be, self.cfg, self.sc
        pre = self.estimate_c2w_list[idx - 1][None]
        if cfg.const_speed_assumption and idx - 2 >= 0:
            pp = be.matrix_to_cam_pose(torch.stack([self.estimate_c2w_list[idx - 2], pre[0]], 0))
            cam_pose = 2 * pp[1:] - pp[0:1]
        else:
            cam_pose = be.matrix_to_cam_pose(pre)
        T = torch.nn.Parameter(cam_pose[:, -3:].clone())
        R = torch.nn.Parameter(cam_pose[:, :4].clone())
        opt = be.Adam([{"params": [T], "lr": cfg.lr_T, "betas": (0.5, 0.999)},
                       {"params": [R], "lr": cfg.lr_R, "betas": (0.5, 0.999)}])
        planes = tuple([p.detach() for p in grp] for grp in self.all_planes)          # Tracker.py:222-232
        for p in self.decoders.parameters():
            p.requires_grad_(False)                                                   # Tracker.py:111-112
        best, best_pose = float("inf"), None
        for _ in range(cfg.tracking_iters):
            pose = torch.cat([R, T], -1)
            c2w = be.cam_pose_to_matrix(pose)
            ro, rd, gd, gc = be.get_samples(cfg.ignore_edge_H, sc.H - cfg.ignore_edge_H, cfg.ignore_edge_W,
                                            sc.W - cfg.ignore_edge_W, cfg.tracking_pixels, sc.H, sc.W, sc.fx, sc.fy,
                                            sc.cx, sc.cy, c2w, gt_depth[None], gt_color[None], self.device)
            ro, rd, gd, gc = self._prefilter(ro, rd, gd, gc, need_depth=True)
            depth, color, sdf, z = be.render_batch_ray(planes, self.decoders, rd, ro, self.truncation, gd)
            loss = be.tracking_loss(depth, color, sdf, z, gd, gc, self.truncation, cfg.tracking_w)
            opt.zero_grad()
            loss.backward()
            opt.step()
            lv = float(loss.detach())
            self.stats["tracking_iters"] += 1
            self.stats["tracking_rays"] += int(gd.shape[0])
            if lv < best:
                best, best_pose = lv, pose.clone().detach()
        for p in self.decoders.parameters():
            p.requires_grad_(True)
        return be.cam_pose_to_matrix(best_pose)[0]

    def map(self, idx, gt_color, gt_depth, gt_c2w, cur_c2w, first):
        """Mapper.py:211-365 for one frame; returns cur_c2w (updated when the window's poses are optimised)."""
        be, cfg, sc = self.be, self.cfg, self.sc
        iters = cfg.iters_first if first else cfg.iters
        lr_factor = cfg.lr_first_factor if first else cfg.lr_factor
        kd, kl = self.keyframe_dict, self.keyframe_list
        if len(kd) == 0:
            frames = []
        else:
            ns = SimpleNamespace(device=self.device, H=sc.H, W=sc.W, fx=sc.fx, fy=sc.fy, cx=sc.cx, cy=sc.cy,
                                 estimate_c2w_list=self.estimate_c2w_list, keyframe_list=kl)
            frames = [int(i) for i in be.keyframe_selection_overlap(ns, gt_color, gt_depth, cur_c2w,
                                                                    cfg.mapping_window_size - 1)] if len(kl) > 2 else []
        if len(kl) > 1:
            frames = sorted(frames + [len(kl) - 1, len(kl) - 2])
        frames += [-1]
        pixs = cfg.mapping_pixels // len(frames)
        dec_params = list(self.decoders.parameters())
        planes_para, c_planes_para = [], []
        for grp in self.all_planes[:3]:                                                # Mapper.py:254-266
            for i, p in enumerate(grp):
                grp[i] = torch.nn.Parameter(p.detach())
                planes_para.append(grp[i])
        for grp in self.all_planes[3:]:
            for i, p in enumerate(grp):
                grp[i] = torch.nn.Parameter(p.detach())
                c_planes_para.append(grp[i])
        gds = torch.stack([kd[f]["depth"] if f != -1 else gt_depth for f in frames], 0)
        gcs = torch.stack([kd[f]["color"] if f != -1 else gt_color for f in frames], 0)
        c2ws = torch.stack([kd[f]["est_c2w"] if f != -1 else cur_c2w for f in frames], 0)
        joint = cfg.joint_opt and len(kl) > 4                                          # Mapper.py:416
        groups = [{"params": dec_params, "lr": cfg.decoders_lr * lr_factor},
                  {"params": planes_para, "lr": cfg.planes_lr * lr_factor},
                  {"params": c_planes_para, "lr": cfg.c_planes_lr * lr_factor}]
        if joint:
            cam_poses = torch.nn.Parameter(be.matrix_to_cam_pose(c2ws[1:]))
            groups.append({"params": [cam_poses], "lr": cfg.joint_opt_cam_lr})
        opt = be.Adam(groups)
        for _ in range(iters):
            c2ws_ = torch.cat([c2ws[0:1], be.cam_pose_to_matrix(cam_poses)], 0) if joint else c2ws
            ro, rd, gd, gc = be.get_samples(0, sc.H, 0, sc.W, pixs, sc.H, sc.W, sc.fx, sc.fy, sc.cx, sc.cy, c2ws_, gds,
                                            gcs, self.device)
            ro, rd, gd, gc = self._prefilter(ro, rd, gd, gc, need_depth=False)
            depth, color, sdf, z = be.render_batch_ray(self.all_planes, self.decoders, rd, ro, self.truncation, gd)
            loss = be.mapping_loss(depth, color, sdf, z, gd, gc, self.truncation, cfg.mapping_w)
            opt.zero_grad()
            loss.backward()
            opt.step()
            self.stats["mapping_iters"] += 1
            self.stats["mapping_rays"] += int(gd.shape[0])
        if joint:                                                                      # Mapper.py:352-363
            new = be.cam_pose_to_matrix(cam_poses.detach())
            k = 0
            for f in frames[1:]:
                if f != -1:
                    kd[f]["est_c2w"] = new[k]
                    k += 1
                else:
                    cur_c2w = new[-1]
        return cur_c2w

    def run(self, frames, on_frame=None):
        """frames: iterable of (idx, colour [H,W,3], depth [H,W], gt_c2w [4,4]) on the device, idx = 0, 1, 2, ..."""
        cfg = self.cfg
        for idx, gt_color, gt_depth, gt_c2w in frames:
            if idx == 0:
                c2w = gt_c2w.clone()                                                   # Tracker.py:263-264
            else:
                c2w = self.track(idx, gt_color, gt_depth)
            self.estimate_c2w_list.append(c2w.detach().clone())
            self.gt_c2w_list.append(gt_c2w.clone())
            if idx % cfg.every_frame == 0:
                cur = self.map(idx, gt_color, gt_depth, gt_c2w, self.estimate_c2w_list[idx], first=(idx == 0))
                if cfg.joint_opt and len(self.keyframe_list) > 4:
                    self.estimate_c2w_list[idx] = cur.detach().clone()                 # Mapper.py:421-422
                if idx % cfg.keyframe_every == 0:
                    self.keyframe_list.append(idx)
                    self.keyframe_dict.append({"gt_c2w": gt_c2w, "idx": idx, "color": gt_color, "depth": gt_depth,
                                               "est_c2w": cur.detach().clone()})
            if on_frame is not None:
                on_frame(self, idx)
        return self.estimate_c2w_list

    # ------------------------------------------------------------------------------------------------------------
    def render_quality(self, gt_color, gt_depth, c2w):
        """PSNR (dB) of the rendered colour and L1 (length units) of the rendered depth against a frame."""
        with torch.no_grad():
            depth, color = self.be.render_img(self.all_planes, self.decoders, c2w, self.truncation, gt_depth)
        valid = gt_depth > 0
        l1 = float((depth.float() - gt_depth)[valid].abs().mean())
        mse = float(((color - gt_color) ** 2).mean())
        return dict(psnr=float(-10.0 * torch.log10(torch.tensor(mse))), depth_l1=l1)
