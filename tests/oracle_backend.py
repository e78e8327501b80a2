"""The interface of myslam_amd.slam.HipBackend over the CPU oracle: the same tracking + mapping loop can then run on
the reference's arithmetic (PyTorch ops, float32, CPU) for quality comparisons at equal iterations.  Test-only."""
import torch
import torch.nn as nn

from oracle import eslam_oracle as orc
from myslam_amd.src import common


class OracleDecoders(nn.Module):
    """Parameter container with the reference's names (src/networks/decoders.py:47-62); evaluation is the oracle's."""

    def __init__(self, learnable_beta=True):
        super().__init__()
        self.linears = nn.ModuleList([nn.Linear(64, 16), nn.Linear(16, 16)])
        self.c_linears = nn.ModuleList([nn.Linear(64, 16), nn.Linear(16, 16)])
        self.output_linear = nn.Linear(16, 1)
        self.c_output_linear = nn.Linear(16, 3)
        self.beta = nn.Parameter(10 * torch.ones(1)) if learnable_beta else 10

    def param_dict(self):
        return {k: v for k, v in self.named_parameters() if k != "beta"}


class OracleBackend:
    def __init__(self, sc):
        self.sc = sc
        self.device = torch.device("cpu")
        self.Decoders = OracleDecoders
        self.Adam = lambda groups: torch.optim.Adam(groups, foreach=False)
        self.matrix_to_cam_pose = common.matrix_to_cam_pose          # plain torch ops, device-agnostic
        self.cam_pose_to_matrix = common.cam_pose_to_matrix
        wd = lambda w: dict(zip(("w_fs", "w_center", "w_tail", "w_depth", "w_color"), w))
        self.tracking_loss = lambda d, c, s, z, gd, gc, tr, w: orc.tracking_loss(d, c, s, z, gd, gc, tr, wd(w))
        self.mapping_loss = lambda d, c, s, z, gd, gc, tr, w: orc.mapping_loss(d, c, s, z, gd, gc, tr, wd(w))

    def get_samples(self, H0, H1, W0, W1, n, H, W, fx, fy, cx, cy, c2ws, depths, colors, device):
        idx = torch.randint((H1 - H0) * (W1 - W0), (n * c2ws.shape[0],))
        return orc.rays_from_pixels(idx, H0, H1, W0, W1, fx, fy, cx, cy, c2ws, depths, colors)

    def aabb_exit(self, ro, rd):
        return orc.aabb_exit(ro.detach(), rd.detach(), self.sc.bound)

    def render_batch_ray(self, all_planes, decoders, rays_d, rays_o, truncation, gt_depth):
        R, ns, ni = rays_o.shape[0], self.sc.n_stratified, self.sc.n_importance
        return orc.render_batch_ray(all_planes, decoders.param_dict(), decoders.beta, self.sc.bound, rays_d, rays_o,
                                    truncation, gt_depth, ns, ni, torch.rand(R, ns + ni), torch.rand(R, ns),
                                    torch.rand(R, ni))

    def render_img(self, all_planes, decoders, c2w, truncation, gt_depth):
        sc = self.sc
        ro, rd = orc.rays_full_image(sc.H, sc.W, sc.fx, sc.fy, sc.cx, sc.cy, c2w)
        ro, rd, gd = ro.reshape(-1, 3), rd.reshape(-1, 3), gt_depth.reshape(-1)
        ds, cs = [], []
        for lo in range(0, ro.shape[0], 20000):
            sl = slice(lo, lo + 20000)
            d, c, _, _ = self.render_batch_ray(all_planes, decoders, rd[sl], ro[sl], truncation, gd[sl])
            ds.append(d)
            cs.append(c)
        return torch.cat(ds).reshape(sc.H, sc.W).double(), torch.cat(cs).reshape(sc.H, sc.W, 3)

    def keyframe_selection_overlap(self, ns, gt_color, gt_depth, c2w, num, num_samples=8, num_rays=50):
        sc = self.sc
        ro, rd, d, _ = self.get_samples(0, sc.H, 0, sc.W, num_rays, sc.H, sc.W, sc.fx, sc.fy, sc.cx, sc.cy, c2w[None],
                                        gt_depth[None], gt_color[None], "cpu")
        kf = torch.stack([ns.estimate_c2w_list[i] for i in ns.keyframe_list], 0)[:-2]
        pct = orc.keyframe_overlap(ro, rd, d, kf, sc.H, sc.W, sc.fx, sc.fy, sc.cx, sc.cy, num_samples)
        n_sel = int((pct != 0).sum())
        return orc.select_overlapping(pct, num, torch.randperm(n_sel))
