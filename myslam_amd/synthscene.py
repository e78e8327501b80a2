"""An analytic RGB-D sequence for end-to-end runs of the hot path's callers (no dataset ships with the reference and
none can be fetched): a box room with three spheres, a smooth procedural albedo, and a camera that circles and pans
inside it.  Depth is exact ray casting in the reference's camera convention (src/common.py:183-201: pixel (i, j) ->
dir = [(i-cx)/fx, -(j-cy)/fy, -1], rays_d = R dir, so the ray parameter t IS the z-depth the dataset readers deliver).
Pure torch, device-agnostic: the GPU loop and the CPU oracle loop of the tests see the same frames.
"""
import math

import torch


class AnalyticRoom:
    def __init__(self, bound, margin=0.3, variant="r01"):
        b = bound.double()
        self.lo = b[:, 0] + margin
        self.hi = b[:, 1] - margin
        c = (self.lo + self.hi) / 2
        ext = (self.hi - self.lo)
        # centres relative to the room, radii relative to its height
        self.spheres = [(c + ext * torch.tensor(o, dtype=torch.float64), float(ext[2]) * r)
                        for o, r in (((0.28, 0.18, -0.25), 0.22), ((-0.30, -0.22, -0.30), 0.18), ((0.05, -0.30, 0.10), 0.15))]
        if variant == "rich":
            # five more spheres against the walls: with the three above, some views of the r01 room (a 90-degree field of
            # view 2.8 m from a wall) showed nothing but one plane, which leaves the translation along it to the smooth albedo
            self.spheres += [(c + ext * torch.tensor(o, dtype=torch.float64), float(ext[2]) * r)
                             for o, r in (((0.05, 0.42, 0.05), 0.16), ((-0.20, 0.40, -0.28), 0.14), ((0.33, 0.38, 0.22), 0.12),
                                          ((0.44, -0.05, 0.00), 0.17), ((-0.44, 0.10, 0.05), 0.17))]

    def cast(self, o, d):
        """o [3], d [...,3] (float32/64) -> depth t [...] and colour [...,3] of the first hit."""
        dt = d.dtype
        lo, hi = self.lo.to(d), self.hi.to(d)
        o = o.to(d)
        safe = torch.where(d.abs() < 1e-9, torch.full_like(d, 1e-9), d)
        t_wall = torch.maximum((lo - o) / safe, (hi - o) / safe).min(dim=-1).values         # exit of the box
        t = t_wall
        obj = torch.zeros(d.shape[:-1], dtype=torch.long, device=d.device)
        dd = (d * d).sum(-1)
        for k, (c, r) in enumerate(self.spheres):
            oc = o - c.to(d)
            bq = (d * oc).sum(-1)
            disc = bq * bq - dd * ((oc * oc).sum() - r * r)
            ts = (-bq - torch.sqrt(disc.clamp(min=0))) / dd
            hit = (disc > 0) & (ts > 1e-3) & (ts < t)
            t = torch.where(hit, ts, t)
            obj = torch.where(hit, torch.full_like(obj, k + 1), obj)
        p = o + d * t[..., None]
        ph = torch.tensor([0.0, 2.1, 4.2], dtype=dt, device=d.device)
        k1 = torch.tensor([[2.1, 0.7, 1.3], [0.9, 2.3, 0.5], [1.1, 0.6, 2.6]], dtype=dt, device=d.device)
        col = 0.5 + 0.35 * torch.sin(p @ k1.T + ph + obj[..., None].to(dt) * 1.7)
        return t, col.clamp(0, 1)


def trajectory(n_frames, bound, radius=0.25, yaw_step_deg=1.5, bob=0.05):
    """Camera-to-world matrices [n,4,4] (float32): a circle of `radius` around the room centre at `yaw_step_deg` per frame,
    looking outwards, with a small vertical bob and pitch.  World z is up; the camera looks along its -z, y up."""
    c = bound.double().mean(1)
    out = torch.eye(4, dtype=torch.float64).repeat(n_frames, 1, 1)
    for i in range(n_frames):
        yaw = math.radians(yaw_step_deg) * i
        pitch = 0.08 * math.sin(0.13 * i)
        fwd = torch.tensor([math.cos(yaw) * math.cos(pitch), math.sin(yaw) * math.cos(pitch), math.sin(pitch)])
        up0 = torch.tensor([0.0, 0.0, 1.0])
        right = torch.linalg.cross(fwd, up0)
        right = right / right.norm()
        up = torch.linalg.cross(right, fwd)
        out[i, :3, 0], out[i, :3, 1], out[i, :3, 2] = right.double(), up.double(), -fwd.double()
        out[i, :3, 3] = c + torch.tensor([radius * math.cos(yaw), radius * math.sin(yaw), bob * math.sin(0.21 * i)],
                                         dtype=torch.float64)
    return out.float()


def render_frame(room, sc, c2w, device="cpu", hole_frac=0.0, seed=0):
    """depth [H,W] float32 (0 in `hole_frac` of the pixels, as sensor holes), colour [H,W,3] float32."""
    dev = torch.device(device)
    i, j = torch.meshgrid(torch.arange(sc.W, dtype=torch.float32, device=dev),
                          torch.arange(sc.H, dtype=torch.float32, device=dev), indexing="xy")
    dirs = torch.stack([(i - sc.cx) / sc.fx, -(j - sc.cy) / sc.fy, -torch.ones_like(i)], -1)
    c2w = c2w.to(dev)
    d = dirs @ c2w[:3, :3].T
    t, col = room.cast(c2w[:3, 3], d)
    t = t.float()
    if hole_frac > 0:
        g = torch.Generator(device="cpu").manual_seed(seed)
        holes = (torch.rand(sc.H, sc.W, generator=g) < hole_frac).to(dev)
        t = torch.where(holes, torch.zeros_like(t), t)
    return t.contiguous(), col.float().contiguous()


def make_sequence(sc, n_frames, device="cpu", hole_frac=0.02, variant="r01"):
    """[(idx, colour [H,W,3], depth [H,W], gt_c2w [4,4])] - the tuples the reference's dataset readers yield
    (src/utils/datasets.py:107-148)."""
    room = AnalyticRoom(sc.bound, variant=variant)
    poses = trajectory(n_frames, sc.bound)
    frames = []
    for k in range(n_frames):
        depth, color = render_frame(room, sc, poses[k], device, hole_frac, seed=k)
        frames.append((k, color, depth, poses[k].to(device)))
    return frames
