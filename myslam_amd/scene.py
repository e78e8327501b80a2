"""Scene geometry for the renderer harness: camera intrinsics, scene bound and tri-plane shapes.

The renderer itself is scene-agnostic; this module exists because plane *shapes* are a function of
the reference's start-up arithmetic (float32 rounding included) and the bench / tests must build
planes of exactly the shapes the reference would.  It restates three reference methods:

  * ESLAM.update_cam   (reference src/ESLAM.py:135-157)  - crop_size rescale, crop_edge shrink
  * ESLAM.load_bound   (reference src/ESLAM.py:159-173)  - round each axis up to a multiple of
                                                           bound_dividable, in float32
  * ESLAM.init_planes  (reference src/ESLAM.py:175-218)  - grid_shape = int(len / res) per axis,
                                                           xy->[1,C,Y,X], xz->[1,C,Z,X], yz->[1,C,Z,Y]

Scene constants are the values of the reference's YAML files (configs/ESLAM.yaml,
configs/Replica/room0.yaml, configs/ScanNet/{scannet,scene0000}.yaml,
configs/TUM_RGBD/{tum,freiburg1_desk}.yaml); the YAMLs themselves do not travel to the GPU box.

Planes are allocated with torch.channels_last strides: same logical [1, C, h, w] shape the reference
uses (so callers index them identically), but one texel's C=32 channels are 128 contiguous bytes, which
is what the HIP gather/scatter kernels coalesce on.  NCHW-contiguous planes are accepted too (slower).
"""
from dataclasses import dataclass, field
from typing import List, Tuple

import numpy as np
import torch

from . import synth

# configs/ESLAM.yaml:5-11,71-78
PLANES_RES = dict(coarse=0.24, fine=0.06, bound_dividable=0.24)
C_PLANES_RES = dict(coarse=0.24, fine=0.03)
C_DIM = 32
TRUNCATION = 0.06

_SCENES = {
    # configs/Replica/room0.yaml:3 ; configs/Replica/replica.yaml:16-26
    "room0": dict(
        bound=[[-1.9, 7.9], [-2.2, 4.5], [-2.5, 2.3]],
        cam=dict(H=680, W=1200, fx=600.0, fy=600.0, cx=599.5, cy=339.5, crop_edge=0),
        n_stratified=32, n_importance=8, learnable_beta=True,
        mapping_pixels=4000, tracking_pixels=2000, ignore_edge=75,
    ),
    # configs/ScanNet/scene0000.yaml:3 ; configs/ScanNet/scannet.yaml:15-27
    "scene0000": dict(
        bound=[[-2.0, 11.0], [-2.0, 11.5], [-2.0, 5.5]],
        cam=dict(H=480, W=640, fx=577.590698, fy=578.729797, cx=318.905426, cy=242.683609, crop_edge=10),
        n_stratified=48, n_importance=8, learnable_beta=True,
        mapping_pixels=4000, tracking_pixels=2000, ignore_edge=75,
    ),
    # configs/TUM_RGBD/freiburg1_desk.yaml:3,8-17 ; configs/TUM_RGBD/tum.yaml:17-29
    "freiburg1_desk": dict(
        bound=[[-4.6, 2.6], [-3.3, 3.2], [-2.0, 4.9]],
        cam=dict(H=480, W=640, fx=517.3, fy=516.5, cx=318.6, cy=255.3, crop_edge=8, crop_size=[384, 512]),
        n_stratified=48, n_importance=8, learnable_beta=False,
        mapping_pixels=5000, tracking_pixels=5000, ignore_edge=20,
    ),
    # NOT a reference scene: a 4 x 3 x 2.4 m synthetic room with a 240 x 320 camera for the end-to-end quality runs
    # (myslam_amd/synthscene.py, tests/test_gpu_slam_quality.py); planes total ~2.3 MB so the CPU oracle can keep up
    "toy": dict(
        bound=[[-2.0, 2.0], [-1.5, 1.5], [-1.2, 1.2]],
        cam=dict(H=240, W=320, fx=160.0, fy=160.0, cx=159.5, cy=119.5, crop_edge=0),
        n_stratified=32, n_importance=8, learnable_beta=True,
        mapping_pixels=1000, tracking_pixels=500, ignore_edge=10,
    ),
}

PLANE_NAMES = ("planes_xy", "planes_xz", "planes_yz", "c_planes_xy", "c_planes_xz", "c_planes_yz")


@dataclass
class Scene:
    name: str
    H: int
    W: int
    fx: float
    fy: float
    cx: float
    cy: float
    bound: torch.Tensor                      # [3,2] float32, CPU (reference keeps decoders.bound on CPU, ESLAM.py:173)
    plane_shapes: List[List[Tuple[int, ...]]] = field(default_factory=list)  # [6][2] of (1,C,h,w)
    n_stratified: int = 32
    n_importance: int = 8
    learnable_beta: bool = True
    truncation: float = TRUNCATION
    scale: float = 1.0

    @property
    def plane_bytes(self):
        return sum(int(np.prod(s)) * 4 for group in self.plane_shapes for s in group)

    def cfg(self, perturb=True):
        """The slice of the reference config dict that Renderer.__init__ reads (Renderer.py:37-41)."""
        return {
            "rendering": {"perturb": perturb, "n_stratified": self.n_stratified,
                          "n_importance": self.n_importance, "learnable_beta": self.learnable_beta},
            "scale": self.scale,
            "model": {"c_dim": C_DIM, "truncation": self.truncation},
        }


def update_cam(cam):
    """Reference src/ESLAM.py:135-157."""
    H, W, fx, fy, cx, cy = cam["H"], cam["W"], cam["fx"], cam["fy"], cam["cx"], cam["cy"]
    if "crop_size" in cam:
        ch, cw = cam["crop_size"]
        sx, sy = cw / W, ch / H
        fx, fy, cx, cy = sx * fx, sy * fy, sx * cx, sy * cy
        W, H = cw, ch
    e = cam.get("crop_edge", 0)
    if e > 0:
        H -= 2 * e
        W -= 2 * e
        cx -= e
        cy -= e
    return H, W, fx, fy, cx, cy


def load_bound(bound, scale=1.0, bound_dividable=PLANES_RES["bound_dividable"]):
    """Reference src/ESLAM.py:166-172: float32 tensor arithmetic, truncating int cast."""
    b = torch.from_numpy(np.array(bound) * scale).float()
    b[:, 1] = (((b[:, 1] - b[:, 0]) / bound_dividable).int() + 1) * bound_dividable + b[:, 0]
    return b


def plane_shapes(bound, c_dim=C_DIM, planes_res=PLANES_RES, c_planes_res=C_PLANES_RES):
    """Reference src/ESLAM.py:185-210.  Returns [6][2] shapes in all_planes order (PLANE_NAMES)."""
    xyz_len = bound[:, 1] - bound[:, 0]

    def level_shapes(res):
        gx, gy, gz = (int(v) for v in (xyz_len / res).tolist())      # float32 divide, truncate
        return (1, c_dim, gy, gx), (1, c_dim, gz, gx), (1, c_dim, gz, gy)   # xy, xz, yz

    out = [[], [], [], [], [], []]
    for res in (planes_res["coarse"], planes_res["fine"]):
        for k, s in enumerate(level_shapes(res)):
            out[k].append(s)
    for res in (c_planes_res["coarse"], c_planes_res["fine"]):
        for k, s in enumerate(level_shapes(res)):
            out[3 + k].append(s)
    return out


def make_scene(name):
    s = _SCENES[name]
    H, W, fx, fy, cx, cy = update_cam(s["cam"])
    bound = load_bound(s["bound"])
    return Scene(name=name, H=H, W=W, fx=fx, fy=fy, cx=cx, cy=cy, bound=bound,
                 plane_shapes=plane_shapes(bound), n_stratified=s["n_stratified"],
                 n_importance=s["n_importance"], learnable_beta=s["learnable_beta"])


def new_plane(shape, device="cpu", dtype=torch.float32):
    """Uninitialised [1,C,h,w] plane with channels-last strides."""
    return torch.empty(shape, device=device, dtype=dtype).contiguous(memory_format=torch.channels_last)


def synth_planes(scene, device="cpu", dtype=torch.float32, stream0=1000, channels_last=True):
    """all_planes 6-tuple of [coarse, fine] lists, filled with synth.plane_fill (bit-reproducible)."""
    groups = []
    k = 0
    for g in scene.plane_shapes:
        lst = []
        for shp in g:
            t = torch.from_numpy(synth.plane_fill(shp, stream0 + k)).to(dtype)
            if channels_last:
                t = t.contiguous(memory_format=torch.channels_last)
            lst.append(t.to(device))
            k += 1
        groups.append(lst)
    return tuple(groups)


def random_planes(scene, device, generator=None, std=0.01, channels_last=True):
    """all_planes drawn N(0, std^2) on `device`, as the reference's init_planes does (ESLAM.py:201-210)."""
    groups = []
    for g in scene.plane_shapes:
        lst = []
        for shp in g:
            t = new_plane(shp, device) if channels_last else torch.empty(shp, device=device)
            t.normal_(mean=0.0, std=std, generator=generator)
            lst.append(t)
        groups.append(lst)
    return tuple(groups)


def center_pose(scene, dtype=torch.float32):
    """c2w with identity rotation at the AABB centre."""
    c2w = torch.eye(4, dtype=dtype)
    c2w[:3, 3] = scene.bound.to(dtype).mean(dim=1)
    return c2w
