"""Sample positions of the bench workload on the CPU for the scatter simulations (numpy only; deliberately independent
of oracle/, which is reserved for tests and bench.py's baseline leg).  Formulas: reference src/common.py:87-99 (pixel ->
ray), src/utils/Renderer.py:46-61,96-102 (depth-guided samples + jitter), src/common.py:204-218 (normalisation)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from myslam_amd import scene as scn, synth


def bench_samples(R=4096, ns=56, ni=8, scene="room0", truncation=0.06):
    sc = scn.make_scene(scene)
    S = ns + ni
    depth_img = synth.depth_image(sc.H, sc.W, 10)
    idx = synth.hash_randint(sc.H * sc.W, (R,), 50_000)
    c2w = scn.center_pose(sc).numpy().astype(np.float64)
    u, v = (idx % sc.W).astype(np.float64), (idx // sc.W).astype(np.float64)
    dirs = np.stack([(u - sc.cx) / sc.fx, -(v - sc.cy) / sc.fy, -np.ones_like(u)], -1)
    rd = dirs @ c2w[:3, :3].T
    ro = np.broadcast_to(c2w[:3, 3], rd.shape)
    gd = depth_img.reshape(-1)[idx].astype(np.float64)
    z_free = 1.2 * gd[:, None] * np.linspace(0.0, 1.0, ns)[None]
    z_surf = gd[:, None] - 1.5 * truncation + 3 * truncation * np.linspace(0.0, 1.0, ni)[None]
    z = np.sort(np.concatenate([z_free, z_surf], 1), 1)
    mids = 0.5 * (z[:, 1:] + z[:, :-1])
    lower, upper = np.concatenate([z[:, :1], mids], 1), np.concatenate([mids, z[:, -1:]], 1)
    z = lower + (upper - lower) * synth.hash_uniform((R, S), 90_000)
    pts = ro[:, None, :] + rd[:, None, :] * z[..., None]
    b = sc.bound.numpy().astype(np.float64)
    pn = (pts - b[:, 0]) / (b[:, 1] - b[:, 0]) * 2 - 1
    return sc, idx, ro, rd, z, pn
