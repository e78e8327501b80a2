"""Shared helpers for the parity tests: fixture loading and regeneration of hash-stream inputs."""
import os

import numpy as np
import torch

from myslam_amd import scene as scn
from myslam_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

RENDER_CASES = [
    "room0_200x32", "room0_200x40_noperturb", "room0_200x40_zero15", "room0_200x40_tracking",
    "room0_4096x64", "scene0000_8192x96_zero10", "freiburg1_desk_5000x56_zero10",
]
SMALL_CASES = RENDER_CASES[:4]


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def params_from(fx, dtype=torch.float32, device="cpu", requires_grad=False):
    out = {}
    for k in fx.files:
        if k.startswith("param:") and k != "param:beta":
            t = torch.from_numpy(fx[k]).to(dtype).to(device)
            out[k[6:]] = t.requires_grad_(requires_grad)
    return out


def rand_inputs(fx):
    """Rebuild the random tensors the reference consumed, expanded to full-batch rows.

    rand_calls records (kind, stream, *shape) in call order: randint (get_samples), then
    rand [R_gt,S] (Renderer.py:59 via :104), and when zero-depth rays exist rand [R0,n_strat]
    (:59 via :121) and rand [R0,n_imp] (common.py:59).
    """
    calls = [c.split(";") for c in fx["rand_calls"].tolist()]
    rands = [c for c in calls if c[0] == "rand"]
    out, k = rand_inputs_from(rands, fx["gt_depth"], int(fx["n_stratified"]), int(fx["n_importance"]), bool(fx["perturb"]))
    assert k == len(rands)
    return out


def rand_inputs_from(rands, gt_depth, ns, ni, perturb):
    """The (t_rand, t_uni, u) of ONE render_batch_ray call from the head of `rands` (a list of recorded rand calls);
    returns them and the number of calls consumed."""
    has = np.asarray(gt_depth) > 0
    R = has.shape[0]
    S = ns + ni
    t_rand = t_uni = u = None
    k = 0
    if perturb:
        c = rands[k]; k += 1
        assert (int(c[2]), int(c[3])) == (int(has.sum()), S), c
        t_rand = np.zeros((R, S), np.float32)
        t_rand[has] = synth.hash_uniform((int(c[2]), int(c[3])), int(c[1]))
    if not has.all():
        R0 = int((~has).sum())
        if perturb:
            c = rands[k]; k += 1
            assert (int(c[2]), int(c[3])) == (R0, ns), c
            t_uni = np.zeros((R, ns), np.float32)
            t_uni[~has] = synth.hash_uniform((R0, ns), int(c[1]))
        c = rands[k]; k += 1
        assert (int(c[2]), int(c[3])) == (R0, ni), c
        u = np.zeros((R, ni), np.float32)
        u[~has] = synth.hash_uniform((R0, ni), int(c[1]))
    cv = lambda a: None if a is None else torch.from_numpy(a)
    return (cv(t_rand), cv(t_uni), cv(u)), k


def render_img_chunk_rands(fx, gt_depth_flat):
    """Per chunk of Renderer.render_img (ray_batch_size rays each) the random tensors the reference drew for it."""
    rands = [c.split(";") for c in fx["rand_calls"].tolist()]
    bs, ns, ni = int(fx["ray_batch_size"]), int(fx["n_stratified"]), int(fx["n_importance"])
    out = []
    for i in range(0, gt_depth_flat.shape[0], bs):
        r, k = rand_inputs_from(rands, gt_depth_flat[i:i + bs], ns, ni, True)
        rands = rands[k:]
        out.append(r)
    assert not rands
    return out


def elementwise_close(a, b, rtol=1e-4, floor=1e-6):
    """|a - b| <= rtol |b| + floor max|b| for EVERY element (north_star: 1e-4 relative; the floor keeps elements that are
    near zero against the tensor's scale - an sdf crossing zero, a dim colour channel - from demanding absolute 1e-10)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    tol = rtol * np.abs(b) + floor * (np.abs(b).max() + 1e-30)
    bad = np.abs(a - b) > tol
    return not bad.any(), (int(bad.sum()), float((np.abs(a - b) / tol).max()))


def scene_and_planes(fx, device="cpu", dtype=torch.float32, channels_last=True, requires_grad=False):
    sc = scn.make_scene(str(fx["scene"]))
    planes = scn.synth_planes(sc, device=device, dtype=dtype, channels_last=channels_last)
    if requires_grad:
        planes = tuple([p.requires_grad_(True) for p in grp] for grp in planes)
    return sc, planes


def flat_planes(all_planes):
    return [p for grp in all_planes for p in grp]


def check_plane_probes(fx, grads, rtol=1e-4):
    """Compare plane gradients (list of 12 [1,C,h,w] tensors, any strides) with the fixture probes."""
    for k, g in enumerate(grads):
        flat = g.detach().cpu().double().contiguous().reshape(-1).numpy()   # logical NCHW order
        l2 = float(fx[f"pg{k}_l2"])
        l1 = float(fx[f"pg{k}_l1"])
        assert abs(np.sqrt((flat ** 2).sum()) - l2) <= rtol * l2 + 1e-12, f"plane {k} l2"
        assert abs(np.abs(flat).sum() - l1) <= rtol * l1 + 1e-12, f"plane {k} l1"
        idx = fx[f"pg{k}_idx"]
        ref = fx[f"pg{k}_val"].astype(np.float64)
        scale = np.abs(ref).max() + 1e-30
        err = np.abs(flat[idx] - ref).max()
        assert err <= rtol * scale, f"plane {k} probes: err {err} scale {scale}"


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))
