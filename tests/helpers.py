"""Shared helpers for the parity tests: fixture loading and regeneration of hash-stream inputs."""
import os

import numpy as np
import torch

from myslam_amd import scene as scn
from myslam_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

RENDER_CASES = [
    "room0_200x32", "room0_200x40_noperturb", "room0_200x40_zero15", "room0_200x40_tracking", "room0_200x40_trained_zero15",
    "room0_4096x64", "scene0000_8192x96_zero10", "freiburg1_desk_5000x56_zero10", "room0_4096x64_trained_zero10",
]
SMALL_CASES = RENDER_CASES[:5]


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
    if "plane_scale" in fx.files and float(fx["plane_scale"]) != 1.0:      # the trained-like fixtures
        for grp in planes:
            for p in grp:
                p.mul_(float(fx["plane_scale"]))
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


# ---- samples whose gradient is legitimately ambiguous between two float32 evaluations ------------------------------------
def ambiguous_samples(p_nor, planes, params, eps=1e-5):
    """bool [N]: samples with a hidden-layer pre-activation a = W h + b of either decoder within float32 summation error of
    zero, |a| < eps (|W| |h| + |b|), in the float64 oracle.  Two correct float32 evaluations with different summation
    orders may put such a unit on different sides of its ReLU, and the sample's whole gradient contribution then differs by
    a finite amount - an error of neither.  About eps x 64 units = 6e-4 of the samples."""
    from oracle import eslam_oracle as orc
    amb = torch.zeros(p_nor.shape[0], dtype=torch.bool)
    for grp, c in ((planes[:3], ""), (planes[3:], "c_")):
        h = orc.plane_features(p_nor, *grp)
        for i in (0, 1):
            W, b = params[f"{c}linears.{i}.weight"], params[f"{c}linears.{i}.bias"]
            a = h @ W.t() + b
            amb |= (a.abs() < eps * (h.abs() @ W.abs().t() + b.abs())).any(-1)
            h = torch.relu(a)
    return amb


def texels_of(p_nor, plane_shapes):
    """For each of the 12 planes (all_planes order: group, level) a bool [N, h*w]-free structure: list of LongTensor [N,4]
    with the flat (y*w + x) indices of the four bilinear corners of every sample (border clamp as decoders.py:79-81)."""
    out = []
    axes = [(0, 1), (0, 2), (1, 2)] * 2
    for g, (ax, ay) in enumerate(axes):
        for lvl in range(2):
            h, w = plane_shapes[g][lvl][2:]
            ix = ((p_nor[:, ax] + 1) / 2 * (w - 1)).clamp(0, w - 1)
            iy = ((p_nor[:, ay] + 1) / 2 * (h - 1)).clamp(0, h - 1)
            x0, y0 = ix.floor().long(), iy.floor().long()
            x1, y1 = (x0 + 1).clamp(max=w - 1), (y0 + 1).clamp(max=h - 1)
            out.append(torch.stack([y0 * w + x0, y0 * w + x1, y1 * w + x0, y1 * w + x1], -1))
    return out


def plane_grads_close(mine, ref32, ref64, p_nor, amb, plane_shapes, rtol=1e-4):
    """Plane gradients (12 arrays [1,C,h,w], all_planes order) against the float32 oracle (comparator) with the float64 one
    as the conditioning bound, excluding the texels touched by ReLU-ambiguous samples (ambiguous_samples).  Returns
    (ok, message)."""
    tex = texels_of(p_nor, plane_shapes)
    for k, (a, r32, r64) in enumerate(zip(mine, ref32, ref64)):
        a, r32, r64 = (np.asarray(t, dtype=np.float64).reshape(a.shape[1], -1) for t in (a, r32, r64))
        keep = np.ones(a.shape[1], dtype=bool)
        if amb.any():
            keep[tex[k][amb].reshape(-1).numpy()] = False
        scale32, scale64 = np.abs(r32).max() + 1e-30, np.abs(r64).max() + 1e-30
        e32 = np.abs(a - r32)[:, keep].max(initial=0.0) / scale32
        e64 = np.abs(a - r64)[:, keep].max(initial=0.0) / scale64
        cond = np.abs(r32 - r64)[:, keep].max(initial=0.0) / scale64
        if e32 > rtol or e64 > max(rtol, 1.5 * cond):
            return False, f"plane {k}: vs float32 oracle {e32:.2e}, vs float64 {e64:.2e} (float32 vs float64 oracle {cond:.2e})"
    return True, ""
