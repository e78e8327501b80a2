#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE implementation (build container only).

    python tests/golden/make_golden.py            # writes tests/golden/*.npz

This is the one place that imports /root/reference.  It never copies reference code: it loads the
reference's modules in place (with empty stand-ins for the third-party packages this image lacks and
the hot path does not use), runs its Renderer / Decoders / get_samples / Mapper.sdf_losses on
synthetic inputs, and stores *inputs and outputs* only.  The reference cannot travel to the GPU box,
the fixtures do.

Random numbers: the reference draws torch.rand / torch.randint internally.  While it runs we swap
those two callables for deterministic hash streams (myslam_amd.synth) so the tests can regenerate
the very same numbers without storing megabytes of them; call order is recorded in each fixture
(`rand_calls`).  Planes and the RGB-D image are regenerated from the same hash streams.
"""
import os
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True

from myslam_amd import scene as scn      # noqa: E402
from myslam_amd import synth             # noqa: E402


def _load_reference():
    def stub(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    p3d = stub("pytorch3d")
    p3d.transforms = stub("pytorch3d.transforms", matrix_to_quaternion=None, quaternion_to_matrix=None)
    empty = SimpleNamespace()
    col = type("C", (), {"__getattr__": lambda self, k: ""})()
    stub("colorama", Fore=col, Style=col)
    for name in ("cv2", "trimesh", "skimage", "skimage.measure"):
        stub(name)
    stub("open3d", __version__="0.13.0")
    sys.path.insert(0, REF)
    os.chdir(REF)
    from src import config                       # noqa
    from src.utils.Renderer import Renderer      # noqa
    from src.common import get_samples, sample_pdf, normalize_3d_coordinate, get_rays   # noqa
    from src.ESLAM import ESLAM                  # noqa
    from src.Mapper import Mapper                # noqa
    from src.Tracker import Tracker              # noqa
    return SimpleNamespace(config=config, Renderer=Renderer, get_samples=get_samples, sample_pdf=sample_pdf,
                           normalize_3d_coordinate=normalize_3d_coordinate, get_rays=get_rays,
                           ESLAM=ESLAM, Mapper=Mapper, Tracker=Tracker)


class HashRNG:
    """Context manager replacing torch.rand / torch.randint by hash streams."""

    def __init__(self, base):
        self.base = base
        self.calls = []

    def __enter__(self):
        self._rand, self._randint = torch.rand, torch.randint
        rng = self

        def rand(*size, **kw):
            shape = tuple(size[0]) if len(size) == 1 and not isinstance(size[0], int) else tuple(size)
            stream = rng.base + len(rng.calls)
            rng.calls.append(("rand", stream) + shape)
            return torch.from_numpy(synth.hash_uniform(shape, stream))

        def randint(high, size, **kw):
            stream = rng.base + len(rng.calls)
            rng.calls.append(("randint", stream, high) + tuple(size))
            return torch.from_numpy(synth.hash_randint(high, tuple(size), stream))

        torch.rand, torch.randint = rand, randint
        return self

    def __exit__(self, *a):
        torch.rand, torch.randint = self._rand, self._randint

    def calls_array(self):
        return np.array([";".join(map(str, c)) for c in self.calls])


SCENE_YAML = {"room0": "configs/Replica/room0.yaml", "scene0000": "configs/ScanNet/scene0000.yaml",
              "freiburg1_desk": "configs/TUM_RGBD/freiburg1_desk.yaml"}


def ref_setup(ref, scene_name, seed=0):
    cfg = ref.config.load_config(SCENE_YAML[scene_name], "configs/ESLAM.yaml")
    cfg["device"] = "cpu"
    ns = SimpleNamespace(cfg=cfg, device="cpu", scale=cfg["scale"])
    ns.H, ns.W, ns.fx, ns.fy, ns.cx, ns.cy = (cfg["cam"][k] for k in ("H", "W", "fx", "fy", "cx", "cy"))
    ref.ESLAM.update_cam(ns)
    torch.manual_seed(seed)
    ns.shared_decoders = ref.config.get_model(cfg)
    ref.ESLAM.load_bound(ns, cfg)
    ref.ESLAM.init_planes(ns, cfg)
    # cross-check our restated scene arithmetic against the reference's
    sc = scn.make_scene(scene_name)
    assert (sc.H, sc.W) == (ns.H, ns.W), (sc.H, sc.W, ns.H, ns.W)
    assert abs(sc.fx - ns.fx) < 1e-12 and abs(sc.cy - ns.cy) < 1e-12
    assert torch.equal(sc.bound, ns.bound), (sc.bound, ns.bound)
    ref_shapes = [[tuple(p.shape) for p in grp] for grp in
                  (ns.shared_planes_xy, ns.shared_planes_xz, ns.shared_planes_yz,
                   ns.shared_c_planes_xy, ns.shared_c_planes_xz, ns.shared_c_planes_yz)]
    assert ref_shapes == [[tuple(s) for s in g] for g in sc.plane_shapes], (ref_shapes, sc.plane_shapes)
    return cfg, ns, sc


def param_dict(decoders):
    return {k: v.detach().numpy().copy() for k, v in decoders.state_dict().items()}


def plane_probes(grads, stream):
    """Norms + sampled entries of each plane gradient (in logical NCHW flat order)."""
    out = {}
    for k, g in enumerate(grads):
        flat = g.reshape(-1).double().numpy()
        nz = np.flatnonzero(flat)
        out[f"pg{k}_l1"] = np.abs(flat).sum()
        out[f"pg{k}_l2"] = np.sqrt((flat ** 2).sum())
        out[f"pg{k}_nnz"] = np.int64(nz.size)
        if nz.size:
            pick = nz[synth.hash_randint(nz.size, (256,), stream + k)]
        else:
            pick = np.zeros(256, dtype=np.int64)
        anyw = synth.hash_randint(flat.size, (64,), stream + 100 + k)
        idx = np.concatenate([pick, anyw])
        out[f"pg{k}_idx"] = idx
        out[f"pg{k}_val"] = flat[idx].astype(np.float32)
    return out


def render_case(ref, name, scene_name, R, n_strat, n_imp, zero_frac, perturb=True, probe=None,
                rays_grad=True, rng_base=50_000, img_stream=10, loss_kind="mapping", plane_scale=1.0, sdf_bias=0.0):
    """plane_scale / sdf_bias: a "trained-like" state.  With the reference's initial state (planes ~ 0.01, sdf ~ 0) the
    first sample of a ray takes 99 % of the compositing weight, so depth, colour and their gradients say nothing about the
    colour features of later samples.  Planes scaled to O(1) features and the SDF head shifted positive spread the weights
    over ~10 samples with a different profile per ray (and give the importance sampler a non-trivial pdf)."""
    cfg, ns, sc = ref_setup(ref, scene_name)
    cfg["rendering"]["perturb"] = perturb
    renderer = ref.Renderer(cfg, ns)
    renderer.n_stratified, renderer.n_importance = n_strat, n_imp
    decoders = ns.shared_decoders
    trunc = cfg["model"]["truncation"]

    planes = scn.synth_planes(sc, channels_last=False)
    all_planes = tuple([torch.nn.Parameter(p * plane_scale) for p in grp] for grp in planes)
    if sdf_bias:
        with torch.no_grad():
            decoders.output_linear.bias += sdf_bias

    depth_img = torch.from_numpy(synth.depth_image(sc.H, sc.W, img_stream, zero_frac))[None]
    color_img = torch.from_numpy(synth.color_image(sc.H, sc.W, img_stream + 2))[None]
    c2w = scn.center_pose(sc)[None].clone()
    if rays_grad:
        c2w.requires_grad_(True)

    with HashRNG(rng_base) as rng:
        rays_o, rays_d, gt_depth, gt_color = ref.get_samples(
            0, sc.H, 0, sc.W, R, sc.H, sc.W, sc.fx, sc.fy, sc.cx, sc.cy, c2w, depth_img, color_img, "cpu")
        # Mapper.py:322-332 pre-filter, run as the caller does
        with torch.no_grad():
            t = (ns.bound.unsqueeze(0) - rays_o.detach().unsqueeze(-1)) / rays_d.detach().unsqueeze(-1)
            t, _ = torch.min(torch.max(t, dim=2)[0], dim=1)
            inside = t >= gt_depth
        rays_o_f, rays_d_f = rays_o[inside], rays_d[inside]
        gt_depth_f, gt_color_f = gt_depth[inside], gt_color[inside]
        ro = rays_o_f.detach().clone().requires_grad_(rays_grad)
        rd = rays_d_f.detach().clone().requires_grad_(rays_grad)
        depth, color, sdf, z = renderer.render_batch_ray(all_planes, decoders, rd, ro, "cpu", trunc,
                                                         gt_depth=gt_depth_f)
    w = cfg["mapping"] if loss_kind == "mapping" else cfg["tracking"]
    lself = SimpleNamespace(truncation=trunc, w_sdf_fs=w["w_sdf_fs"], w_sdf_center=w["w_sdf_center"],
                            w_sdf_tail=w["w_sdf_tail"])
    if loss_kind == "mapping":      # Mapper.py:337-346
        m = gt_depth_f > 0
        loss = ref.Mapper.sdf_losses(lself, sdf[m], z[m], gt_depth_f[m])
        loss = loss + w["w_color"] * torch.square(gt_color_f - color).mean()
        loss = loss + w["w_depth"] * torch.square(gt_depth_f[m] - depth[m]).mean()
    else:                            # Tracker.py:192-204
        err = (gt_depth_f - depth.detach()).abs()
        m = err < 10 * err.median()
        loss = ref.Tracker.sdf_losses(lself, sdf[m], z[m], gt_depth_f[m])
        loss = loss + w["w_color"] * torch.square(gt_color_f - color)[m].mean()
        loss = loss + w["w_depth"] * torch.square(gt_depth_f[m] - depth[m]).mean()
    loss.backward()

    Re = int(ro.shape[0])
    pr = np.arange(Re) if probe is None else np.linspace(0, Re - 1, probe).astype(np.int64)
    fx = dict(
        scene=scene_name, R=np.int64(R), R_eff=np.int64(Re), n_stratified=np.int64(n_strat),
        n_importance=np.int64(n_imp), zero_frac=np.float64(zero_frac), perturb=np.bool_(perturb),
        truncation=np.float64(trunc), img_stream=np.int64(img_stream), rng_base=np.int64(rng_base),
        loss_kind=loss_kind, rand_calls=rng.calls_array(), plane_scale=np.float64(plane_scale),
        beta_is_param=np.bool_(isinstance(decoders.beta, torch.nn.Parameter)),
        beta=np.float32(float(decoders.beta)),
        rays_o=ro.detach().numpy(), rays_d=rd.detach().numpy(), gt_depth=gt_depth_f.numpy(),
        gt_color=gt_color_f.numpy(),
        probe=pr, depth=depth.detach().numpy()[pr], color=color.detach().numpy()[pr],
        sdf=sdf.detach().numpy()[pr], z_vals=z.detach().numpy()[pr],
        depth_sum=depth.detach().double().sum().numpy(), color_sum=color.detach().double().sum().numpy(),
        sdf_sum=sdf.detach().double().sum().numpy(), z_sum=z.detach().double().sum().numpy(),
        sdf_l2=sdf.detach().double().norm().numpy(),
        loss=loss.detach().double().numpy(),
    )
    for k, v in param_dict(decoders).items():
        fx["param:" + k] = v
    for k, p in decoders.named_parameters():
        fx["grad:" + k] = p.grad.numpy().copy()
    if rays_grad:
        fx["g_rays_o"] = ro.grad.numpy()[pr]
        fx["g_rays_d"] = rd.grad.numpy()[pr]
        fx["g_rays_o_sum"] = ro.grad.double().sum(0).numpy()
        fx["g_rays_d_sum"] = rd.grad.double().sum(0).numpy()
    grads = [p.grad for grp in all_planes for p in grp]
    fx.update(plane_probes(grads, 77_000))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **fx)
    print(f"{name}: R_eff={Re} loss={float(loss):.6f} zero-depth rays={(gt_depth_f == 0).sum().item()}")


def get_samples_case(ref, name="get_samples_room0_b3"):
    """a1 with three cameras, a crop window and gradients w.r.t. c2ws (Tracker.py:169-172 style window)."""
    cfg, ns, sc = ref_setup(ref, "room0")
    b, n = 3, 50
    H0, H1, W0, W1 = 75, sc.H - 75, 75, sc.W - 75
    depth_img = torch.from_numpy(np.stack([synth.depth_image(sc.H, sc.W, 20 + i) for i in range(b)]))
    color_img = torch.from_numpy(np.stack([synth.color_image(sc.H, sc.W, 30 + i) for i in range(b)]))
    # three proper rotations from hash numbers (QR of a random matrix), translations inside the AABB
    c2ws = torch.eye(4).repeat(b, 1, 1)
    for i in range(b):
        m = synth.hash_uniform((3, 3), 40 + i).astype(np.float64) - 0.5
        q, r = np.linalg.qr(m)
        q = q * np.sign(np.diag(r))
        if np.linalg.det(q) < 0:
            q[:, 0] = -q[:, 0]
        c2ws[i, :3, :3] = torch.from_numpy(q).float()
        c2ws[i, :3, 3] = ns.bound.mean(1) + torch.from_numpy(synth.hash_uniform((3,), 45 + i)) - 0.5
    c2ws.requires_grad_(True)
    with HashRNG(60_000) as rng:
        ro, rd, d, c = ref.get_samples(H0, H1, W0, W1, n, sc.H, sc.W, sc.fx, sc.fy, sc.cx, sc.cy,
                                       c2ws, depth_img, color_img, "cpu")
    wo = torch.from_numpy(synth.hash_uniform(tuple(ro.shape), 61_000)) - 0.5
    wd = torch.from_numpy(synth.hash_uniform(tuple(rd.shape), 61_001)) - 0.5
    ((ro * wo).sum() + (rd * wd).sum()).backward()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), b=np.int64(b), n=np.int64(n),
                        window=np.array([H0, H1, W0, W1]), c2ws=c2ws.detach().numpy(),
                        rand_calls=rng.calls_array(), rays_o=ro.detach().numpy(), rays_d=rd.detach().numpy(),
                        depth=d.numpy(), color=c.numpy(), g_c2ws=c2ws.grad.numpy())
    # full-image rays (render_img path, common.py:183-201)
    ro_i, rd_i = ref.get_rays(sc.H, sc.W, sc.fx, sc.fy, sc.cx, sc.cy, c2ws[1].detach(), "cpu")
    sel = np.linspace(0, sc.H * sc.W - 1, 512).astype(np.int64)
    np.savez_compressed(os.path.join(OUT, "get_rays_room0.npz"), c2w=c2ws[1].detach().numpy(), sel=sel,
                        rays_o=ro_i.reshape(-1, 3).numpy()[sel], rays_d=rd_i.reshape(-1, 3).numpy()[sel])
    print(name, "ok")


def decoder_case(ref, name="decoders_room0_points"):
    """Decoders.forward on free points, a tenth of them outside the AABB (border clamp), + sample_pdf."""
    cfg, ns, sc = ref_setup(ref, "room0")
    decoders = ns.shared_decoders
    planes = scn.synth_planes(sc, channels_last=False)
    all_planes = tuple(list(grp) for grp in planes)
    N = 2000
    u = torch.from_numpy(synth.hash_uniform((N, 3), 70_000))
    lo, hi = ns.bound[:, 0], ns.bound[:, 1]
    p = lo + (hi - lo) * (u * 1.2 - 0.1)          # 10 % margin outside on each side
    p[0] = lo                                       # exact corners
    p[1] = hi
    with torch.no_grad():
        raw = decoders(p, all_planes=all_planes)
        pn = ref.normalize_3d_coordinate(p.clone(), ns.bound)
    # sample_pdf known answers
    bins = torch.sort(torch.from_numpy(synth.hash_uniform((40, 23), 70_001)), -1).values * 3
    wts = torch.from_numpy(synth.hash_uniform((40, 22), 70_002))
    wts[3] = 0                                      # all-zero weights row -> denom guard
    wts[5, :10] = 0
    with HashRNG(70_010) as rng:
        smp = ref.sample_pdf(bins, wts, 8, det=False, device="cpu")
    fx = dict(points=p.numpy(), raw=raw.numpy(), p_nor=pn.numpy(), bins=bins.numpy(), weights=wts.numpy(),
              pdf_samples=smp.numpy(), rand_calls=rng.calls_array())
    for k, v in param_dict(decoders).items():
        fx["param:" + k] = v
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **fx)
    print(name, "ok")


def render_img_case(ref, name="render_img_room0_30x44"):
    """Renderer.render_img (src/utils/Renderer.py:155-204) on a small image: three chunks of the reference's ray batching
    with a ragged last one, 5 % of the pixels without depth (importance branch), perturbation on, a rotated camera.
    Outputs stored in full: depth [H,W] float64, colour [H,W,3]."""
    cfg, ns, sc = ref_setup(ref, "room0")
    H, W = 30, 44
    ns.H, ns.W, ns.fx, ns.fy, ns.cx, ns.cy = H, W, 22.0, 22.0, (W - 1) / 2, (H - 1) / 2
    cfg["rendering"]["perturb"] = True
    renderer = ref.Renderer(cfg, ns, ray_batch_size=500)
    renderer.n_stratified, renderer.n_importance = 16, 8
    decoders = ns.shared_decoders
    trunc = cfg["model"]["truncation"]
    planes = scn.synth_planes(sc, channels_last=False)
    all_planes = tuple(list(grp) for grp in planes)
    c2w = keyframe_poses(1, sc, stream=810)[0]
    gt_depth = torch.from_numpy(synth.depth_image(H, W, 820, 0.05))
    with HashRNG(830) as rng:
        depth, color = renderer.render_img(all_planes, decoders, c2w, trunc, "cpu", gt_depth=gt_depth)
    assert depth.dtype == torch.float64 and tuple(depth.shape) == (H, W) and tuple(color.shape) == (H, W, 3)
    fx = dict(H=np.int64(H), W=np.int64(W), fx=np.float64(ns.fx), fy=np.float64(ns.fy), cx=np.float64(ns.cx),
              cy=np.float64(ns.cy), ray_batch_size=np.int64(500), n_stratified=np.int64(16), n_importance=np.int64(8),
              truncation=np.float64(trunc), c2w=c2w.numpy(), depth_stream=np.int64(820), zero_frac=np.float64(0.05),
              rand_calls=rng.calls_array(), depth=depth.numpy(), color=color.numpy(),
              n_zero=np.int64(int((gt_depth == 0).sum())))
    for k, v in param_dict(decoders).items():
        fx["param:" + k] = v
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **fx)
    print(name, "ok: zero-depth pixels", int((gt_depth == 0).sum()), "rand calls", len(rng.calls))


def tum_lists(n_rgb=60, stream=900):
    """Synthetic TUM-format lists shared by the fixture and the tests: colour frames at ~30 Hz with jittered timestamps and a
    gap, depth frames at a slightly different phase (some further than max_dt from any colour frame), poses at ~100 Hz."""
    u = synth.hash_uniform((n_rgb,), stream).astype(np.float64)
    t_rgb = 1305031000.0 + np.arange(n_rgb) / 30.0 + 0.004 * (u - 0.5)
    t_rgb[40:] += 0.5                                      # a pause in the recording
    v = synth.hash_uniform((n_rgb,), stream + 1).astype(np.float64)
    t_dep = t_rgb + 0.02 + 0.13 * (v > 0.85)               # 15 % of the depth frames are 0.15 s late: no partner within 0.08 s
    t_dep = np.delete(t_dep, [7, 8])                       # and two are missing
    n_pose = 400
    t_pose = t_rgb[0] - 0.05 + np.arange(n_pose) / 100.0   # ends before the last colour frames: those lose their pose
    q = synth.hash_uniform((n_pose, 4), stream + 2).astype(np.float64) - 0.5
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    tr = synth.hash_uniform((n_pose, 3), stream + 3).astype(np.float64) * 2 - 1
    return t_rgb, t_dep, t_pose, np.concatenate([tr, q], 1)


def write_tum_lists(folder, t_rgb, t_dep, t_pose, vecs):
    with open(os.path.join(folder, "rgb.txt"), "w") as f:
        f.write("".join(f"{t:.6f} rgb/{t:.6f}.png\n" for t in t_rgb))
    with open(os.path.join(folder, "depth.txt"), "w") as f:
        f.write("".join(f"{t:.6f} depth/{t:.6f}.png\n" for t in t_dep))
    with open(os.path.join(folder, "groundtruth.txt"), "w") as f:
        f.write("# timestamp tx ty tz qx qy qz qw\n")
        f.write("".join(f"{t:.4f} " + " ".join(f"{x:.6f}" for x in v) + "\n" for t, v in zip(t_pose, vecs)))


def datasets_case(ref, name="datasets_lists"):
    """The parts of the reference's dataset readers that do not need OpenCV (src/utils/datasets.py): TUM_RGBD.loadtum
    (timestamp association, frame-rate thinning, re-basing, flip) on synthetic lists, Replica.load_poses and
    ScanNet.load_poses on synthetic trajectories.  numpy 2 removed the np.unicode_ alias the reference's parse_list names:
    it is restored for the duration of the call."""
    import tempfile
    from src.utils import datasets as rds
    t_rgb, t_dep, t_pose, vecs = tum_lists()
    out = {}
    with tempfile.TemporaryDirectory() as d:
        write_tum_lists(d, t_rgb, t_dep, t_pose, vecs)
        ns = SimpleNamespace()
        ns.parse_list = lambda path, skiprows=0: rds.TUM_RGBD.parse_list(ns, path, skiprows)
        ns.associate_frames = lambda a, b, c, max_dt=0.08: rds.TUM_RGBD.associate_frames(ns, a, b, c, max_dt)
        ns.pose_matrix_from_quaternion = lambda v: rds.TUM_RGBD.pose_matrix_from_quaternion(ns, v)
        had = hasattr(np, "unicode_")
        if not had:
            np.unicode_ = np.str_
        try:
            images, depths, poses = rds.TUM_RGBD.loadtum(ns, d, frame_rate=32)
        finally:
            if not had:
                del np.unicode_
        out["tum_images"] = np.array([os.path.relpath(p, d) for p in images])
        out["tum_depths"] = np.array([os.path.relpath(p, d) for p in depths])
        out["tum_poses"] = torch.stack(poses).numpy()
        # Replica traj.txt (16 numbers per line) and ScanNet pose/<n>.txt (4 lines of 4)
        K = 12
        mats = keyframe_poses(K, scn.make_scene("room0"), stream=950).double().numpy()
        with open(os.path.join(d, "traj.txt"), "w") as f:
            f.write("".join(" ".join(f"{x:.9e}" for x in m.reshape(-1)) + "\n" for m in mats))
        rns = SimpleNamespace(n_img=K)
        rds.Replica.load_poses(rns, os.path.join(d, "traj.txt"))
        out["replica_poses"] = torch.stack(rns.poses).numpy()
        os.makedirs(os.path.join(d, "pose"))
        for k, m in enumerate(mats):
            with open(os.path.join(d, "pose", f"{k}.txt"), "w") as f:
                f.write("".join(" ".join(f"{x:.6f}" for x in row) + "\n" for row in m))
        sns = SimpleNamespace()
        rds.ScanNet.load_poses(sns, os.path.join(d, "pose"))
        out["scannet_poses"] = torch.stack(sns.poses).numpy()
        out["mats"] = mats
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, "ok:", len(out["tum_images"]), "TUM frames kept of", len(t_rgb))


def keyframe_poses(K, sc, stream=700):
    """K camera poses inside the scene: yaw angles all round the compass (so some views overlap the current frame and
    some look away), small pitch, translation within 1 m of the AABB centre.  Shared by the fixture and the tests."""
    out = torch.eye(4).repeat(K, 1, 1)
    ang = synth.hash_uniform((K, 2), stream).astype(np.float64)
    tr = synth.hash_uniform((K, 3), stream + 1).astype(np.float64)
    for k in range(K):
        yaw, pitch = 2 * np.pi * ang[k, 0], 0.4 * (ang[k, 1] - 0.5)
        Ry = np.array([[np.cos(yaw), 0, np.sin(yaw)], [0, 1, 0], [-np.sin(yaw), 0, np.cos(yaw)]])
        Rx = np.array([[1, 0, 0], [0, np.cos(pitch), -np.sin(pitch)], [0, np.sin(pitch), np.cos(pitch)]])
        out[k, :3, :3] = torch.from_numpy(Ry @ Rx).float()
        out[k, :3, 3] = sc.bound.mean(1) + torch.from_numpy(tr[k] - 0.5).float() * 2.0
    return out


def keyframe_overlap_case(ref, name="keyframe_overlap_room0"):
    """Mapper.keyframe_selection_overlap (src/Mapper.py:146-209) run as an unbound method on a namespace; randperm is
    swapped for argsort of a hash stream so that the selection order is reproducible."""
    sc = scn.make_scene("room0")
    K = 26                                         # the method ignores the last two (already in the window)
    c2ws = keyframe_poses(K, sc)
    cur = keyframe_poses(1, sc, stream=710)[0]
    depth = torch.from_numpy(synth.depth_image(sc.H, sc.W, 720, 0.1))
    color = torch.from_numpy(synth.color_image(sc.H, sc.W, 721))
    ns = SimpleNamespace(device="cpu", H=sc.H, W=sc.W, fx=sc.fx, fy=sc.fy, cx=sc.cx, cy=sc.cy,
                         estimate_c2w_list=c2ws, keyframe_list=list(range(K)))
    real_randperm = torch.randperm
    out = {}
    try:
        torch.randperm = lambda n, **kw: torch.from_numpy(np.argsort(synth.hash_uniform((n,), 730), kind="stable"))
        for label, num in (("all", K), ("four", 4)):
            with HashRNG(740) as rng:
                sel = ref.Mapper.keyframe_selection_overlap(ns, color, depth, cur, num)
            out["selected_" + label] = np.array([int(i) for i in sel], dtype=np.int64)
            out["rand_calls"] = rng.calls_array()
    finally:
        torch.randperm = real_randperm
    assert 0 < len(out["selected_all"]) < K - 2, out["selected_all"]
    np.savez_compressed(os.path.join(OUT, name + ".npz"), K=K, cur_c2w=cur.numpy(), c2ws=c2ws.numpy(),
                        num_rays=50, num_samples=8, **out)
    print(name, "ok", out["selected_all"], out["selected_four"])


def main():
    ref = _load_reference()
    os.makedirs(OUT, exist_ok=True)
    only = set(sys.argv[1:])

    def want(n):
        return not only or n in only

    if want("get_samples"):
        get_samples_case(ref)
    if want("decoders"):
        decoder_case(ref)
    if want("keyframe_overlap"):
        keyframe_overlap_case(ref)
    if want("render_img"):
        render_img_case(ref)
    if want("datasets"):
        datasets_case(ref)
    # BASELINE.json configs[0]: 200 rays x 32 samples (24+8), CPU plumbing case - stored in full
    if want("room0_200x32"):
        render_case(ref, "room0_200x32", "room0", 200, 24, 8, 0.0)
    # same size, reference's own Replica sample counts, perturbation off (closed-form z)
    if want("room0_200x40_noperturb"):
        render_case(ref, "room0_200x40_noperturb", "room0", 200, 32, 8, 0.0, perturb=False)
    # importance branch: 15 % of the rays without depth
    if want("room0_200x40_zero15"):
        render_case(ref, "room0_200x40_zero15", "room0", 200, 32, 8, 0.15)
    # tracking loss (median mask), pose gradients only matter there
    if want("room0_200x40_tracking"):
        render_case(ref, "room0_200x40_tracking", "room0", 200, 32, 8, 0.0, loss_kind="tracking")
    # a trained-like state (compositing weights spread over many samples), 15 % of the rays without depth
    if want("room0_200x40_trained_zero15"):
        render_case(ref, "room0_200x40_trained_zero15", "room0", 200, 32, 8, 0.15, plane_scale=60.0, sdf_bias=0.55)
    # BASELINE.json configs[1]: 4096 x 64 (56+8) - 64-ray probe + checksums
    if want("room0_4096x64"):
        render_case(ref, "room0_4096x64", "room0", 4096, 56, 8, 0.0, probe=64)
    # the bench shape in the trained-like state, 10 % of the rays without depth
    if want("room0_4096x64_trained_zero10"):
        render_case(ref, "room0_4096x64_trained_zero10", "room0", 4096, 56, 8, 0.10, probe=64, plane_scale=60.0, sdf_bias=0.55)
    # BASELINE.json configs[3]: scene0000, 8192 x 96 (88+8), 10 % zero-depth
    if want("scene0000_8192x96_zero10"):
        render_case(ref, "scene0000_8192x96_zero10", "scene0000", 8192, 88, 8, 0.10, probe=64)
    # BASELINE.json configs[4] shape: freiburg1_desk, 5000 x 56, beta is a Python int there
    if want("freiburg1_desk_5000x56_zero10"):
        render_case(ref, "freiburg1_desk_5000x56_zero10", "freiburg1_desk", 5000, 48, 8, 0.10, probe=64)


if __name__ == "__main__":
    main()
