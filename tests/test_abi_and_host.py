"""CPU-side checks (no GPU needed): the C-ABI library loads and exports every symbol include/eslam_hip.h declares,
the ctypes prototypes cover them, the product path refuses CPU tensors, scene arithmetic reproduces the reference's
plane shapes, and the ray-sharded data-parallel scheme (global loss denominators + one flat gradient all-reduce) is
exact - exercised with two gloo ranks on the CPU, with the oracle standing in for the kernels.
"""
import ctypes
import os
import re
import socket

import numpy as np
import pytest
import torch

from tests import helpers as hp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "eslam_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(eslam_[a-z0-9_]+)\s*\(", src)) - {"eslam_plane_t", "eslam_decoders_t"})


def test_library_exports_every_declared_symbol():
    from myslam_amd import _hip
    lib = _hip.load_library()
    names = _header_functions()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/eslam_hip.h but not exported"
        assert n in _hip.SIGNATURES, f"{n} has no ctypes prototype in myslam_amd/_hip.py"
    assert lib.eslam_abi_version() == _hip.ABI_VERSION == 5
    assert lib.eslam_bwd_workspace_bytes(262144) > 262144 * 128 * 4
    assert lib.eslam_bwd_workspace_bytes(-1) == -1


def test_struct_layout_matches_header():
    from myslam_amd import _hip
    assert ctypes.sizeof(_hip.PlaneDesc) == 8 + 8 + 4 + 4 + 3 * 8 + 8
    assert ctypes.sizeof(_hip.DecodersDesc) == 13 * 8
    assert ctypes.sizeof(_hip.AdamTensor) == 4 * 8 + 8 + 8
    assert _hip.N_DEC_PARAMS == 2 * (16 * 64 + 16 + 16 * 16 + 16) + 17 + 51
    # the ray-order buffer (ABI 5): the binding's size formula must be the header's macro
    import re
    hdr = open(os.path.join(ROOT, "include", "eslam_hip.h")).read()
    assert int(re.search(r"#define ESLAM_RAY_ORDERS (\d+)", hdr).group(1)) == _hip.RAY_ORDERS
    m = re.search(r"#define ESLAM_RAY_ORDER_WORDS\(R\) \(ESLAM_RAY_ORDERS \* \(int64_t\)\(R\) \+ (\d+)\)", hdr)
    assert m and _hip.ray_order_words(1000) == _hip.RAY_ORDERS * 1000 + int(m.group(1))


def test_missing_library_fails_loudly(tmp_path):
    from myslam_amd import _hip
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _hip.load_library(str(tmp_path / "nope.so"))


def test_product_path_rejects_cpu_tensors():
    from types import SimpleNamespace
    from myslam_amd import scene as scn
    from myslam_amd.src.networks.decoders import Decoders
    from myslam_amd.src.utils.Renderer import Renderer
    sc = scn.make_scene("room0")
    r = Renderer(sc.cfg(), SimpleNamespace(bound=sc.bound, device="cpu", H=sc.H, W=sc.W, fx=sc.fx, fy=sc.fy, cx=sc.cx,
                                           cy=sc.cy))
    dec = Decoders()
    dec.bound = sc.bound
    planes = tuple([torch.zeros(s) for s in grp] for grp in sc.plane_shapes)
    ro = torch.zeros(4, 3)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        r.render_batch_ray(planes, dec, ro + 1, ro, "cpu", 0.06, gt_depth=torch.ones(4))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        dec(ro, all_planes=planes)
    with pytest.raises(AttributeError):            # reference dereferences gt_depth=None (Renderer.py:91)
        r.render_batch_ray(planes, dec, ro + 1, ro, "cpu", 0.06)


def test_adam_host_side():
    """optim.Adam mirrors torch.optim.Adam's constructor / param_groups / state_dict (Mapper.py:291-306) and has no
    CPU implementation."""
    from myslam_amd import optim
    a, b = torch.nn.Parameter(torch.zeros(3, 4)), torch.nn.Parameter(torch.zeros(5))
    opt = optim.Adam([{"params": [a], "lr": 0}, {"params": [b], "lr": 0}])
    ref = torch.optim.Adam([{"params": [a], "lr": 0}, {"params": [b], "lr": 0}])
    opt.param_groups[0]["lr"] = 0.005
    ref.param_groups[0]["lr"] = 0.005
    keys = lambda o: {k: v for k, v in o.state_dict()["param_groups"][0].items() if k in ("lr", "betas", "eps", "weight_decay", "amsgrad", "params")}
    assert keys(opt) == keys(ref)
    with pytest.raises(ValueError):
        optim.Adam([a], weight_decay=0.1)
    with pytest.raises(ValueError):
        optim.Adam([a], amsgrad=True)
    with pytest.raises(ValueError):
        optim.Adam([a], betas=(1.0, 0.999))
    opt.step()                                   # no gradients yet: nothing to do, no GPU needed
    assert len(opt.state) == 0
    a.grad = torch.ones_like(a)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        opt.step()
    opt.zero_grad()
    assert a.grad is None


def test_renderer_pickles():
    """The reference pickles its Renderer into two spawned processes (ESLAM.py:246-260)."""
    import pickle
    from types import SimpleNamespace
    from myslam_amd import scene as scn
    from myslam_amd.src.utils.Renderer import Renderer
    sc = scn.make_scene("room0")
    r = Renderer(sc.cfg(), SimpleNamespace(bound=sc.bound, device="cpu", H=sc.H, W=sc.W, fx=sc.fx, fy=sc.fy, cx=sc.cx,
                                           cy=sc.cy))
    r2 = pickle.loads(pickle.dumps(r))
    assert r2.n_stratified == 32 and r2._bound6 == r._bound6 and torch.equal(r2.bound, r.bound)


def test_scene_shapes_match_reference():
    """Plane shapes / bound / intrinsics the reference computes (room0 27.15 MB, fine y = 111)."""
    from myslam_amd import scene as scn
    sc = scn.make_scene("room0")
    assert [tuple(s) for s in sc.plane_shapes[0]] == [(1, 32, 27, 41), (1, 32, 111, 164)]
    assert [tuple(s) for s in sc.plane_shapes[5]] == [(1, 32, 21, 27), (1, 32, 168, 223)]
    assert sc.plane_bytes == 27147008
    assert np.allclose(sc.bound.numpy(), [[-1.9, 7.94], [-2.2, 4.52], [-2.5, 2.54]], atol=1e-5)
    s2 = scn.make_scene("scene0000")
    assert (s2.H, s2.W) == (460, 620) and max(max(s[2:]) for g in s2.plane_shapes for s in g) == 456
    s3 = scn.make_scene("freiburg1_desk")
    assert (s3.H, s3.W) == (368, 496) and not s3.learnable_beta
    p = scn.new_plane((1, 32, 5, 7))
    assert p.stride() == (32 * 35, 1, 7 * 32, 32)            # one texel = 32 contiguous floats


def test_decoders_state_dict_keys_and_pose_helpers():
    from myslam_amd.src.networks.decoders import Decoders
    from myslam_amd.src import common
    keys = set(Decoders().state_dict().keys())
    assert keys == {"beta", "linears.0.weight", "linears.0.bias", "linears.1.weight", "linears.1.bias",
                    "output_linear.weight", "output_linear.bias", "c_linears.0.weight", "c_linears.0.bias",
                    "c_linears.1.weight", "c_linears.1.bias", "c_output_linear.weight", "c_output_linear.bias"}
    assert Decoders(learnable_beta=False).beta == 10
    with pytest.raises(NotImplementedError):
        Decoders(c_dim=16)
    # quaternion round trip (common.py:155-181 without pytorch3d)
    q = torch.tensor([[0.9, 0.1, -0.3, 0.2], [0.1, 0.7, 0.2, -0.6]])
    q = q / q.norm(dim=1, keepdim=True)
    pose = torch.cat([q, torch.tensor([[1.0, 2.0, 3.0], [-1.0, 0.5, 0.0]])], 1)
    m = common.cam_pose_to_matrix(pose)
    assert torch.allclose(m[:, :3, :3] @ m[:, :3, :3].transpose(1, 2), torch.eye(3).expand(2, 3, 3), atol=1e-6)
    back = common.matrix_to_cam_pose(m)
    sign = torch.sign((back[:, :4] * q).sum(1, keepdim=True))
    assert torch.allclose(back[:, :4] * sign, q, atol=1e-6) and torch.allclose(back[:, 4:], pose[:, 4:])


def test_shard_slices_tile_the_batch():
    from myslam_amd.parallel import shard_slice
    for n in (0, 1, 7, 4096, 8191):
        for w in (1, 2, 3, 8):
            cover = []
            for r in range(w):
                lo, hi = shard_slice(n, r, w)
                assert 0 <= lo <= hi <= n
                cover += list(range(lo, hi))
            assert cover == list(range(n))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _window_problem(dtype=torch.float64):
    """A small mapping iteration on the CPU with the oracle standing in for the kernels: a window of three cameras (the first
    pose fixed, the other two optimised: src/Mapper.py:288-294,312-316), pixels drawn once for the whole window
    (Mapper.py:318-319), the AABB pre-filter as a mask (Mapper.py:322-332), 15 % of the pixels without depth."""
    from oracle import eslam_oracle as orc
    from myslam_amd import scene as scn, synth
    from myslam_amd.src import common
    sc = scn.make_scene("room0")
    planes = scn.synth_planes(sc, dtype=dtype, channels_last=True)
    planes = tuple([p.requires_grad_(True) for p in grp] for grp in planes)
    fx = hp.load("room0_200x40_zero15")
    params = hp.params_from(fx, dtype=dtype, requires_grad=True)
    beta = torch.tensor([10.0], dtype=dtype, requires_grad=True)
    b, n, ns, ni = 3, 40, 24, 8
    c0 = scn.center_pose(sc).to(dtype)
    c2ws = c0[None].repeat(b, 1, 1)
    for k in range(1, b):
        q = torch.tensor([1.0, 0.03 * k, -0.02 * k, 0.05 * k], dtype=dtype)
        c2ws[k, :3, :3] = common.quaternion_to_matrix(q / q.norm()) @ c0[:3, :3]
        c2ws[k, :3, 3] += torch.tensor([0.3 * k, -0.2 * k, 0.1], dtype=dtype)
    poses = common.matrix_to_cam_pose(c2ws[1:]).detach().clone().requires_grad_(True)
    depths = torch.stack([torch.from_numpy(synth.depth_image(sc.H, sc.W, 20 + k, 0.15)).to(dtype) for k in range(b)])
    depths[1] *= 2.6                               # some depths beyond the bound: the pre-filter mask drops those rays
    colors = torch.stack([torch.from_numpy(synth.color_image(sc.H, sc.W, 30 + k)).to(dtype) for k in range(b)])
    idx = torch.from_numpy(synth.hash_randint(sc.H * sc.W, (b * n,), 777))
    R, S = b * n, ns + ni
    t_rand = torch.from_numpy(synth.hash_uniform((R, S), 91)).to(dtype)
    t_uni = torch.from_numpy(synth.hash_uniform((R, ns), 92)).to(dtype)
    u = torch.from_numpy(synth.hash_uniform((R, ni), 93)).to(dtype)

    def rays():
        c = torch.cat([c2ws[0:1], common.cam_pose_to_matrix(poses)], 0)
        ro, rd, gd, gc = orc.rays_from_pixels(idx, 0, sc.H, 0, sc.W, sc.fx, sc.fy, sc.cx, sc.cy, c, depths, colors)
        with torch.no_grad():
            keep = orc.aabb_exit(ro, rd, sc.bound.to(dtype)) >= gd
        return ro, rd, gd, gc, keep

    def render(sl, ro, rd, gd):
        return orc.render_batch_ray(planes, params, beta, sc.bound, rd[sl], ro[sl], sc.truncation, gd[sl], ns, ni, t_rand[sl],
                                    t_uni[sl], u[sl])
    plist = hp.flat_planes(planes) + [params[k] for k in orc.DECODER_KEYS] + [beta, poses]
    return dict(sc=sc, planes=planes, plist=plist, rays=rays, render=render, t_rand=t_rand, ns=ns, ni=ni, R=R, orc=orc)


def _local_sums(depth, color, sdf, z, gd, gc, keep, tr):
    """This rank's five squared-error sums of the mapping loss (what the forward kernel's epilogue accumulates)."""
    m = (gd > 0) & keep
    d = gd[m][:, None]
    zz, ss = z[m], sdf[m]
    front = zz < d - tr
    back = zz > d + tr
    center = (zz > d - 0.4 * tr) & (zz < d + 0.4 * tr)
    tail = ~front & ~back & ~center
    pred = zz + ss * tr
    return torch.stack([((ss - 1) ** 2)[front].sum(), ((pred - d) ** 2)[center].sum(), ((pred - d) ** 2)[tail].sum(),
                        ((gd[m] - depth[m]) ** 2).sum(), ((gc[keep] - color[keep]) ** 2).sum()])


def _dp_worker(rank, world, port, ret, compact):
    """One gloo rank of the collective-free scheme parallel.ShardedMapper runs on the GPU: every rank draws the WHOLE batch,
    forms the loss's global set sizes from the whole batch's depth-guided z_vals (no exchange) and the union of texels the
    batch can touch from ray geometry (no exchange), renders ITS SLICE, and ONE all-reduce sums [tail | marked texels] of
    the flat gradient buffer - pose gradients of the window included."""
    import torch.distributed as dist
    from myslam_amd import parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    pr = _window_problem()
    sc, orc = pr["sc"], pr["orc"]
    tr = sc.truncation
    ro, rd, gd, gc, keep = pr["rays"]()
    lo, hi = parallel.shard_slice(pr["R"], rank, world)
    sl = slice(lo, hi)
    # global set sizes, redundantly: the depth-guided sampler of the whole batch (rows of rays without depth are not looked at)
    z_all = torch.zeros(pr["R"], pr["ns"] + pr["ni"], dtype=torch.float64)
    has = gd > 0
    z_all[has] = orc.depth_guided_z(gd[has], pr["ns"], pr["ni"], tr, pr["t_rand"][has])
    acc = parallel.set_sizes_from_z(z_all, gd.detach(), tr, keep)
    cnts = acc[list(parallel._ACC_COUNT_SLOTS)]
    depth, color, sdf, z = pr["render"](sl, ro, rd, gd)
    assert torch.equal(z[has[sl]], z_all[sl][has[sl]])          # the replayed sampler IS the render's sampler
    sums = _local_sums(depth, color, sdf, z, gd[sl], gc[sl], keep[sl], tr)
    w = orc.MAPPING_W
    wv = torch.tensor([w["w_fs"], w["w_center"], w["w_tail"], w["w_depth"], w["w_color"]], dtype=torch.float64)
    local = (wv * sums / cnts).sum()              # local sums over GLOBAL denominators: this rank's share of the loss
    plist = pr["plist"]
    fg = parallel.FlatGrads(plist, extra=16)
    for v, g in zip(fg.views, torch.autograd.grad(local, plist)):
        v.copy_(g)
    fg.extra[list(parallel._ACC_SUM_SLOTS)] = sums.detach()
    fg.extra[list(parallel._ACC_COUNT_SLOTS)] = torch.zeros(5, dtype=torch.float64)      # (the global counts are known everywhere)
    n_plane = sum(p.numel() for p in plist[:12])
    if compact:
        shapes = [(p.shape[2], p.shape[3]) for p in plist[:12]]
        base = [fg.offsets[i] // 32 for i in range(12)]
        touched = parallel.mark_rays(shapes, [float(v) for v in sc.bound.reshape(-1)], ro.detach(), rd.detach(), gd.detach(), tr,
                                     base, n_plane // 32)
        nz = (fg.flat[:n_plane].view(-1, 32) != 0).any(1)
        assert bool((touched.bool() | ~nz).all()), "a texel outside the conservative marking received gradient"
        sent, dense = fg.exchange_union(touched, n_plane)
        if rank == 0:
            ret["exchange"] = (sent, dense, int(touched.sum()))
            ret["touched"] = touched.numpy().copy()
    else:
        fg.all_reduce()
    fg.assign()
    if rank == 0:
        ret["loss"] = float((wv * fg.extra[list(parallel._ACC_SUM_SLOTS)] / cnts).sum())
        ret["flat"] = fg.flat[:fg.offsets[-1]].clone().numpy()
        ret["strides_ok"] = all(p.grad.stride() == p.stride() for p in plist)
        ret["pose_grad"] = plist[-1].grad.clone().numpy()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,compact", [(2, False), (2, True), (3, True)])
def test_ray_sharded_mapping_iteration_equals_single_process(world, compact):
    """2 / 3 gloo ranks, dense and marked-texel exchange: the summed flat buffer - plane, decoder, beta AND pose gradients of a
    keyframe window with joint_opt - equals the unsharded iteration's gradients to float64 rounding; no collective besides
    the one gradient all-reduce."""
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_dp_worker, args=(world, _free_port(), ret, compact), nprocs=world, join=True)
    pr = _window_problem()
    sc, orc = pr["sc"], pr["orc"]
    ro, rd, gd, gc, keep = pr["rays"]()
    depth, color, sdf, z = pr["render"](slice(0, pr["R"]), ro, rd, gd)
    loss = orc.mapping_loss(depth[keep], color[keep], sdf[keep], z[keep], gd[keep], gc[keep], sc.truncation)   # Mapper.py:329-346
    assert 0.5 < float(keep.float().mean()) < 0.98 and 0 < int((gd[keep] == 0).sum())
    grads = torch.autograd.grad(loss, pr["plist"])
    assert abs(ret["loss"] - float(loss)) <= 1e-10 * abs(float(loss))
    flat_ref = np.concatenate([g.contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1).reshape(-1).numpy()
                               if g.dim() == 4 else g.reshape(-1).numpy() for g in grads])
    assert ret["strides_ok"]
    assert np.abs(ret["flat"] - flat_ref).max() <= 1e-9 * np.abs(flat_ref).max()
    gp = grads[-1].numpy()
    assert np.abs(gp).max() > 0 and np.abs(ret["pose_grad"] - gp).max() <= 1e-9 * np.abs(gp).max()
    if compact:
        sent, dense, marked = ret["exchange"]
        assert sent < 0.5 * dense, (sent, dense)          # 120 rays can touch a small part of the 27 MB of planes
        # the marking is conservative (every texel with gradient is marked) but not loose
        n_plane = sum(p.numel() for p in pr["plist"][:12])
        nz = (np.abs(flat_ref[:n_plane].reshape(-1, 32)) > 0).any(1)
        assert not (nz & (ret["touched"] == 0)).any()
        assert marked <= 4 * int(nz.sum()), (marked, int(nz.sum()))


def test_set_sizes_from_z_counts_the_loss_regions():
    """parallel.set_sizes_from_z against the oracle's own masks (src/Mapper.py:124-134), with a ray mask."""
    from oracle import eslam_oracle as orc
    from myslam_amd import parallel
    g = torch.Generator().manual_seed(5)
    gd = torch.rand(50, generator=g) * 2 + 0.3
    gd[::7] = 0
    keep = torch.rand(50, generator=g) > 0.2
    tr = 0.06
    z = torch.zeros(50, 40)
    has = gd > 0
    z[has] = orc.depth_guided_z(gd[has], 32, 8, tr, torch.rand(int(has.sum()), 40, generator=g))
    acc = parallel.set_sizes_from_z(z, gd, tr, keep)
    m = has & keep
    d = gd[m][:, None]
    front = z[m] < d - tr
    back = z[m] > d + tr
    center = (z[m] > d - 0.4 * tr) & (z[m] < d + 0.4 * tr)
    want = [front.sum(), center.sum(), (~front & ~back & ~center).sum(), m.sum(), 3 * keep.sum()]
    assert [int(acc[k]) for k in parallel._ACC_COUNT_SLOTS] == [int(v) for v in want]
    assert float(acc[list(parallel._ACC_SUM_SLOTS)].abs().sum()) == 0.0
