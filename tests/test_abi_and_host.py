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
    assert lib.eslam_abi_version() == _hip.ABI_VERSION == 4
    assert lib.eslam_bwd_workspace_bytes(262144) > 262144 * 128 * 4
    assert lib.eslam_bwd_workspace_bytes(-1) == -1


def test_struct_layout_matches_header():
    from myslam_amd import _hip
    assert ctypes.sizeof(_hip.PlaneDesc) == 8 + 8 + 4 + 4 + 3 * 8 + 8
    assert ctypes.sizeof(_hip.DecodersDesc) == 13 * 8
    assert ctypes.sizeof(_hip.AdamTensor) == 4 * 8 + 8 + 8
    assert _hip.N_DEC_PARAMS == 2 * (16 * 64 + 16 + 16 * 16 + 16) + 17 + 51


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


def _dp_worker(rank, world, port, ret, compact):
    """One gloo rank: oracle forward on its ray shard, two-phase loss with all-reduced denominators, gradients
    all-reduced through parallel.FlatGrads - the same sequence parallel.ShardedMapper runs on the GPU."""
    import torch.distributed as dist
    from oracle import eslam_oracle as orc
    from myslam_amd.parallel import FlatGrads, shard_slice
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    fx = hp.load("room0_200x40_zero15")
    sc, planes = hp.scene_and_planes(fx, dtype=torch.float64, channels_last=True, requires_grad=True)
    params = hp.params_from(fx, dtype=torch.float64, requires_grad=True)
    beta = torch.tensor([10.0], dtype=torch.float64, requires_grad=True)
    t_rand, t_uni, u = (t.double() for t in hp.rand_inputs(fx))
    lo, hi = shard_slice(int(fx["R_eff"]), rank, world)
    sl = slice(lo, hi)
    ro = torch.from_numpy(fx["rays_o"]).double()[sl]
    rd = torch.from_numpy(fx["rays_d"]).double()[sl]
    gd = torch.from_numpy(fx["gt_depth"]).double()[sl]
    gc = torch.from_numpy(fx["gt_color"]).double()[sl]
    tr = float(fx["truncation"])
    depth, color, sdf, z = orc.render_batch_ray(planes, params, beta, sc.bound, rd, ro, tr, gd, 32, 8, t_rand[sl],
                                                t_uni[sl], u[sl])
    # phase 1: local set sizes and squared-error sums (what eslam_loss_reduce accumulates) -> all-reduce
    m = gd > 0
    d = gd[m][:, None]
    zz, ss = z[m], sdf[m]
    front = zz < d - tr
    back = zz > d + tr
    center = (zz > d - 0.4 * tr) & (zz < d + 0.4 * tr)
    tail = ~front & ~back & ~center
    pred = zz + ss * tr
    sums = torch.stack([((ss - 1) ** 2)[front].sum(), ((pred - d) ** 2)[center].sum(), ((pred - d) ** 2)[tail].sum(),
                        ((gd[m] - depth[m]) ** 2).sum(), ((gc - color) ** 2).sum()])
    cnts = torch.tensor([front.sum(), center.sum(), tail.sum(), m.sum(), gc.numel()], dtype=torch.float64)
    dist.all_reduce(cnts)
    # phase 2: local loss scaled by GLOBAL denominators; its gradient is this rank's share of the global gradient
    w = orc.MAPPING_W
    wv = torch.tensor([w["w_fs"], w["w_center"], w["w_tail"], w["w_depth"], w["w_color"]], dtype=torch.float64)
    local = (wv * sums / cnts).sum()
    plist = hp.flat_planes(planes) + [params[k] for k in orc.DECODER_KEYS] + [beta]
    fg = FlatGrads(plist)
    grads = torch.autograd.grad(local, plist)
    for v, g in zip(fg.views, grads):
        v.copy_(g)
    if compact:      # block-sparse exchange: only the texel rows some rank touched, plus the dense decoder tail
        sent, dense = fg.all_reduce_compact(sum(p.numel() for p in plist[:12]))
        if rank == 0:
            ret["exchange"] = (sent, dense)
    else:
        fg.all_reduce()
    fg.assign()
    total = local.detach().clone()
    dist.all_reduce(total)
    if rank == 0:
        ret["loss"] = float(total)
        ret["flat"] = fg.flat.clone().numpy()
        ret["strides_ok"] = all(p.grad.stride() == p.stride() for p in plist)
    dist.destroy_process_group()


@pytest.mark.parametrize("world,compact", [(2, False), (2, True), (3, True)])
def test_ray_sharded_data_parallel_equals_single_process(world, compact):
    import torch.multiprocessing as mp
    from tests.test_oracle_golden import run_oracle
    mgr = mp.Manager()
    ret = mgr.dict()
    port = _free_port()
    mp.spawn(_dp_worker, args=(world, port, ret, compact), nprocs=world, join=True)
    fx = hp.load("room0_200x40_zero15")
    ref = run_oracle(fx, torch.float64)
    assert abs(ret["loss"] - float(ref["loss"])) <= 1e-10 * abs(float(ref["loss"]))
    from oracle import eslam_oracle as orc
    ref_list = [p.grad for p in hp.flat_planes(ref["planes"])] + [ref["params"][k].grad for k in orc.DECODER_KEYS] + \
               [ref["beta"].grad]
    # flat buffer holds the gradients in the parameters' own (channels-last) memory order
    flat_ref = np.concatenate([g.contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1).reshape(-1).numpy()
                               if g.dim() == 4 else g.reshape(-1).numpy() for g in ref_list])
    assert ret["strides_ok"]
    assert np.abs(ret["flat"] - flat_ref).max() <= 1e-9 * np.abs(flat_ref).max()
    if compact:
        sent, dense = ret["exchange"]
        assert sent < 0.5 * dense, (sent, dense)          # 200 rays touch a small part of the 27 MB of planes


def _sync_worker(rank, world, port, ret):
    import torch.distributed as dist
    from myslam_amd import parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 1000
    g = torch.Generator().manual_seed(100 + rank)
    acc = torch.floor(torch.rand(16, generator=g) * 1000)
    touched = (torch.rand(n, generator=g) < 0.1).to(torch.uint8)
    buf = parallel.sync_pack(acc, touched, torch.zeros(parallel.sync_words(n), dtype=torch.int32))
    dist.all_reduce(buf)
    gacc, union = torch.zeros(16), torch.zeros(n, dtype=torch.uint8)
    parallel.sync_unpack(buf, acc, gacc, union)
    ret[rank] = (acc.numpy(), touched.numpy(), gacc.numpy(), union.numpy())
    dist.destroy_process_group()


def test_sync_collective_gives_global_set_sizes_and_the_union():
    """The ONE int32 all-reduce between forward and backward of the ray-sharded step (parallel.sync_pack / sync_unpack):
    3 gloo ranks end with the summed set sizes and the union of their touched texels; the loss's sums stay local."""
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_sync_worker, args=(3, _free_port(), ret), nprocs=3, join=True)
    accs = np.stack([ret[r][0] for r in range(3)])
    union = np.maximum.reduce([ret[r][1] for r in range(3)])
    for r in range(3):
        gacc, u = ret[r][2], ret[r][3]
        assert np.array_equal(u, union)
        for k in (0, 1, 2, 6, 9):
            assert gacc[k] == accs[:, k].sum()
        for k in (3, 4, 5, 7, 8):
            assert gacc[k] == accs[r, k]
