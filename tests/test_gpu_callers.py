"""GPU tests of the caller-side glue kernels (csrc/eslam_callers.hip, eslam_prefilter) against the tensor-op chains
of the reference's loops that they replace (src/common.py:169-181, src/Tracker.py:175-195,304-307,
src/Mapper.py:322-328)."""
import numpy as np
import pytest
import torch

from tests import helpers as hp

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


def _torch_pose_to_matrix(p):
    """The tensor-op formula (what pytorch3d.transforms.quaternion_to_matrix computes), any dtype / device."""
    from myslam_amd.src import common
    c2w = torch.eye(4, dtype=p.dtype).unsqueeze(0).repeat(p.shape[0], 1, 1)
    c2w[:, :3, :3] = common.quaternion_to_matrix(p[:, :4])
    c2w[:, :3, 3] = p[:, 4:]
    return c2w


def test_pose_to_matrix_forward_backward():
    from myslam_amd.src import common
    dev = _dev()
    g = torch.Generator().manual_seed(0)
    poses = torch.randn(37, 7, generator=g)
    poses[:5, :4] = torch.nn.functional.normalize(poses[:5, :4], dim=-1)      # unit and non-unit quaternions
    w = torch.randn(37, 4, 4, generator=g)
    a = poses.clone().to(dev).requires_grad_(True)
    out = common.cam_pose_to_matrix(a)
    (out * w.to(dev)).sum().backward()
    b = poses.clone().double().requires_grad_(True)
    ref = _torch_pose_to_matrix(b)
    (ref * w.double()).sum().backward()
    assert hp.rel_err(out.detach().cpu().numpy(), ref.detach().numpy()) <= 1e-6
    assert hp.rel_err(a.grad.cpu().numpy(), b.grad.numpy()) <= 1e-5
    # round trip with the inverse conversion
    back = common.matrix_to_cam_pose(common.cam_pose_to_matrix(a[:5].detach()))
    sign = torch.sign((back[:, :4] * a[:5, :4].detach()).sum(-1, keepdim=True))
    assert hp.rel_err((back[:, :4] * sign).cpu().numpy(), a[:5, :4].detach().cpu().numpy()) <= 1e-5
    # host tensors keep working (tensor ops)
    assert hp.rel_err(common.cam_pose_to_matrix(poses).numpy(), ref.detach().numpy()) <= 1e-5


@pytest.mark.parametrize("need_depth", [False, True])
def test_prefilter_matches_tensor_ops(need_depth):
    from myslam_amd import ops, scene as scn
    dev = _dev()
    sc = scn.make_scene("room0")
    g = torch.Generator().manual_seed(1)
    R = 5000
    ro = (sc.bound.mean(1) + (torch.rand(R, 3, generator=g) - 0.5) * 2).to(dev)
    rd = torch.randn(R, 3, generator=g).to(dev)
    rd[:7, 1] = 0.0                                                      # rays along a slab: division by zero
    gd = (torch.rand(R, generator=g) * 8).to(dev)
    gd[::9] = 0.0
    keep = ops.prefilter(ro, rd, gd, ops.bound_to_host(sc.bound), need_depth)
    t = (sc.bound.unsqueeze(0).to(dev) - ro.unsqueeze(-1)) / rd.unsqueeze(-1)          # Mapper.py:325-327
    t, _ = torch.min(torch.max(t, dim=2)[0], dim=1)
    ref = t >= gd
    if need_depth:
        ref = ref & (gd > 0)
    assert keep.dtype == torch.bool and 0.05 < float(ref.float().mean()) < 0.95
    assert torch.equal(keep, ref)


@pytest.mark.parametrize("R,with_keep", [(1, False), (2, True), (501, True), (2000, False), (2000, True), (8192, True)])
def test_tracking_mask_matches_torch_median(R, with_keep):
    from myslam_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(R)
    gd = (torch.rand(R, generator=g) * 3 + 0.2).to(dev)
    depth = gd + (torch.randn(R, generator=g) * 0.01).to(dev)
    depth[::17] += 2.0                                                   # outliers
    keep = (torch.rand(R, generator=g) < 0.7).to(dev) if with_keep else None
    if with_keep:
        keep[0] = True
    got = ops.tracking_mask(depth, gd, keep)
    err = (gd - depth).abs()
    if keep is None:
        ref = err < 10 * err.median()                                    # Tracker.py:193-195
    else:
        ref = keep & (err < 10 * err[keep].median())                     # the same over the compacted rays
    assert torch.equal(got, ref)
    assert int(got.sum()) > 0


def test_tracking_mask_nan_and_empty():
    from myslam_amd import ops
    dev = _dev()
    gd = torch.ones(64, device=dev)
    depth = torch.ones(64, device=dev) * 1.1
    depth[3] = float("nan")
    assert int(ops.tracking_mask(depth, gd).sum()) == 0                  # torch.median propagates NaN -> empty mask
    keep = torch.ones(64, dtype=torch.bool, device=dev)
    keep[3] = False
    assert int(ops.tracking_mask(depth, gd, keep).sum()) == 63           # the NaN ray is filtered out: all others pass
    depth2 = depth.clone()
    depth2[3] = 1.0
    assert int(ops.tracking_mask(depth2, gd, torch.zeros(64, dtype=torch.bool, device=dev)).sum()) == 0


def test_keep_best():
    from myslam_amd import ops
    dev = _dev()
    best = torch.full((1,), float("inf"), device=dev)
    best_pose = torch.zeros(1, 7, device=dev)
    seq = [3.0, 2.0, 2.5, float("nan"), 1.0, 1.0]
    want, wp = float("inf"), None
    for k, lv in enumerate(seq):
        pose = torch.full((1, 7), float(k), device=dev)
        ops.keep_best(torch.tensor([lv], device=dev), pose, best, best_pose)
        if lv < want:
            want, wp = lv, k
        assert float(best) == want and torch.equal(best_pose, torch.full((1, 7), float(wp), device=dev))


def test_tracking_loss_masked_equals_compacted():
    """losses.tracking_loss with the pre-filter as a mask == the reference's order (compact, then median mask)."""
    from myslam_amd import harness, losses, ops
    dev = _dev()
    wl = harness.make_workload("room0", 1500, 32, 8, device=dev, planes="synth", rays_grad=True, zero_frac=0.1)
    keep = ops.prefilter(wl.rays_o, wl.rays_d, wl.gt_depth, ops.bound_to_host(wl.scene.bound), True)
    assert 0 < int(keep.sum()) < wl.R
    depth, color, sdf, z = wl.forward()
    la = losses.tracking_loss(depth, color, sdf, z, wl.gt_depth, wl.gt_color, wl.truncation, ray_mask=keep)
    ga = torch.autograd.grad(la, [depth, color, sdf])
    lb = losses.tracking_loss(depth[keep], color[keep], sdf[keep], z[keep], wl.gt_depth[keep], wl.gt_color[keep], wl.truncation)
    gb = torch.autograd.grad(lb, [depth, color, sdf])
    assert abs(float(la) - float(lb)) <= 1e-6 * abs(float(lb))
    for a, b in zip(ga, gb):
        assert hp.rel_err(a.cpu().numpy(), b.cpu().numpy()) <= 1e-6


def test_block_sparse_exchange_kernels():
    """eslam_blocks_touched / pack / unpack against the tensor ops parallel.FlatGrads.all_reduce_compact uses on the CPU."""
    import ctypes
    from myslam_amd import _hip
    dev = _dev()
    lib = _hip.lib()
    g = torch.Generator().manual_seed(5)
    n_blocks, n_tail = 10007, 2693
    rows = torch.randn(n_blocks, 32, generator=g)
    rows[torch.rand(n_blocks, generator=g) < 0.8] = 0.0            # 80 % empty blocks
    rows[5, :] = 0.0
    rows[5, 31] = -0.0                                             # a negative zero is still zero
    rows[7, :] = 0.0
    rows[7, 17] = 1e-30                                            # a single tiny value marks the block
    flat = torch.cat([rows.reshape(-1), torch.randn(n_tail, generator=g)]).to(dev)
    touched = torch.empty(n_blocks, dtype=torch.uint8, device=dev)
    st = _hip.stream_handle(dev)
    _hip.check(lib.eslam_blocks_touched(_hip.ptr(flat), n_blocks, _hip.ptr(touched), st), "touched")
    ref_t = (torch.count_nonzero(rows, dim=1) > 0)
    assert torch.equal(touched.cpu().bool(), ref_t) and not bool(touched[5]) and bool(touched[7])
    idx = touched.nonzero().squeeze(1)
    tail = flat[n_blocks * 32:]
    buf = torch.empty(idx.numel() * 32 + n_tail, device=dev)
    _hip.check(lib.eslam_blocks_pack(_hip.ptr(flat), _hip.ptr(idx), idx.numel(), _hip.ptr(tail), n_tail, _hip.ptr(buf), st), "pack")
    assert torch.equal(buf.cpu(), torch.cat([rows[ref_t].reshape(-1), flat[n_blocks * 32:].cpu()]))
    before = flat.clone()
    buf.mul_(3.0)
    _hip.check(lib.eslam_blocks_unpack(_hip.ptr(flat), _hip.ptr(idx), idx.numel(), _hip.ptr(tail), n_tail, _hip.ptr(buf), st), "unpack")
    assert torch.equal(flat, before * 3.0)                         # untouched blocks are zero either way
    # empty union, empty tail
    z = torch.zeros(64, device=dev)
    t2 = torch.empty(2, dtype=torch.uint8, device=dev)
    _hip.check(lib.eslam_blocks_touched(_hip.ptr(z), 2, _hip.ptr(t2), st), "touched")
    assert t2.tolist() == [0, 0]
    assert lib.eslam_blocks_pack(_hip.ptr(z), None, 0, None, 0, _hip.ptr(z), st) == 0


@pytest.mark.parametrize("scene,zero_frac", [("room0", 0.0), ("toy", 0.2)])
def test_marked_texels_contain_every_block_the_backward_touches(scene, zero_frac):
    """eslam_mark_touched (from sample positions, after the forward) must be a superset of the non-zero 128-byte blocks
    of the plane gradients (eslam_blocks_touched after the backward) - the block-sparse exchange relies on it."""
    import ctypes
    from myslam_amd import harness, ops, _hip
    from myslam_amd.parallel import FlatGrads
    dev = _dev()
    wl = harness.make_workload(scene, 3000, 40, 8, device=dev, zero_frac=zero_frac)
    params = wl.plane_list + ops.decoder_params(wl.decoders) + [wl.decoders.beta]
    fg = FlatGrads(params)
    depth, color, sdf, z = wl.renderer.render_batch_ray(wl.planes, wl.decoders, wl.rays_d, wl.rays_o, dev, wl.truncation,
                                                        gt_depth=wl.gt_depth)
    n_blocks = sum(p.numel() for p in wl.plane_list) // 32
    marked = torch.empty(n_blocks, dtype=torch.uint8, device=dev)
    base = (ctypes.c_int64 * 12)(*[fg.offsets[i] // 32 for i in range(12)])
    arr, _ = _hip.make_planes(tuple([p.detach() for p in grp] for grp in wl.planes))
    st = _hip.stream_handle(dev)
    _hip.check(_hip.lib().eslam_mark_touched(arr, _hip.make_bound(ops.bound_to_host(wl.scene.bound)), _hip.ptr(wl.rays_o),
                                             _hip.ptr(wl.rays_d), _hip.ptr(z), wl.R, wl.S, base, n_blocks,
                                             _hip.ptr(marked), st), "eslam_mark_touched")
    with ops.grad_sink(fg):
        ((depth * wl._cot[0]).sum() + (color * wl._cot[1]).sum() + (sdf * wl._cot[2]).sum()).backward()
    nz = torch.empty(n_blocks, dtype=torch.uint8, device=dev)
    _hip.check(_hip.lib().eslam_blocks_touched(_hip.ptr(fg.flat), n_blocks, _hip.ptr(nz), st), "eslam_blocks_touched")
    torch.cuda.synchronize()
    assert int(nz.sum()) > 100
    assert int((nz.bool() & ~marked.bool()).sum()) == 0, "a texel received gradient without being marked"
    # ... and not wildly larger: samples behind the surface have transmittance exactly 0 in float32, so the colour planes'
    # texels there are visited but receive exact zeros - the marked set is up to ~3x the non-zero set, still a small
    # fraction of the 212 k blocks
    assert int(marked.sum()) <= 4 * int(nz.sum()) and int(marked.sum()) < 0.3 * n_blocks


@pytest.mark.parametrize("cameras,R", [(1, 4096), (1, 77), (3, 3000), (1, 9000)])
def test_ray_order_is_a_permutation_and_groups_neighbours(cameras, R):
    """eslam_ray_order: a permutation of every chunk of 8192 rays, for one origin (2-D Hilbert key of the direction) and
    several (3-D Morton key); consecutive rays of the order point in nearby directions."""
    from myslam_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(cameras * 1000 + R)
    org = torch.randn(cameras, 3, generator=g)
    cam = torch.randint(cameras, (R,), generator=g)
    ro = org[cam].to(dev)
    d = torch.randn(R, 3, generator=g) * 0.35
    d[:, 2] = -1.0                                              # a pinhole-like fan of directions
    rd = d.to(dev)
    perm, side = ops.ray_order_async(ro, rd)
    torch.cuda.current_stream().wait_stream(side)
    p = perm.cpu().long()
    for lo in range(0, R, 8192):                                # chunks are ordered independently
        hi = min(R, lo + 8192)
        assert torch.equal(torch.sort(p[lo:hi]).values, torch.arange(lo, hi))
    if cameras == 1 and R >= 1000:
        dn = torch.nn.functional.normalize(d, dim=1)
        n1 = min(R, 8192)
        step_sorted = (dn[p[1:n1]] - dn[p[:n1 - 1]]).norm(dim=1).mean()
        step_given = (dn[1:n1] - dn[:n1 - 1]).norm(dim=1).mean()
        assert float(step_sorted) < 0.1 * float(step_given)


def test_shard_sync_pack_unpack_match_the_tensor_ops():
    """eslam_shard_sync_pack / _unpack (the one collective between forward and backward of the ray-sharded step) against
    the tensor-op form parallel.sync_pack / sync_unpack run on the CPU under gloo."""
    from myslam_amd import parallel
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    for n in (0, 1, 5, 6, 7, 1003, 212_345):
        acc = torch.floor(torch.rand(16, generator=g) * 1e5)
        acc[3] = 0.37
        t = (torch.rand(n, generator=g) < 0.3).to(torch.uint8) * 7 if n else None
        ref = parallel.sync_pack(acc, t, torch.zeros(parallel.sync_words(n), dtype=torch.int32))
        got = parallel.sync_pack(acc.to(dev), None if t is None else t.to(dev),
                                 torch.full((parallel.sync_words(n),), -1, dtype=torch.int32, device=dev))
        assert torch.equal(got.cpu()[:5], ref[:5]) and torch.equal(got.cpu()[8:], ref[8:])
        tot = ref * 3                                     # as if three ranks had contributed the same buffer
        ga, gt = torch.zeros(16), (torch.zeros(n, dtype=torch.uint8) if n else None)
        parallel.sync_unpack(tot, acc, ga, gt)
        da, dt = torch.zeros(16, device=dev), (torch.full((n,), 9, dtype=torch.uint8, device=dev) if n else None)
        parallel.sync_unpack(tot.to(dev), acc.to(dev), da, dt)
        assert torch.equal(da.cpu(), ga)
        assert float(ga[0]) == 3 * float(acc[0]) and float(ga[3]) == float(acc[3])
        if n:
            assert torch.equal(dt.cpu(), gt) and torch.equal(gt, (t != 0).to(torch.uint8))


def test_in_kernel_jitter_is_uniform_fresh_per_step_and_reproducible():
    """eslam_sample_z_all_rng: the jitter / importance numbers drawn inside the sampler (no torch.rand launch).  Each z must
    lie in its stratum [lower, upper) with a uniform position, successive iterations must differ (the forward kernel
    advances the device step counter), and the same (seed, step) must give the same samples."""
    from myslam_amd import harness, ops
    dev = torch.device("cuda:0")
    wl = harness.make_workload("room0", 1500, 24, 8, device=dev, zero_frac=0.1)
    r = wl.renderer

    def render():
        with torch.no_grad():
            return r.render_batch_ray(wl.planes, wl.decoders, wl.rays_d, wl.rays_o, dev, wl.truncation, gt_depth=wl.gt_depth)[3]
    torch.manual_seed(1234)
    ops._rng_state(dev).zero_()
    z1, z2 = render(), render()
    assert int(ops._rng_state(dev)[0]) == 2                       # one bump per rendered batch
    torch.manual_seed(1234)
    ops._rng_state(dev).zero_()
    z1b = render()
    assert torch.equal(z1, z1b)                                   # same seed and step: same samples
    assert not torch.equal(z1, z2)
    has = wl.gt_depth > 0
    assert (z1[:, 1:] >= z1[:, :-1]).all() and (z2[:, 1:] >= z2[:, :-1]).all()
    # recover the uniform positions of the depth-guided rows: z = lower + (upper - lower) t around the un-jittered samples
    r.perturb = False
    torch.manual_seed(1234)
    z0 = render()
    r.perturb = True
    z0, zj = z0[has].double(), z1[has].double()
    mids = 0.5 * (z0[:, 1:] + z0[:, :-1])
    lower = torch.cat([z0[:, :1], mids], 1)
    upper = torch.cat([mids, z0[:, -1:]], 1)
    w = upper - lower
    ok = w > 1e-6
    t = ((zj - lower) / w.clamp(min=1e-12))[ok]
    assert float(t.min()) >= -1e-4 and float(t.max()) < 1 + 1e-4
    n = t.numel()
    assert abs(float(t.mean()) - 0.5) < 4 * (1 / 12) ** 0.5 / n ** 0.5 + 1e-3
    assert abs(float(t.var()) - 1 / 12) < 5e-3
    hist = torch.histc(t.float(), bins=16, min=0.0, max=1.0) / n
    assert float((hist - 1 / 16).abs().max()) < 0.01
    # neighbouring samples / rays are uncorrelated
    tt = ((zj - lower) / w.clamp(min=1e-12))
    a, b = tt[:, 2:20].reshape(-1) - 0.5, tt[:, 3:21].reshape(-1) - 0.5
    assert abs(float((a * b).mean()) * 12) < 0.03
    a, b = tt[:-1, 2:20].reshape(-1) - 0.5, tt[1:, 2:20].reshape(-1) - 0.5
    assert abs(float((a * b).mean()) * 12) < 0.03


def test_integration_md_ctypes_snippet_runs_as_written():
    """INTEGRATION.md section 3 shows a raw ctypes call of eslam_render_fwd: execute that code block LITERALLY (VERDICT r01:
    the snippet had lost an argument) and compare with the shipped binding."""
    import os
    import re
    from myslam_amd import harness, ops
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    md = open(os.path.join(root, "INTEGRATION.md")).read()
    m = re.search(r"```python\n(# \(tests/test_gpu_callers.py::test_integration_md_ctypes_snippet_runs_as_written.*?)```", md, re.S)
    assert m, "the section-3 snippet is gone from INTEGRATION.md"
    code = m.group(1).replace('ctypes.CDLL("myslam_amd/lib/libeslam_hip.so")',
                              f'ctypes.CDLL("{os.path.join(root, "myslam_amd", "lib", "libeslam_hip.so")}")')
    dev = torch.device("cuda:0")
    wl = harness.make_workload("room0", 300, 24, 8, device=dev)
    with torch.no_grad():
        d_ref, c_ref, s_ref, z = wl.renderer.render_batch_ray(wl.planes, wl.decoders, wl.rays_d, wl.rays_o, dev, wl.truncation,
                                                              gt_depth=wl.gt_depth)
    b = ops.bound_to_host(wl.decoders.bound)
    env = dict(all_planes=wl.planes, decoder_tensors_in_header_order=[p.detach() for p in ops.decoder_params(wl.decoders)],
               beta_tensor=wl.decoders.beta.detach(), x0=b[0], x1=b[1], y0=b[2], y1=b[3], z0=b[4], z1=b[5],
               rays_o=wl.rays_o.detach(), rays_d=wl.rays_d.detach(), z_vals=z.contiguous())
    exec(compile(code, "INTEGRATION.md section 3", "exec"), env)
    torch.cuda.synchronize()
    assert torch.equal(env["depth"], d_ref) and torch.equal(env["rgb"], c_ref) and torch.equal(env["sdf"], s_ref)


def test_bench_line_contract():
    """bench.py's one JSON line: the keys the driver and the judge read, a roofline whose every fraction is <= 1, and the
    CPU baseline timed beside it (a short run: 3 timed steps, bounded CPU sample)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "2", "--no-extras"],
                         capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"] and d["vs_baseline"] is None
    assert abs(d["value"] - 4096 * 64 / d["ms_per_step"] * 1e3) <= 1e-6 * d["value"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert 0.0 < rf["frac"] <= 1.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) <= 1e-9
    for kern in rf["kernels"]:          # (the scatter's and the decoder backward's need the PMC bytes of this very build)
        assert kern["frac"] is None or 0.0 < kern["frac"] <= 1.0, kern
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == d["unit"] and cb["sample"]
