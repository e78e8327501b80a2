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


def test_texel_list_and_block_move_kernels():
    """eslam_blocks_compact (ascending list + length of the marked texels, on the device) and eslam_blocks_pack_dev /
    _unpack_dev / _zero_dev against the tensor ops parallel.FlatGrads.exchange_union runs under gloo."""
    from myslam_amd import _hip
    dev = _dev()
    lib = _hip.lib()
    g = torch.Generator().manual_seed(5)
    st = _hip.stream_handle(dev)
    for n_blocks, dens in ((10007, 0.2), (4096, 1.0), (4097, 0.0), (16, 0.5), (212_345, 0.13)):
        n_tail, tail_pad = 2693 + 147, 2848
        touched = (torch.rand(n_blocks, generator=g) < dens).to(torch.uint8) * 3          # any non-zero byte marks
        rows = torch.randn(n_blocks, 32, generator=g)
        flat = torch.cat([rows.reshape(-1), torch.randn(n_tail, generator=g)]).to(dev)
        td = touched.to(dev)
        scratch = torch.zeros(int(lib.eslam_blocks_compact_scratch_words(n_blocks)), dtype=torch.int32, device=dev)
        idx = torch.full((n_blocks,), -1, dtype=torch.int32, device=dev)
        meta = torch.tensor([-5, 41], dtype=torch.int32, device=dev)
        _hip.check(lib.eslam_blocks_compact(_hip.ptr(td), n_blocks, _hip.ptr(scratch), _hip.ptr(idx), _hip.ptr(meta), None, 0, st), "compact")
        want = touched.nonzero().squeeze(1)
        n = int(meta[0])
        assert n == want.numel() and int(meta[1]) == 42                   # the stamp counts launches
        assert torch.equal(idx[:n].cpu().long(), want) and torch.equal(td.cpu(), touched)
        # again, with the list's length + stamp written to pinned host memory by the kernel and the map cleared behind the read
        import ctypes
        hp_, dp_ = ctypes.c_void_p(), ctypes.c_void_p()
        _hip.check(lib.eslam_host_meta_alloc(ctypes.byref(hp_), ctypes.byref(dp_)), "host_meta_alloc")
        host = (ctypes.c_int32 * 2).from_address(hp_.value)
        idx2 = torch.full((n_blocks,), -1, dtype=torch.int32, device=dev)
        _hip.check(lib.eslam_blocks_compact(_hip.ptr(td), n_blocks, _hip.ptr(scratch), _hip.ptr(idx2), _hip.ptr(meta), dp_, 1, st), "compact")
        torch.cuda.synchronize()
        assert (host[0], host[1]) == (n, 43) and int(meta[1]) == 43
        assert torch.equal(idx2[:n], idx[:n]) and int(td.count_nonzero()) == 0
        _hip.check(lib.eslam_host_meta_free(hp_), "host_meta_free")
        tail = flat[n_blocks * 32:]
        buf = torch.full((tail_pad + n_blocks * 32,), 7.0, device=dev)
        step = torch.tensor([5, 0, 0, 0], dtype=torch.int32, device=dev)
        _hip.check(lib.eslam_blocks_pack_dev(_hip.ptr(flat), _hip.ptr(idx), _hip.ptr(meta), n_blocks, _hip.ptr(tail), n_tail,
                                             tail_pad, _hip.ptr(buf), _hip.ptr(step), st), "pack")
        assert int(step[0]) == 6                                           # the iteration's random-number step, advanced here
        ref = torch.cat([flat[n_blocks * 32:].cpu(), torch.zeros(tail_pad - n_tail), rows[want].reshape(-1)])
        assert torch.equal(buf[:tail_pad + 32 * n].cpu(), ref)
        assert n == n_blocks or float(buf[tail_pad + 32 * n]) == 7.0         # nothing written past the list
        before = flat.clone()
        buf[:tail_pad + 32 * n].mul_(3.0)
        _hip.check(lib.eslam_blocks_unpack_dev(_hip.ptr(flat), _hip.ptr(idx), _hip.ptr(meta), n_blocks, _hip.ptr(tail), n_tail,
                                               tail_pad, _hip.ptr(buf), st), "unpack")
        exp = before.clone()
        exp[:n_blocks * 32].view(-1, 32)[want.to(dev)] *= 3.0
        exp[n_blocks * 32:] *= 3.0
        assert torch.equal(flat, exp)                                      # unlisted blocks untouched
        _hip.check(lib.eslam_blocks_zero_dev(_hip.ptr(flat), _hip.ptr(idx), _hip.ptr(meta), n_blocks, _hip.ptr(tail), n_tail, st), "zero")
        exp[:n_blocks * 32].view(-1, 32)[want.to(dev)] = 0.0
        exp[n_blocks * 32:] = 0.0
        assert torch.equal(flat, exp)


@pytest.mark.parametrize("scene,zero_frac,state", [("room0", 0.0, "initial"), ("toy", 0.2, "initial"), ("scene0000", 0.1, "trained")])
def test_rays_marking_contains_every_block_the_backward_touches(scene, zero_frac, state):
    """eslam_mark_rays (from ray geometry alone, before anything is sampled) must be a superset of the non-zero 128-byte
    blocks of the plane gradients after a backward with fresh random jitter - the ray-sharded exchange relies on it - and
    equals the tensor-op form the gloo tests run on the CPU."""
    from myslam_amd import harness, ops, parallel
    dev = _dev()
    wl = harness.make_workload(scene, 3000, 40, 8, device=dev, zero_frac=zero_frac, state=state)
    params = wl.plane_list + ops.decoder_params(wl.decoders) + [wl.decoders.beta]
    fg = parallel.FlatGrads(params)
    n_blocks = sum(p.numel() for p in wl.plane_list) // 32
    base = [fg.offsets[i] // 32 for i in range(12)]
    b6 = ops.bound_to_host(wl.scene.bound)
    marked = parallel.mark_rays(None, b6, wl.rays_o, wl.rays_d, wl.gt_depth, wl.truncation, base, n_blocks, planes=wl.planes)
    for it in range(3):                                   # three draws of the jitter / importance samples
        fg.flat.zero_()
        with ops.grad_sink(fg):                           # (forward AND backward inside: a sink is a property of the whole call)
            depth, color, sdf, z = wl.renderer.render_batch_ray(wl.planes, wl.decoders, wl.rays_d, wl.rays_o, dev, wl.truncation,
                                                                gt_depth=wl.gt_depth)
            ((depth * wl._cot[0]).sum() + (color * wl._cot[1]).sum() + (sdf * wl._cot[2]).sum()).backward()
        nz = (fg.flat[:n_blocks * 32].view(-1, 32) != 0).any(1)
        assert int(nz.sum()) > 100
        assert int((nz & ~marked.bool()).sum()) == 0, "a texel received gradient without being marked"
    # conservative, not loose: a few times the touched set, a small part of the planes
    assert int(marked.sum()) <= 5 * int(nz.sum()) and int(marked.sum()) < 0.4 * n_blocks, (int(marked.sum()), int(nz.sum()), n_blocks)
    shapes = [(p.shape[2], p.shape[3]) for p in wl.plane_list]
    cpu = parallel.mark_rays(shapes, b6, wl.rays_o.detach().cpu(), wl.rays_d.detach().cpu(), wl.gt_depth.cpu(), wl.truncation, base, n_blocks)
    # float32 on the device, float64 in the tensor-op form: the boxes differ only where a coordinate sits within rounding of the
    # MARK_EPS margin
    diff = int((cpu.bool() != marked.cpu().bool()).sum())
    assert diff <= 0.002 * int(marked.sum()) + 4, (diff, int(marked.sum()))


def test_loss_set_sizes_equal_the_forward_kernels_counts():
    """eslam_loss_set_sizes (the five set sizes of a whole batch without rendering it: the depth-guided sampler replayed in
    LDS) against the counts the forward kernel's loss epilogue forms for the same rays - injected jitter numbers, in-kernel
    numbers (same seed, step and GLOBAL ray index), a ray mask, 10 % depth-less rays; the slices' counts add up."""
    from myslam_amd import harness, losses, ops, parallel
    dev = _dev()
    wl = harness.make_workload("scene0000", 2000, 88, 8, device=dev, zero_frac=0.1)
    g = torch.Generator().manual_seed(9)
    keep = (torch.rand(wl.R, generator=g) > 0.15).to(dev)
    r = wl.renderer
    cs = list(parallel._ACC_COUNT_SLOTS)
    for mask in (None, keep):
        # in-kernel numbers: set sizes first (the forward kernel advances the step), then the render of the whole batch
        acc = ops.loss_set_sizes(wl.gt_depth, mask, r.n_stratified, r.n_importance, wl.truncation, True)
        with torch.no_grad():
            _, _, _, z, pre = r.render_batch_ray_with_loss(wl.planes, wl.decoders, wl.rays_d, wl.rays_o, dev, wl.truncation,
                                                           wl.gt_depth, wl.gt_color, losses.MAPPING_W, ray_mask=mask)
        assert torch.equal(acc[cs], pre.acc[cs]), (acc[cs], pre.acc[cs])
        assert float(acc[cs].min()) > 0 and float(acc[list(parallel._ACC_SUM_SLOTS)].abs().sum()) == 0.0
        assert torch.equal(acc[cs], parallel.set_sizes_from_z(z, wl.gt_depth, wl.truncation, mask)[cs])
        # two slices rendered under ops.ray_offset draw the whole batch's numbers: their counts add up to the same sizes
        acc2 = ops.loss_set_sizes(wl.gt_depth, mask, r.n_stratified, r.n_importance, wl.truncation, True)
        parts = []
        for lo, hi in ((0, 700), (700, wl.R)):
            if lo > 0:
                ops._rng_state(dev)[0] -= 1            # (same step as the first slice: a sharded rank renders ONE slice per step)
            with torch.no_grad(), ops.ray_offset(lo):
                _, _, _, _, p2 = r.render_batch_ray_with_loss(wl.planes, wl.decoders, wl.rays_d[lo:hi], wl.rays_o[lo:hi], dev,
                                                              wl.truncation, wl.gt_depth[lo:hi], wl.gt_color[lo:hi], losses.MAPPING_W,
                                                              ray_mask=None if mask is None else mask[lo:hi])
            parts.append(p2.acc[cs].clone())
        assert torch.equal(acc2[cs], parts[0] + parts[1])
    # injected numbers
    t_rand = wl._rand[0]
    acc = ops.loss_set_sizes(wl.gt_depth, None, r.n_stratified, r.n_importance, wl.truncation, True, t_rand=t_rand)
    with torch.no_grad():
        _, _, _, z, pre = r.render_batch_ray_with_loss(wl.planes, wl.decoders, wl.rays_d, wl.rays_o, dev, wl.truncation, wl.gt_depth,
                                                       wl.gt_color, losses.MAPPING_W, _rand=wl._rand)
    assert torch.equal(acc[cs], pre.acc[cs])


@pytest.mark.parametrize("cameras,R", [(1, 4096), (1, 77), (3, 3000), (1, 9000)])
def test_ray_order_is_a_permutation_and_groups_neighbours(cameras, R):
    """eslam_ray_order: three permutations (one per plane orientation) of every chunk of 8192 rays - for one origin keyed on the
    azimuth of the direction projected into the plane, for a camera-major batch of several origins on (camera, azimuth), for
    origins in random order on a 3-D Morton code."""
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
    assert perm.numel() == 3 * R + 4                            # one order per plane orientation (xy, xz, yz) + the fan's extent
    fan = perm[3 * R:3 * R + 3].view(torch.float32).cpu()
    assert (fan > 0.1).all() and (fan < 6.3).all() if cameras == 1 else (fan == 0).all()
    for o, (a, b) in enumerate([(0, 1), (0, 2), (1, 2)]):
        p = perm[o * R:(o + 1) * R].cpu().long()
        for lo in range(0, R, 8192):                            # chunks are ordered independently
            hi = min(R, lo + 8192)
            assert torch.equal(torch.sort(p[lo:hi]).values, torch.arange(lo, hi))
        if cameras == 1 and R >= 1000:
            # one origin: order o sorts the rays by the azimuth of their direction projected into plane o - neighbours in the
            # order are neighbours in that angle (against the projected mean direction: no wrap-around inside the fan)
            n1 = min(R, 8192)
            m = torch.nn.functional.normalize(d, dim=1).sum(0)
            ang = torch.atan2(m[a] * d[:, b] - m[b] * d[:, a], m[a] * d[:, a] + m[b] * d[:, b])
            step_sorted = (ang[p[1:n1]] - ang[p[:n1 - 1]]).abs().mean()
            step_given = (ang[1:n1] - ang[:n1 - 1]).abs().mean()
            assert float(step_sorted) < 0.02 * float(step_given)
            # non-decreasing up to the key's quantisation, except at the ONE seam where the angle wraps (the kernel measures it
            # against the chunk's own projected mean direction: a camera looking along the plane's normal fills the circle)
            assert int(((ang[p[1:n1]] - ang[p[:n1 - 1]]) < -2e-3).sum()) <= 1
    # (several origins in random order: the three orders are the same Morton order up to the arbitrary order inside a key's cell)
    if cameras > 1 and R <= 8192:
        # a keyframe window as get_samples builds it: camera-major.  Order o then keeps every camera's rays together and sorts
        # them by the azimuth of their direction projected into plane o, against the camera's own projected mean direction
        cam_sorted, _ = torch.sort(cam)
        ro2 = org[cam_sorted].to(dev)
        perm2, side2 = ops.ray_order_async(ro2, rd)
        torch.cuda.current_stream().wait_stream(side2)
        assert float(perm2[3 * R:3 * R + 3].view(torch.float32).abs().max()) == 0.0
        for o, (a, b) in enumerate([(0, 1), (0, 2), (1, 2)]):
            p = perm2[o * R:(o + 1) * R].cpu().long()
            assert torch.equal(torch.sort(p).values, torch.arange(R))
            assert bool((cam_sorted[p][1:] >= cam_sorted[p][:-1]).all())            # cameras stay together, in their order
            dn = torch.nn.functional.normalize(d, dim=1)
            for c in range(cameras):
                sel = p[cam_sorted[p] == c]
                if len(sel) < 50:
                    continue
                m = dn[cam_sorted == c].sum(0)
                ang = torch.atan2(m[a] * d[sel, b] - m[b] * d[sel, a], m[a] * d[sel, a] + m[b] * d[sel, b])
                assert int(((ang[1:] - ang[:-1]) < -0.05).sum()) <= 1       # sorted up to the key's quantisation, one wrap at most


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
    # the reproducibility contract (ops._rng_seed): the key folds torch's CPU seed with the device generator's, or is given by
    # ops.seed; a new key restarts the step counter, so seeding twice with one value reproduces the samples
    torch.manual_seed(77)
    za = render()
    assert int(ops._rng_state(dev)[0]) == 1 and not torch.equal(za, z1)
    torch.manual_seed(1234)
    assert torch.equal(render(), z1)
    torch.cuda.manual_seed(5)                                     # the device generator alone
    zc = render()
    assert int(ops._rng_state(dev)[0]) == 1 and not torch.equal(zc, z1)
    ops.seed(99)
    zd = render()
    ops.seed(99)
    assert torch.equal(render(), zd) and not torch.equal(zd, z1)
    ops.seed(None)
    # rays lo .. of a batch rendered under ops.ray_offset(lo) draw the numbers of the whole batch's rows lo ..
    ops.seed(4321)
    zw = render()
    ops.seed(4321)                                                # (restarts the step counter)
    with torch.no_grad(), ops.ray_offset(500):
        zs = r.render_batch_ray(wl.planes, wl.decoders, wl.rays_d[500:900], wl.rays_o[500:900], dev, wl.truncation,
                                gt_depth=wl.gt_depth[500:900])[3]
    assert torch.equal(zs[has[500:900]], zw[500:900][has[500:900]])          # depth-guided rows: bit for bit
    assert torch.allclose(zs, zw[500:900], rtol=1e-5, atol=1e-6)
    ops.seed(None)


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
