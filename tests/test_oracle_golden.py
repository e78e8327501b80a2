"""Pin the oracle: replay every golden fixture (outputs of the reference itself, see
tests/golden/make_golden.py) through oracle/eslam_oracle.py on the CPU.

Tolerance: both sides are float32 with different op orders (explicit bilinear vs grid_sample, einsum
vs broadcast-sum), so agreement is at float32 rounding: 2e-5 relative to the tensor's max for
per-element outputs, 1e-4 for gradient tensors (sums over up to 786k samples).
"""
import numpy as np
import pytest
import torch

from oracle import eslam_oracle as orc
from tests import helpers as hp

OUT_RTOL = 2e-5
GRAD_RTOL = 1e-4


def run_oracle(fx, dtype=torch.float32):
    sc, planes = hp.scene_and_planes(fx, dtype=dtype, channels_last=False, requires_grad=True)
    params = hp.params_from(fx, dtype=dtype, requires_grad=True)
    if bool(fx["beta_is_param"]):
        beta = torch.tensor([float(fx["beta"])], dtype=dtype, requires_grad=True)
    else:
        beta = float(fx["beta"])
    t_rand, t_uni, u = hp.rand_inputs(fx)
    cv = lambda a: None if a is None else a.to(dtype)
    ro = torch.from_numpy(fx["rays_o"]).to(dtype).requires_grad_(True)
    rd = torch.from_numpy(fx["rays_d"]).to(dtype).requires_grad_(True)
    gd = torch.from_numpy(fx["gt_depth"]).to(dtype)
    gc = torch.from_numpy(fx["gt_color"]).to(dtype)
    tr = float(fx["truncation"])
    depth, color, sdf, z = orc.render_batch_ray(
        planes, params, beta, sc.bound, rd, ro, tr, gd, int(fx["n_stratified"]), int(fx["n_importance"]),
        cv(t_rand), cv(t_uni), cv(u))
    loss_fn = orc.mapping_loss if str(fx["loss_kind"]) == "mapping" else orc.tracking_loss
    loss = loss_fn(depth, color, sdf, z, gd, gc, tr)
    loss.backward()
    return dict(depth=depth, color=color, sdf=sdf, z=z, loss=loss, ro=ro, rd=rd, params=params, beta=beta,
                planes=planes)


@pytest.mark.parametrize("case", hp.RENDER_CASES)
def test_render_matches_reference(case):
    fx = hp.load(case)
    r = run_oracle(fx)
    pr = fx["probe"]
    assert hp.rel_err(r["z"].detach().numpy()[pr], fx["z_vals"]) <= 2e-6
    assert hp.rel_err(r["sdf"].detach().numpy()[pr], fx["sdf"]) <= OUT_RTOL
    assert hp.rel_err(r["depth"].detach().numpy()[pr], fx["depth"]) <= OUT_RTOL
    assert hp.rel_err(r["color"].detach().numpy()[pr], fx["color"]) <= OUT_RTOL
    for name, ten in (("depth_sum", r["depth"]), ("color_sum", r["color"]), ("sdf_sum", r["sdf"]), ("z_sum", r["z"])):
        got = float(ten.detach().double().sum())
        ref = float(fx[name])
        assert abs(got - ref) <= 1e-5 * max(abs(ref), float(ten.detach().abs().double().sum()) * 1e-1), name
    assert abs(float(r["loss"]) - float(fx["loss"])) <= 1e-5 * abs(float(fx["loss"]))
    # gradients
    for k, p in r["params"].items():
        assert hp.rel_err(p.grad.numpy(), fx["grad:" + k]) <= GRAD_RTOL, k
    if bool(fx["beta_is_param"]):
        assert hp.rel_err(r["beta"].grad.numpy(), fx["grad:beta"]) <= GRAD_RTOL
    assert hp.rel_err(r["ro"].grad.numpy()[pr], fx["g_rays_o"]) <= GRAD_RTOL
    assert hp.rel_err(r["rd"].grad.numpy()[pr], fx["g_rays_d"]) <= GRAD_RTOL
    assert hp.rel_err(r["ro"].grad.double().sum(0).numpy(), fx["g_rays_o_sum"]) <= GRAD_RTOL
    hp.check_plane_probes(fx, [p.grad for p in hp.flat_planes(r["planes"])], rtol=GRAD_RTOL)


def test_get_samples_matches_reference():
    fx = hp.load("get_samples_room0_b3")
    from myslam_amd import scene as scn, synth
    sc = scn.make_scene("room0")
    b, n = int(fx["b"]), int(fx["n"])
    H0, H1, W0, W1 = (int(v) for v in fx["window"])
    call = fx["rand_calls"].tolist()[0].split(";")
    assert call[0] == "randint" and int(call[2]) == (H1 - H0) * (W1 - W0)
    idx = torch.from_numpy(synth.hash_randint(int(call[2]), (b * n,), int(call[1])))
    depth_img = torch.from_numpy(np.stack([synth.depth_image(sc.H, sc.W, 20 + i) for i in range(b)]))
    color_img = torch.from_numpy(np.stack([synth.color_image(sc.H, sc.W, 30 + i) for i in range(b)]))
    c2ws = torch.from_numpy(fx["c2ws"]).requires_grad_(True)
    ro, rd, d, c = orc.rays_from_pixels(idx, H0, H1, W0, W1, sc.fx, sc.fy, sc.cx, sc.cy, c2ws, depth_img, color_img)
    assert np.array_equal(d.numpy(), fx["depth"])
    assert np.array_equal(c.numpy(), fx["color"])
    assert hp.rel_err(ro.detach().numpy(), fx["rays_o"]) <= 1e-7
    assert hp.rel_err(rd.detach().numpy(), fx["rays_d"]) <= 1e-6
    wo = torch.from_numpy(synth.hash_uniform(tuple(ro.shape), 61_000)) - 0.5
    wd = torch.from_numpy(synth.hash_uniform(tuple(rd.shape), 61_001)) - 0.5
    ((ro * wo).sum() + (rd * wd).sum()).backward()
    assert hp.rel_err(c2ws.grad.numpy(), fx["g_c2ws"]) <= 1e-5


def test_get_rays_matches_reference():
    fx = hp.load("get_rays_room0")
    from myslam_amd import scene as scn
    sc = scn.make_scene("room0")
    ro, rd = orc.rays_full_image(sc.H, sc.W, sc.fx, sc.fy, sc.cx, sc.cy, torch.from_numpy(fx["c2w"]))
    sel = fx["sel"]
    assert hp.rel_err(ro.reshape(-1, 3).numpy()[sel], fx["rays_o"]) <= 1e-7
    assert hp.rel_err(rd.reshape(-1, 3).numpy()[sel], fx["rays_d"]) <= 1e-6


def test_decoders_and_sample_pdf_match_reference():
    fx = hp.load("decoders_room0_points")
    from myslam_amd import scene as scn, synth
    sc = scn.make_scene("room0")
    planes = scn.synth_planes(sc, channels_last=False)
    params = hp.params_from(fx)
    p = torch.from_numpy(fx["points"])
    assert hp.rel_err(orc.normalize_points(p, sc.bound).numpy(), fx["p_nor"]) <= 1e-7
    raw = orc.decode(p, planes, params, sc.bound)
    assert hp.rel_err(raw.numpy(), fx["raw"]) <= OUT_RTOL
    call = fx["rand_calls"].tolist()[0].split(";")
    u = torch.from_numpy(synth.hash_uniform((int(call[2]), int(call[3])), int(call[1])))
    smp = orc.invert_cdf(torch.from_numpy(fx["bins"]), torch.from_numpy(fx["weights"]), u)
    assert hp.rel_err(smp.numpy(), fx["pdf_samples"]) <= 1e-6


def test_oracle_float64_agrees_with_float32():
    """The float64 oracle is what tight gradient checks of the kernels use; make sure it is the same function."""
    fx = hp.load("room0_200x32")
    r32 = run_oracle(fx, torch.float32)
    r64 = run_oracle(fx, torch.float64)
    assert hp.rel_err(r32["sdf"].detach().numpy(), r64["sdf"].detach().numpy()) <= 1e-5
    assert hp.rel_err(r32["depth"].detach().numpy(), r64["depth"].detach().numpy()) <= 1e-5
    for k in r32["params"]:
        assert hp.rel_err(r32["params"][k].grad.numpy(), r64["params"][k].grad.numpy()) <= 1e-4


def test_grid_sample_variant_of_the_oracle_matches_too(monkeypatch):
    """bench.py times the oracle with F.grid_sample (what the reference calls); same function, same fixture."""
    monkeypatch.setattr(orc, "BILINEAR_IMPL", "grid_sample")
    fx = hp.load("room0_200x40_zero15")
    r = run_oracle(fx)
    pr = fx["probe"]
    assert hp.rel_err(r["sdf"].detach().numpy()[pr], fx["sdf"]) <= OUT_RTOL
    assert hp.rel_err(r["depth"].detach().numpy()[pr], fx["depth"]) <= OUT_RTOL
    for k, p in r["params"].items():
        assert hp.rel_err(p.grad.numpy(), fx["grad:" + k]) <= GRAD_RTOL, k
    assert hp.rel_err(r["rd"].grad.numpy()[pr], fx["g_rays_d"]) <= GRAD_RTOL
    hp.check_plane_probes(fx, [p.grad for p in hp.flat_planes(r["planes"])], rtol=GRAD_RTOL)


def test_render_img_matches_reference():
    """Renderer.render_img (src/utils/Renderer.py:155-204) restated with the oracle's pieces: full-image rays, the
    reference's ray batching (ragged last chunk), depth as float64."""
    fx = hp.load("render_img_room0_30x44")
    from myslam_amd import scene as scn, synth
    sc = scn.make_scene("room0")
    planes = scn.synth_planes(sc, channels_last=False)
    params = hp.params_from(fx)
    H, W = int(fx["H"]), int(fx["W"])
    ro, rd = orc.rays_full_image(H, W, float(fx["fx"]), float(fx["fy"]), float(fx["cx"]), float(fx["cy"]),
                                 torch.from_numpy(fx["c2w"]))
    ro, rd = ro.reshape(-1, 3), rd.reshape(-1, 3)
    gd = torch.from_numpy(synth.depth_image(H, W, int(fx["depth_stream"]), float(fx["zero_frac"]))).reshape(-1)
    assert int((gd == 0).sum()) == int(fx["n_zero"]) > 0
    rands = hp.render_img_chunk_rands(fx, gd.numpy())
    bs, ns, ni = int(fx["ray_batch_size"]), int(fx["n_stratified"]), int(fx["n_importance"])
    assert len(rands) == 3 and (H * W) % bs != 0
    depth, color = [], []
    with torch.no_grad():
        for k, i in enumerate(range(0, H * W, bs)):
            d, c, _, _ = orc.render_batch_ray(planes, params, 10.0, sc.bound, rd[i:i + bs], ro[i:i + bs], float(fx["truncation"]),
                                              gd[i:i + bs], ns, ni, *rands[k])
            depth.append(d.double())
            color.append(c)
    depth, color = torch.cat(depth).reshape(H, W), torch.cat(color).reshape(H, W, 3)
    assert hp.rel_err(depth.numpy(), fx["depth"]) <= OUT_RTOL
    assert hp.rel_err(color.numpy(), fx["color"]) <= OUT_RTOL
    ok, info = hp.elementwise_close(depth.numpy(), fx["depth"], rtol=1e-4)
    assert ok, info
