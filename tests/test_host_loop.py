"""CPU tests of the host-side pieces around the hot path: the analytic RGB-D sequence, the ATE evaluator, and the
tracking + mapping loop's control flow (keyframe window, joint optimisation switch, pose bookkeeping) driven by the
oracle backend at a tiny size."""
import numpy as np
import torch


def test_ate_alignment_known_answers():
    from myslam_amd import eval_ate
    rng = np.random.default_rng(0)
    gt = np.tile(np.eye(4), (20, 1, 1))
    gt[:, :3, 3] = rng.normal(size=(20, 3))
    # a rigidly moved copy aligns to zero error; without alignment the offset shows
    th = 0.7
    Rz = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]])
    est = gt.copy()
    est[:, :3, 3] = gt[:, :3, 3] @ Rz.T + np.array([1.0, -2.0, 0.5])
    assert eval_ate.evaluate(est, gt)["rmse"] < 1e-12
    assert eval_ate.evaluate(est, gt, do_align=False)["rmse"] > 1.0
    # a mirrored trajectory must NOT align to zero (reflection-free solution, eval_ate.py:88-90)
    mir = gt.copy()
    mir[:, 0, 3] *= -1
    assert eval_ate.evaluate(mir, gt)["rmse"] > 0.1
    # isotropic noise of sigma per axis -> rmse ~ sigma * sqrt(3)
    noisy = gt.copy()
    noisy[:, :3, 3] += rng.normal(scale=0.01, size=(20, 3))
    r = eval_ate.evaluate(noisy, gt)
    assert 0.008 < r["rmse"] < 0.03 and r["median"] <= r["max"]
    rot, trans, err = eval_ate.align(est[:, :3, 3].T, gt[:, :3, 3].T)
    assert np.allclose(rot @ rot.T, np.eye(3), atol=1e-12) and np.linalg.det(rot) > 0 and err.shape == (20,)


def test_analytic_sequence_geometry():
    from myslam_amd import scene as scn, synthscene
    sc = scn.make_scene("toy")
    room = synthscene.AnalyticRoom(sc.bound)
    poses = synthscene.trajectory(8, sc.bound)
    for p in poses:                                     # rigid, right-handed, inside the room
        R = p[:3, :3]
        assert torch.allclose(R @ R.T, torch.eye(3), atol=1e-6) and abs(float(torch.det(R)) - 1) < 1e-5
        assert bool(((p[:3, 3] > room.lo.float()) & (p[:3, 3] < room.hi.float())).all())
    depth, color = synthscene.render_frame(room, sc, poses[0])
    assert depth.shape == (sc.H, sc.W) and color.shape == (sc.H, sc.W, 3)
    assert float(depth.min()) > 0.05 and 0.0 <= float(color.min()) and float(color.max()) <= 1.0
    # back-project every pixel with its depth: the point lies on a wall of the box or on a sphere
    i, j = torch.meshgrid(torch.arange(sc.W, dtype=torch.float32), torch.arange(sc.H, dtype=torch.float32), indexing="xy")
    dirs = torch.stack([(i - sc.cx) / sc.fx, -(j - sc.cy) / sc.fy, -torch.ones_like(i)], -1) @ poses[0][:3, :3].T
    pts = (poses[0][:3, 3] + dirs * depth[..., None]).double()
    wall = torch.minimum((pts - room.lo).abs().min(-1).values, (pts - room.hi).abs().min(-1).values)
    sph = torch.stack([((pts - c).norm(dim=-1) - r).abs() for c, r in room.spheres], -1).min(-1).values
    assert float(torch.minimum(wall, sph).max()) < 1e-4
    # the same frame twice is identical; holes are zeros
    d2, _ = synthscene.render_frame(room, sc, poses[0], hole_frac=0.1, seed=3)
    assert 0.05 < float((d2 == 0).float().mean()) < 0.15 and torch.equal(d2[d2 > 0], depth[d2 > 0])


def test_loop_control_flow_on_the_oracle_backend():
    """11 frames, mapping every 2nd: the window grows 1, 1, 3, 4, 5, 6 frames (overlap-selected keyframes + the last two +
    the current one, Mapper.py:236-247), poses join the optimisation once more than 4 keyframes exist (Mapper.py:416),
    every frame gets a pose, the first pose is the ground truth."""
    from myslam_amd import scene as scn, slam, synthscene
    from tests.oracle_backend import OracleBackend
    sc = scn.make_scene("toy")
    cfg = slam.SlamConfig(tracking_pixels=60, tracking_iters=2, ignore_edge_H=10, ignore_edge_W=10, mapping_pixels=120,
                          iters_first=3, iters=2, every_frame=2, keyframe_every=2, mapping_window_size=4)
    frames = synthscene.make_sequence(sc, 11)
    torch.manual_seed(0)
    s = slam.Slam(sc, cfg, device="cpu", backend=OracleBackend(sc))
    windows = []
    orig = s.be.get_samples

    def spy(H0, H1, W0, W1, n, H, W, fx, fy, cx, cy, c2ws, depths, colors, device):
        if H0 == 0 and n == 120 // c2ws.shape[0]:        # mapping iterations (keyframe selection draws 50 rays of one frame)
            windows.append((c2ws.shape[0], n, bool(c2ws.requires_grad)))
        return orig(H0, H1, W0, W1, n, H, W, fx, fy, cx, cy, c2ws, depths, colors, device)

    s.be.get_samples = spy
    est = s.run(frames)
    assert len(est) == 11 and torch.equal(est[0], frames[0][3])
    assert s.keyframe_list == [0, 2, 4, 6, 8, 10]
    assert s.stats["mapping_iters"] == 3 + 2 * 5 and s.stats["tracking_iters"] == 2 * 10
    assert len(windows) == 3 + 2 * 5
    per_call = [windows[0]] + windows[3::2]              # first iteration of each mapping call
    assert [w for w, _, _ in per_call] == [1, 1, 3, 4, 5, 6]
    assert [g for _, _, g in per_call] == [False, False, False, False, False, True]     # joint_opt once len(keyframes) > 4
    for e in est:
        R = e[:3, :3]
        assert torch.allclose(R @ R.T, torch.eye(3), atol=1e-4)


def test_checkpoint_round_trip_in_the_reference_format(tmp_path):
    from myslam_amd import checkpoint
    from myslam_amd.src.networks.decoders import Decoders
    torch.manual_seed(3)
    dec = Decoders(learnable_beta=True)
    gt = [torch.eye(4) for _ in range(5)]
    est = [torch.eye(4) + 0.01 * k for k in range(5)]
    path = str(tmp_path / "00004.tar")
    checkpoint.save(path, dec, gt, est, [0, 4], 4)
    raw = torch.load(path, weights_only=True)
    assert set(raw) == {"decoder_state_dict", "gt_c2w_list", "estimate_c2w_list", "keyframe_list", "idx"}      # Logger.py:41-47
    assert set(raw["decoder_state_dict"]) == {"beta", "linears.0.weight", "linears.0.bias", "linears.1.weight",
                                              "linears.1.bias", "c_linears.0.weight", "c_linears.0.bias",
                                              "c_linears.1.weight", "c_linears.1.bias", "output_linear.weight",
                                              "output_linear.bias", "c_output_linear.weight", "c_output_linear.bias"}
    dec2 = Decoders(learnable_beta=True)
    ck = checkpoint.load(path, dec2)
    assert ck["idx"] == 4 and ck["keyframe_list"] == [0, 4] and ck["estimate_c2w_list"].shape == (5, 4, 4)
    for (k, a), (_, b) in zip(dec.state_dict().items(), dec2.state_dict().items()):
        assert torch.equal(a, b), k
    torch.save({"something": 1}, path)
    try:
        checkpoint.load(path)
        assert False
    except KeyError:
        pass
