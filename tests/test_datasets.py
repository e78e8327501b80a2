"""Dataset readers (myslam_amd/src/utils/datasets.py; reference src/utils/datasets.py:54-263) on tiny sequences written to
tmp_path in each on-disk format.  CPU only.  Pinned to the reference where the reference runs without OpenCV
(tests/golden/datasets_lists.npz: TUM association / thinning / re-basing, Replica and ScanNet trajectories); the image
path (PIL instead of cv2) is checked against known answers written into the files."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch
from PIL import Image

from myslam_amd.src.utils import datasets as ds
from tests import helpers as hp
from tests.golden.make_golden import tum_lists, write_tum_lists


def _cfg(dataset, folder, H, W, png_depth_scale=6553.5, crop_edge=0, **cam):
    c = dict(H=H, W=W, fx=float(W), fy=float(W), cx=(W - 1) / 2, cy=(H - 1) / 2, png_depth_scale=png_depth_scale,
             crop_edge=crop_edge)
    c.update(cam)
    return dict(dataset=dataset, cam=c, data=dict(input_folder=str(folder)))


ARGS = SimpleNamespace(input_folder=None)


def _write_frame(color_path, depth_path, color_u8, depth_u16):
    os.makedirs(os.path.dirname(color_path), exist_ok=True)
    os.makedirs(os.path.dirname(depth_path), exist_ok=True)
    Image.fromarray(color_u8).save(color_path, quality=95) if color_path.endswith(".jpg") else Image.fromarray(color_u8).save(color_path)
    Image.fromarray(depth_u16).save(depth_path)


def test_pose_lists_match_the_reference(tmp_path):
    """TUM_RGBD.loadtum, Replica.load_poses, ScanNet.load_poses against outputs of the reference's own functions."""
    fx = hp.load("datasets_lists")
    t_rgb, t_dep, t_pose, vecs = tum_lists()
    write_tum_lists(str(tmp_path), t_rgb, t_dep, t_pose, vecs)
    images, depths, poses = ds.TUM_RGBD.loadtum(ds.TUM_RGBD.__new__(ds.TUM_RGBD), str(tmp_path), frame_rate=32)
    assert [os.path.relpath(p, tmp_path) for p in images] == fx["tum_images"].tolist()
    assert [os.path.relpath(p, tmp_path) for p in depths] == fx["tum_depths"].tolist()
    assert np.array_equal(torch.stack(poses).numpy(), fx["tum_poses"])          # bit for bit
    assert np.array_equal(poses[0].numpy(), np.diag([1, -1, -1, 1]).astype(np.float32))   # first pose: identity, flipped
    # max_dt: every kept triple is within 0.08 s; a stricter gap drops frames
    assoc = ds.TUM_RGBD.associate_frames(t_rgb, t_dep, t_pose)
    assert all(abs(t_dep[j] - t_rgb[i]) < 0.08 and abs(t_pose[k] - t_rgb[i]) < 0.08 for i, j, k in assoc)
    assert len(ds.TUM_RGBD.associate_frames(t_rgb, t_dep, t_pose, max_dt=0.015)) < len(assoc)
    assert len(ds.TUM_RGBD.associate_frames(t_rgb, t_dep, None)[0]) == 2
    # Replica traj.txt / ScanNet pose/*.txt
    mats = fx["mats"]
    with open(tmp_path / "traj.txt", "w") as f:
        f.write("".join(" ".join(f"{x:.9e}" for x in m.reshape(-1)) + "\n" for m in mats))
    r = SimpleNamespace(n_img=len(mats))
    ds.Replica.load_poses(r, str(tmp_path / "traj.txt"))
    assert np.array_equal(torch.stack(r.poses).numpy(), fx["replica_poses"])
    os.makedirs(tmp_path / "pose")
    for k, m in enumerate(mats):
        with open(tmp_path / "pose" / f"{k}.txt", "w") as f:
            f.write("".join(" ".join(f"{x:.6f}" for x in row) + "\n" for row in m))
    s = SimpleNamespace()
    ds.ScanNet.load_poses(s, str(tmp_path / "pose"))
    assert np.array_equal(torch.stack(s.poses).numpy(), fx["scannet_poses"])
    # the flip: columns 1 and 2 of the rotation negated (c2w[:3, 1:3] *= -1), translation untouched
    assert np.allclose(fx["replica_poses"][3][:3, 1:3], -mats[3][:3, 1:3].astype(np.float32))
    assert np.allclose(fx["replica_poses"][3][:3, [0, 3]], mats[3][:3, [0, 3]].astype(np.float32))


def test_replica_items(tmp_path):
    H, W, n = 12, 20, 4
    rng = np.random.default_rng(0)
    depths = rng.integers(0, 65535, size=(n, H, W), dtype=np.uint16)
    depths[:, 0, 0] = 0
    colors = np.zeros((n, H, W, 3), np.uint8)       # smooth images (JPEG is lossy on noise): R ramps along x, G along y, B = 40 k
    colors[..., 0] = (np.arange(W) * 12)[None, None]
    colors[..., 1] = (np.arange(H) * 20)[None, :, None]
    colors[..., 2] = (40 * np.arange(n))[:, None, None]
    for k in range(n):
        _write_frame(str(tmp_path / "results" / f"frame{k:06d}.jpg"), str(tmp_path / "results" / f"depth{k:06d}.png"),
                     colors[k], depths[k])
    with open(tmp_path / "traj.txt", "w") as f:
        for k in range(n + 2):                     # more lines than frames, as in the real files
            m = np.eye(4)
            m[:3, 3] = [k, 2 * k, 3 * k]
            f.write(" ".join(f"{x:.6e}" for x in m.reshape(-1)) + "\n")
    d = ds.get_dataset(_cfg("replica", tmp_path, H, W, crop_edge=0), ARGS, scale=1.0, device="cpu")
    assert isinstance(d, ds.Replica) and len(d) == n
    idx, color, depth, pose = d[2]
    assert idx == 2 and depth.dtype == torch.float32 and tuple(depth.shape) == (H, W) and tuple(color.shape) == (H, W, 3)
    assert torch.equal(depth, torch.from_numpy(depths[2].astype(np.float32) / 6553.5))       # png depth scale, exact
    assert float(depth[0, 0]) == 0.0                                                         # sensor holes stay 0
    # colour: RGB order, [0,1], exactly what PIL decodes from the file
    ref = np.asarray(Image.open(tmp_path / "results" / "frame000002.jpg").convert("RGB")) / 255.
    assert np.array_equal(color.numpy(), ref) and color.dtype == torch.float64
    assert np.abs(color.numpy() * 255 - colors[2]).max() < 12         # (JPEG is lossy, but it IS that image, channels in RGB order)
    assert torch.equal(pose, torch.tensor([[1., 0, 0, 2], [0, -1, 0, 4], [0, 0, -1, 6], [0, 0, 0, 1]]))
    # scale multiplies depth and translation; crop_edge trims the border
    d2 = ds.Replica(_cfg("replica", tmp_path, H, W, crop_edge=2), ARGS, scale=2.0, device="cpu")
    _, c2, z2, p2 = d2[1]
    assert tuple(z2.shape) == (H - 4, W - 4) and tuple(c2.shape) == (H - 4, W - 4, 3)
    assert torch.equal(z2, torch.from_numpy(depths[1].astype(np.float32) / 6553.5)[2:-2, 2:-2] * 2.0)
    assert torch.equal(p2[:3, 3], torch.tensor([2., 4, 6]))


def test_scannet_items_numeric_order_resize_and_crop_size(tmp_path):
    Hd, Wd, Hc, Wc = 12, 16, 24, 32             # ScanNet: colour images are larger than depth images
    order = [0, 1, 2, 10, 100]                  # lexicographic order would be 0, 1, 10, 100, 2
    for k in order:
        col = np.zeros((Hc, Wc, 3), np.uint8)
        col[..., 0] = k                          # frame number in the red channel
        col[..., 1] = np.arange(Wc, dtype=np.uint8)[None] * 4
        dep = np.full((Hd, Wd), 1000 + k, np.uint16)
        os.makedirs(tmp_path / "color", exist_ok=True)
        os.makedirs(tmp_path / "depth", exist_ok=True)
        os.makedirs(tmp_path / "pose", exist_ok=True)
        Image.fromarray(col).save(tmp_path / "color" / f"{k}.jpg", quality=100, subsampling=0)
        Image.fromarray(dep).save(tmp_path / "depth" / f"{k}.png")
        m = np.eye(4)
        m[0, 3] = k
        with open(tmp_path / "pose" / f"{k}.txt", "w") as f:
            f.write("".join(" ".join(f"{x:.6f}" for x in row) + "\n" for row in m))
    cfg = _cfg("scannet", tmp_path, Hd, Wd, png_depth_scale=1000.0, crop_edge=1, crop_size=[10, 14])
    d = ds.get_dataset(cfg, ARGS, scale=1.0, device="cpu")
    assert [int(os.path.basename(p)[:-4]) for p in d.color_paths] == order
    for n, k in enumerate(order):
        _, color, depth, pose = d[n]
        assert tuple(depth.shape) == (8, 12) and tuple(color.shape) == (8, 12, 3)       # crop_size 10 x 14, then 1-px edge
        assert torch.allclose(depth, torch.full((8, 12), (1000 + k) / 1000.0))
        assert abs(float(color[..., 0].mean()) * 255 - k) < 1.5
        assert float(pose[0, 3]) == k
    # the colour resize is bilinear on pixel centres: a horizontal ramp stays a ramp with the same end-to-end slope
    _, color, _, _ = ds.ScanNet(_cfg("scannet", tmp_path, Hd, Wd, png_depth_scale=1000.0), ARGS, 1.0, "cpu")[0]
    g = color[5, :, 1].numpy() * 255
    assert tuple(color.shape) == (Hd, Wd, 3)
    assert np.allclose(np.diff(g)[1:-1], 8.0, atol=0.6)          # 32 -> 16 columns: steps of 2 source pixels x 4


def test_tum_items_and_undistort(tmp_path):
    H, W = 16, 24
    t_rgb = 100.0 + np.arange(6) / 10.0
    os.makedirs(tmp_path / "rgb")
    os.makedirs(tmp_path / "depth")
    with open(tmp_path / "rgb.txt", "w") as fr, open(tmp_path / "depth.txt", "w") as fd, \
            open(tmp_path / "groundtruth.txt", "w") as fp:
        fr.write("# color images\n# file\n# timestamp filename\n")
        fd.write("# depth maps\n# file\n# timestamp filename\n")
        fp.write("# timestamp tx ty tz qx qy qz qw\n")
        for k, t in enumerate(t_rgb):
            col = np.zeros((H, W, 3), np.uint8)
            col[..., 2] = 10 * k
            Image.fromarray(col).save(tmp_path / "rgb" / f"{t:.6f}.png")
            Image.fromarray(np.full((H, W), 5000 * (k + 1), np.uint16)).save(tmp_path / "depth" / f"{t + 0.01:.6f}.png")
            fr.write(f"{t:.6f} rgb/{t:.6f}.png\n")
            fd.write(f"{t + 0.01:.6f} depth/{t + 0.01:.6f}.png\n")
            fp.write(f"{t + 0.002:.4f} {k:.4f} 0.0000 0.0000 0.0000 0.0000 0.0000 1.0000\n")
    # (np.loadtxt treats '#' lines as comments, as it does for the real TUM lists)
    cfg = _cfg("tumrgbd", tmp_path, H, W, png_depth_scale=5000.0, distortion=[0.0, 0.0, 0.0, 0.0, 0.0])
    d = ds.get_dataset(cfg, ARGS, scale=1.0, device="cpu")
    assert len(d) == 6
    for k in range(6):
        _, color, depth, pose = d[k]
        assert torch.equal(depth, torch.full((H, W), float(k + 1)))
        assert np.allclose(color[..., 2].numpy() * 255, 10 * k) and float(color[..., :2].abs().max()) == 0.0   # RGB, not BGR
        want = torch.diag(torch.tensor([1., -1, -1, 1]))
        want[0, 3] = float(k)                    # translation relative to the first frame
        assert torch.allclose(pose, want, atol=1e-6)
    # undistort: zero coefficients are the identity; a radial term moves pixels towards the centre as the model says
    img = np.zeros((H, W, 3), np.uint8)
    img[..., 0] = (np.arange(W)[None] * 10).astype(np.uint8)
    K = ds.as_intrinsics_matrix([20.0, 20.0, (W - 1) / 2, (H - 1) / 2])
    assert np.array_equal(ds.undistort(img, K, np.zeros(5)), img)
    k1 = 0.2
    out = ds.undistort(img, K, np.array([k1, 0, 0, 0, 0.0]))
    v, u = 3, 20
    x, y = (u - K[0, 2]) / 20.0, (v - K[1, 2]) / 20.0
    r2 = x * x + y * y
    src_u = 20.0 * x * (1 + k1 * r2) + K[0, 2]                  # where the distorted image shows this ray
    assert abs(float(out[v, u, 0]) - 10 * src_u) <= 1.0          # the ramp is linear in u: bilinear sampling is exact to rounding
    assert out[0, 0, 0] == 0 or out.shape == img.shape          # corners map outside the image: filled with 0


def test_seq_sampler_and_named_but_absent_exr_reader():
    """SeqSampler (reference datasets.py:33-48; imported by the reference's Mapper from this module): every step-th index,
    plus the last index when asked for and not already there."""
    assert list(ds.SeqSampler(10, 4)) == [0, 4, 8, 9] and len(ds.SeqSampler(10, 4)) == 4
    assert list(ds.SeqSampler(9, 4)) == [0, 4, 8]                       # the last index is already a multiple of the step
    assert list(ds.SeqSampler(10, 4, include_last=False)) == [0, 4, 8]
    assert list(ds.SeqSampler(1, 5)) == [0] and list(ds.SeqSampler(0, 5)) == []
    loader = torch.utils.data.DataLoader(list(range(10)), batch_size=1, sampler=ds.SeqSampler(10, 3))
    assert [int(b[0]) for b in loader] == [0, 3, 6, 9]
    with pytest.raises(ImportError):
        ds.readEXR_onlydepth("x.exr")
    with pytest.raises(KeyError):
        ds.get_dataset(dict(dataset="kitti"), ARGS, 1.0)
