"""RGB-D sequence readers: same classes, constructor arguments and items as reference src/utils/datasets.py:54-263
(`get_dataset`, `SeqSampler`, `BaseDataset`, `Replica`, `ScanNet`, `TUM_RGBD`; `readEXR_onlydepth` is named but
not built: no OpenEXR here and no shipped configuration uses it), with PIL + numpy + torch instead of OpenCV (cv2 is not
available in this image).  Host code: the callers either side of the hot path need it to
feed real sequences to the tracking / mapping loops; nothing here touches the GPU kernels.

Item: `(index, color [H',W',3] float RGB in [0,1], depth [H',W'] float32 metres * scale, c2w [4,4] float32)` after the
optional resize to `crop_size` and the `crop_edge` crop (datasets.py:88-114).

What is bit-for-bit the reference's arithmetic: 16-bit PNG depth / png_depth_scale * scale, the crop_size resize (the same
F.interpolate calls), the edge crop, trajectory parsing, the OpenGL flip `c2w[:3, 1:3] *= -1`, TUM's timestamp association
(max_dt = 0.08 s), frame-rate thinning and re-basing on the first pose, ScanNet's numeric file order.
What OpenCV did and is restated here - parity unpinned, no cv2 to compare with: JPEG decoding (PIL's libjpeg; may differ
from cv2's by +-1 of 255), `cv2.resize` of the colour image to the depth image's size (bilinear with pixel centres,
here F.interpolate(align_corners=False)), `cv2.undistort` (TUM only: the inverse-map bilinear resampling below, in float
instead of cv2's 5-bit fixed-point weights).
"""
import glob
import os

import numpy as np
import torch
import torch.nn.functional as F
from torch.utils.data import Dataset


def as_intrinsics_matrix(intrinsics):
    """[fx, fy, cx, cy] -> 3x3 K (reference src/common.py:27-38)."""
    K = np.eye(3)
    K[0, 0], K[1, 1], K[0, 2], K[1, 2] = intrinsics
    return K


def _imread_color(path):
    """RGB uint8 [H,W,3] (cv2.imread + cvtColor(BGR2RGB) of datasets.py:91,100)."""
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im.convert("RGB"))


def _imread_depth(path):
    """The PNG's own sample type (cv2.IMREAD_UNCHANGED of datasets.py:92): uint16 for the depth maps of all three datasets."""
    from PIL import Image
    with Image.open(path) as im:
        a = np.asarray(im)
    if a.ndim != 2:
        raise ValueError(f"{path}: expected a single-channel depth image, got shape {a.shape}")
    return a


def undistort(color, K, dist):
    """cv2.undistort(color, K, dist) restated: for every pixel of the OUTPUT (undistorted, same K) image, where the
    distorted input image shows the same ray, sampled bilinearly, 0 outside.  dist = (k1, k2, p1, p2[, k3])."""
    d = np.zeros(5, dtype=np.float64)
    d[:min(5, len(dist))] = np.asarray(dist, dtype=np.float64).reshape(-1)[:5]
    k1, k2, p1, p2, k3 = d
    H, W = color.shape[:2]
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    u, v = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
    x, y = (u - cx) / fx, (v - cy) / fy
    r2 = x * x + y * y
    radial = 1 + k1 * r2 + k2 * r2 * r2 + k3 * r2 * r2 * r2
    xd = x * radial + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
    yd = y * radial + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
    mu, mv = fx * xd + cx, fy * yd + cy
    grid = torch.from_numpy(np.stack([2 * mu / (W - 1) - 1, 2 * mv / (H - 1) - 1], -1)).float()[None]
    img = torch.from_numpy(np.array(color)).permute(2, 0, 1)[None].float()
    out = F.grid_sample(img, grid, mode="bilinear", padding_mode="zeros", align_corners=True)[0].permute(1, 2, 0)
    return out.round().clamp(0, 255).to(torch.uint8).numpy() if color.dtype == np.uint8 else out.numpy()


def get_dataset(cfg, args, scale, device='cuda:0'):
    """The reader class registered for cfg['dataset'] (datasets.py:51-52)."""
    try:
        reader = dataset_dict[cfg['dataset']]
    except KeyError:
        raise KeyError(f"unknown dataset {cfg['dataset']!r}; known: {sorted(dataset_dict)}") from None
    return reader(cfg, args, scale, device=device)


class SeqSampler(torch.utils.data.Sampler):
    """Indices 0, step, 2*step, ... of a sequence of n_samples frames, plus the last frame when include_last
    (the mapper's frame reader walks the sequence with it; datasets.py:33-48, Mapper.py:102)."""

    def __init__(self, n_samples, step, include_last=True):
        self.n_samples, self.step, self.include_last = int(n_samples), int(step), bool(include_last)

    def _indices(self):
        idx = list(range(0, self.n_samples, self.step))
        if self.include_last and self.n_samples > 0 and idx[-1] != self.n_samples - 1:
            idx.append(self.n_samples - 1)
        return idx

    def __iter__(self):
        return iter(self._indices())

    def __len__(self):
        return len(self._indices())


def readEXR_onlydepth(filename):
    """datasets.py:14-31 reads the Z channel of an OpenEXR file.  No dataset the reference ships a config for stores depth
    that way (Replica, ScanNet and TUM are 16-bit PNG) and the OpenEXR bindings are not in this image: named so that
    `from src.utils.datasets import readEXR_onlydepth` resolves, raising at call time."""
    raise ImportError("readEXR_onlydepth needs the OpenEXR Python bindings, which this build does not depend on; "
                      "the PNG readers cover Replica, ScanNet and TUM RGB-D")


# ---- the per-frame pipeline, one small function per step; BaseDataset.__getitem__ strings them together ----------------
def _load_rgbd(color_path, depth_path, png_depth_scale, K_dist=None):
    """Files -> colour [Hc,Wc,3] float64 in [0,1] (RGB) and depth [Hd,Wd] float32 in metres.  Lens distortion (TUM) is
    removed from the colour image only: the depth maps are registered to the undistorted view already."""
    rgb = _imread_color(color_path)
    if K_dist is not None:
        rgb = undistort(rgb, *K_dist)
    metres = _imread_depth(depth_path).astype(np.float32) / png_depth_scale
    return torch.from_numpy(rgb / 255.), torch.from_numpy(metres)


def _match_size(color, hw):
    """Colour image brought to the depth image's size (ScanNet ships 1296x968 colour beside 640x480 depth): bilinear on
    pixel centres, which is what cv2.resize's default does."""
    if tuple(color.shape[:2]) == tuple(hw):
        return color
    chw = color.permute(2, 0, 1)[None]
    return F.interpolate(chw, tuple(hw), mode='bilinear', align_corners=False)[0].permute(1, 2, 0).contiguous()


def _resize_pair(color, depth, size):
    """cfg cam.crop_size is a RESIZE of both images to (H', W'), intrinsics being rescaled by the caller (ESLAM.update_cam):
    colour bilinear with aligned corners, depth nearest (no mixing of depths across an edge)."""
    if size is None:
        return color, depth
    c = F.interpolate(color.permute(2, 0, 1)[None], size, mode='bilinear', align_corners=True)[0]
    d = F.interpolate(depth[None, None], size, mode='nearest')[0, 0]
    return c.permute(1, 2, 0).contiguous(), d


def _trim(color, depth, edge):
    """cfg cam.crop_edge: drop `edge` pixels on every side (the rim of ScanNet / TUM colour images is invalid)."""
    if edge <= 0:
        return color, depth
    return color[edge:-edge, edge:-edge], depth[edge:-edge, edge:-edge]


def _cam_key(cam, key, default=None):
    return cam[key] if key in cam else default


class BaseDataset(Dataset):
    """What the three readers share: camera / scale bookkeeping and the frame pipeline.  A subclass provides
    `color_paths`, `depth_paths`, `poses` (list of [4,4] float32 c2w in the renderer's convention) and `n_img`."""

    def __init__(self, cfg, args, scale, device='cuda:0'):
        super().__init__()
        cam = cfg['cam']
        self.name, self.device, self.scale = cfg['dataset'], device, scale
        for key in ('H', 'W', 'fx', 'fy', 'cx', 'cy', 'png_depth_scale', 'crop_edge'):
            setattr(self, key, cam[key])
        dist = _cam_key(cam, 'distortion')
        self.distortion = None if dist is None else np.array(dist)
        self.crop_size = _cam_key(cam, 'crop_size')
        override = getattr(args, 'input_folder', None)
        self.input_folder = cfg['data']['input_folder'] if override is None else override

    def __len__(self):
        return self.n_img

    def __getitem__(self, index):
        K_dist = None
        if self.distortion is not None:
            K_dist = (as_intrinsics_matrix([self.fx, self.fy, self.cx, self.cy]), self.distortion)
        color, depth = _load_rgbd(self.color_paths[index], self.depth_paths[index], self.png_depth_scale, K_dist)
        color = _match_size(color, depth.shape)
        color, depth = _resize_pair(color, depth * self.scale, self.crop_size)
        color, depth = _trim(color, depth, self.crop_edge)
        # the stored pose itself is scaled, in place, as the reference does it: an item read twice is scaled twice when
        # scale != 1 (every config of the reference has scale 1)
        c2w = self.poses[index]
        c2w[:3, 3] *= self.scale
        return index, color, depth, c2w


def _to_renderer_frame(mats):
    """[n,4,4] float64 dataset poses (camera looks along +z, y down) -> list of float32 tensors in the renderer's OpenGL
    convention (looks along -z, y up): the y and z columns of the rotation change sign, the translation does not."""
    out = np.array(mats, dtype=np.float64, copy=True).reshape(-1, 4, 4)
    out[:, :3, 1:3] *= -1.0
    return [torch.from_numpy(m).float() for m in out]


def _frame_number(path):
    return int(os.path.splitext(os.path.basename(path))[0])


class Replica(BaseDataset):
    """<folder>/results/frameNNNNNN.jpg + depthNNNNNN.png, <folder>/traj.txt: one row-major 4x4 c2w per line."""

    def __init__(self, cfg, args, scale, device='cuda:0'):
        super().__init__(cfg, args, scale, device)
        frames = os.path.join(self.input_folder, 'results')
        self.color_paths = sorted(glob.glob(os.path.join(frames, 'frame*.jpg')))
        self.depth_paths = sorted(glob.glob(os.path.join(frames, 'depth*.png')))
        self.n_img = len(self.color_paths)
        self.load_poses(os.path.join(self.input_folder, 'traj.txt'))

    def load_poses(self, path):
        # the file may hold more lines than there are frames on disk: only the first n_img are used
        table = np.loadtxt(path, dtype=np.float64, ndmin=2, max_rows=self.n_img if self.n_img else None)
        self.poses = _to_renderer_frame(table[:self.n_img].reshape(-1, 4, 4))


class ScanNet(BaseDataset):
    """<folder>/color/<k>.jpg, depth/<k>.png, pose/<k>.txt (4 lines of 4 numbers); k is a plain integer, so the files are
    ordered numerically, not lexicographically."""

    def __init__(self, cfg, args, scale, device='cuda:0'):
        super().__init__(cfg, args, scale, device)
        listing = lambda sub, ext: sorted(glob.glob(os.path.join(self.input_folder, sub, '*' + ext)), key=_frame_number)
        self.color_paths, self.depth_paths = listing('color', '.jpg'), listing('depth', '.png')
        self.load_poses(os.path.join(self.input_folder, 'pose'))
        self.n_img = len(self.color_paths)

    def load_poses(self, path):
        files = sorted(glob.glob(os.path.join(path, '*.txt')), key=_frame_number)
        self.poses = _to_renderer_frame([np.loadtxt(f, dtype=np.float64) for f in files]) if files else []


class TUM_RGBD(BaseDataset):
    """TUM RGB-D layout: rgb.txt / depth.txt (timestamp, file) and groundtruth.txt | pose.txt (timestamp, tx ty tz qx qy qz qw,
    one header line).  Reference: datasets.py:169-257."""

    MAX_DT = 0.08           # s: largest timestamp gap for a colour / depth / pose triple (datasets.py:183)
    FRAME_RATE = 32         # frames closer than 1/32 s to the last kept one are dropped (datasets.py:174, :222-227)

    def __init__(self, cfg, args, scale, device='cuda:0'):
        super(TUM_RGBD, self).__init__(cfg, args, scale, device)
        self.color_paths, self.depth_paths, self.poses = self.loadtum(self.input_folder, frame_rate=self.FRAME_RATE)
        self.n_img = len(self.color_paths)

    @staticmethod
    def parse_list(filepath, skiprows=0):
        return np.loadtxt(filepath, delimiter=' ', dtype=str, skiprows=skiprows)

    @classmethod
    def associate_frames(cls, tstamp_image, tstamp_depth, tstamp_pose, max_dt=None):
        """For every colour timestamp the nearest depth (and pose) timestamp; kept when all gaps are below max_dt.
        Returns (i, j) or (i, j, k) index tuples in colour order, as datasets.py:183-200 does."""
        max_dt = cls.MAX_DT if max_dt is None else max_dt
        t = np.asarray(tstamp_image, dtype=np.float64)[:, None]
        j = np.abs(np.asarray(tstamp_depth, dtype=np.float64)[None] - t).argmin(1)
        keep = np.abs(np.asarray(tstamp_depth)[j] - t[:, 0]) < max_dt
        cols = [np.arange(t.shape[0]), j]
        if tstamp_pose is not None:
            k = np.abs(np.asarray(tstamp_pose, dtype=np.float64)[None] - t).argmin(1)
            keep &= np.abs(np.asarray(tstamp_pose)[k] - t[:, 0]) < max_dt
            cols.append(k)
        return [tuple(int(c[n]) for c in cols) for n in np.flatnonzero(keep)]

    @staticmethod
    def pose_matrices(vecs):
        """[n,7] rows (tx, ty, tz, qx, qy, qz, qw: scalar-last, the TUM file order) -> [n,4,4] camera-to-world matrices."""
        from scipy.spatial.transform import Rotation
        vecs = np.asarray(vecs, dtype=np.float64).reshape(-1, 7)
        mats = np.tile(np.eye(4), (vecs.shape[0], 1, 1))
        mats[:, :3, :3] = Rotation.from_quat(vecs[:, 3:]).as_matrix()
        mats[:, :3, 3] = vecs[:, :3]
        return mats

    @classmethod
    def pose_matrix_from_quaternion(cls, pvec):
        """One row of the trajectory file -> 4x4 matrix (the name the reference's callers use, datasets.py:249)."""
        return cls.pose_matrices(pvec)[0]

    def loadtum(self, datapath, frame_rate=-1):
        pose_file = next((f for f in ('groundtruth.txt', 'pose.txt') if os.path.isfile(os.path.join(datapath, f))), None)
        if pose_file is None:
            raise FileNotFoundError(f"{datapath}: neither groundtruth.txt nor pose.txt")
        rgb = self.parse_list(os.path.join(datapath, 'rgb.txt'))
        dep = self.parse_list(os.path.join(datapath, 'depth.txt'))
        gt = self.parse_list(os.path.join(datapath, pose_file), skiprows=1)
        t_rgb = rgb[:, 0].astype(np.float64)
        triples = self.associate_frames(t_rgb, dep[:, 0].astype(np.float64), gt[:, 0].astype(np.float64))
        # thin to the frame rate: keep a triple when its colour image is more than 1/frame_rate after the last kept one
        kept = triples[:1]
        for tr in triples[1:]:
            if t_rgb[tr[0]] - t_rgb[kept[-1][0]] > 1.0 / frame_rate:
                kept.append(tr)
        if not kept:
            return [], [], []
        images = [os.path.join(datapath, rgb[i, 1]) for i, _, _ in kept]
        depths = [os.path.join(datapath, dep[j, 1]) for _, j, _ in kept]
        # trajectory re-based on the first kept frame (whose pose becomes exactly the identity), then the OpenGL flip
        world = self.pose_matrices(gt[[k for _, _, k in kept], 1:].astype(np.float64))
        rebased = np.stack([np.eye(4)] + [m for m in np.linalg.inv(world[0]) @ world[1:]])
        return images, depths, _to_renderer_frame(rebased)


dataset_dict = {
    "replica": Replica,
    "scannet": ScanNet,
    "tumrgbd": TUM_RGBD
}
