"""RGB-D sequence readers: same classes, constructor arguments and items as reference src/utils/datasets.py:54-263
(`get_dataset`, `BaseDataset`, `Replica`, `ScanNet`, `TUM_RGBD`), with PIL + numpy + torch instead of OpenCV (cv2 is not
available in this image).  Host code: the callers either side of the hot path need it to
feed real sequences to the tracking / mapping loops; nothing here touches the GPU kernels.

Item: `(index, color [H',W',3] float RGB in [0,1], depth [H',W'] float32 metres * scale, c2w [4,4] float32)` after the
optional resize to `crop_size` and the `crop_edge` crop, exactly as datasets.py:88-114.

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
    return dataset_dict[cfg['dataset']](cfg, args, scale, device=device)


class BaseDataset(Dataset):
    def __init__(self, cfg, args, scale, device='cuda:0'):
        super(BaseDataset, self).__init__()
        self.name = cfg['dataset']
        self.device = device
        self.scale = scale
        self.png_depth_scale = cfg['cam']['png_depth_scale']
        self.H, self.W, self.fx, self.fy, self.cx, self.cy = (cfg['cam'][k] for k in ('H', 'W', 'fx', 'fy', 'cx', 'cy'))
        self.distortion = np.array(cfg['cam']['distortion']) if 'distortion' in cfg['cam'] else None
        self.crop_size = cfg['cam']['crop_size'] if 'crop_size' in cfg['cam'] else None
        self.input_folder = cfg['data']['input_folder'] if getattr(args, 'input_folder', None) is None else args.input_folder
        self.crop_edge = cfg['cam']['crop_edge']

    def __len__(self):
        return self.n_img

    def __getitem__(self, index):
        color_data = _imread_color(self.color_paths[index])
        depth_data = _imread_depth(self.depth_paths[index])
        if self.distortion is not None:
            # undistortion is only applied on the colour image, not on depth (datasets.py:93-96)
            color_data = undistort(color_data, as_intrinsics_matrix([self.fx, self.fy, self.cx, self.cy]), self.distortion)
        color_data = color_data / 255.                                   # float64, as in the reference
        depth_data = depth_data.astype(np.float32) / self.png_depth_scale
        H, W = depth_data.shape
        color_data = torch.from_numpy(color_data)
        if tuple(color_data.shape[:2]) != (H, W):                        # cv2.resize(color, (W, H)), datasets.py:102
            color_data = F.interpolate(color_data.permute(2, 0, 1)[None], (H, W), mode='bilinear',
                                       align_corners=False)[0].permute(1, 2, 0).contiguous()
        depth_data = torch.from_numpy(depth_data) * self.scale

        if self.crop_size is not None:
            # follow the pre-processing step in lietorch, actually is resize (datasets.py:106-113)
            color_data = color_data.permute(2, 0, 1)
            color_data = F.interpolate(color_data[None], self.crop_size, mode='bilinear', align_corners=True)[0]
            depth_data = F.interpolate(depth_data[None, None], self.crop_size, mode='nearest')[0, 0]
            color_data = color_data.permute(1, 2, 0).contiguous()

        edge = self.crop_edge
        if edge > 0:
            # crop image edge, there are invalid values on the edge of the colour image
            color_data = color_data[edge:-edge, edge:-edge]
            depth_data = depth_data[edge:-edge, edge:-edge]
        pose = self.poses[index]
        pose[:3, 3] *= self.scale          # (in place, as the reference does: reading an item twice scales twice when scale != 1)
        return index, color_data, depth_data, pose


def _flip_yz(c2w):
    """The datasets' camera looks along +z with y down; the renderer's looks along -z with y up (datasets.py:132-133)."""
    c2w[:3, 1] *= -1
    c2w[:3, 2] *= -1
    return torch.from_numpy(c2w).float()


class Replica(BaseDataset):
    def __init__(self, cfg, args, scale, device='cuda:0'):
        super(Replica, self).__init__(cfg, args, scale, device)
        self.color_paths = sorted(glob.glob(f'{self.input_folder}/results/frame*.jpg'))
        self.depth_paths = sorted(glob.glob(f'{self.input_folder}/results/depth*.png'))
        self.n_img = len(self.color_paths)
        self.load_poses(f'{self.input_folder}/traj.txt')

    def load_poses(self, path):
        self.poses = []
        with open(path, "r") as f:
            lines = f.readlines()
        for i in range(self.n_img):
            self.poses.append(_flip_yz(np.array(list(map(float, lines[i].split()))).reshape(4, 4)))


class ScanNet(BaseDataset):
    def __init__(self, cfg, args, scale, device='cuda:0'):
        super(ScanNet, self).__init__(cfg, args, scale, device)
        number = lambda x: int(os.path.basename(x)[:-4])
        self.color_paths = sorted(glob.glob(os.path.join(self.input_folder, 'color', '*.jpg')), key=number)
        self.depth_paths = sorted(glob.glob(os.path.join(self.input_folder, 'depth', '*.png')), key=number)
        self.load_poses(os.path.join(self.input_folder, 'pose'))
        self.n_img = len(self.color_paths)

    def load_poses(self, path):
        self.poses = []
        pose_paths = sorted(glob.glob(os.path.join(path, '*.txt')), key=lambda x: int(os.path.basename(x)[:-4]))
        for pose_path in pose_paths:
            with open(pose_path, "r") as f:
                ls = [list(map(float, line.split(' '))) for line in f.readlines()]
            self.poses.append(_flip_yz(np.array(ls).reshape(4, 4)))


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
    def pose_matrix_from_quaternion(pvec):
        """(tx, ty, tz, qx, qy, qz, qw) -> 4x4 camera-to-world matrix."""
        from scipy.spatial.transform import Rotation
        pose = np.eye(4)
        pose[:3, :3] = Rotation.from_quat(pvec[3:]).as_matrix()
        pose[:3, 3] = pvec[:3]
        return pose

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
        vecs = gt[:, 1:].astype(np.float64)
        first_inv = np.linalg.inv(self.pose_matrix_from_quaternion(vecs[kept[0][2]])) if kept else None
        images, depths, poses = [], [], []
        for n, (i, j, k) in enumerate(kept):
            images.append(os.path.join(datapath, rgb[i, 1]))
            depths.append(os.path.join(datapath, dep[j, 1]))
            # trajectory re-based on the first kept frame (whose pose becomes the identity), then the OpenGL flip
            c2w = np.eye(4) if n == 0 else first_inv @ self.pose_matrix_from_quaternion(vecs[k])
            poses.append(_flip_yz(c2w))
        return images, depths, poses


dataset_dict = {
    "replica": Replica,
    "scannet": ScanNet,
    "tumrgbd": TUM_RGBD
}
