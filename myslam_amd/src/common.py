"""Mirror of the hot-path helpers of reference src/common.py (get_samples and friends, get_rays,
normalize_3d_coordinate, sample_pdf is inside the importance-sampling kernel).  Pose helpers
(matrix_to_cam_pose / cam_pose_to_matrix, common.py:155-181) are restated without pytorch3d.
"""
import numpy as np
import torch

from .. import ops


def get_samples(H0, H1, W0, W1, n, H, W, fx, fy, cx, cy, c2ws, depths, colors, device):
    """common.py:141-153.  Returns rays_o [b*n,3], rays_d [b*n,3], depth [b*n], color [b*n,3]; differentiable in
    c2ws.  One torch.randint draw for all images, as common.py:108."""
    b = c2ws.shape[0]
    indices = torch.randint((H1 - H0) * (W1 - W0), (n * b,), device=c2ws.device)
    return get_samples_at(indices, H0, H1, W0, W1, n, H, W, fx, fy, cx, cy, c2ws, depths, colors)


def get_samples_at(indices, H0, H1, W0, W1, n, H, W, fx, fy, cx, cy, c2ws, depths, colors):
    """get_samples with the pixel draw supplied by the caller (tests, ray-sharded data parallelism)."""
    geom = (int(H0), int(H1), int(W0), int(W1), int(n), int(H), int(W), float(fx), float(fy), float(cx), float(cy))
    return ops.SampleRaysFn.apply(c2ws, indices, depths, colors, geom)


def get_rays(H, W, fx, fy, cx, cy, c2w, device):
    """common.py:183-201.  Returns rays_o, rays_d of shape [H, W, 3]."""
    if isinstance(c2w, np.ndarray):
        c2w = torch.from_numpy(c2w)
    c2w = c2w.to(device=device, dtype=torch.float32)
    ro, rd = ops.image_rays(int(H), int(W), float(fx), float(fy), float(cx), float(cy), c2w)
    return ro.reshape(H, W, 3), rd.reshape(H, W, 3)


def normalize_3d_coordinate(p, bound):
    """common.py:204-218 (kept for callers such as Mesher; the kernels normalise internally)."""
    p = p.reshape(-1, 3)
    bound = bound.to(p.device)
    return torch.stack([((p[:, k] - bound[k, 0]) / (bound[k, 1] - bound[k, 0])) * 2 - 1.0 for k in range(3)], -1)


def random_select(l, k):
    """common.py:80-85."""
    return list(np.random.permutation(np.array(range(l)))[:min(l, k)])


def quaternion_to_matrix(q):
    """Real-first unit quaternion -> rotation matrix (what pytorch3d.transforms.quaternion_to_matrix computes)."""
    r, i, j, k = torch.unbind(q, -1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack((1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                     two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                     two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1)
    return o.reshape(q.shape[:-1] + (3, 3))


def matrix_to_quaternion(m):
    """Rotation matrix -> real-first quaternion with non-negative real part convention of the largest component
    (Shepperd's method, the algorithm pytorch3d 0.7.1 uses)."""
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = torch.unbind(m.reshape(m.shape[:-2] + (9,)), -1)

    def sp(x):
        return torch.sqrt(torch.clamp(x, min=0))

    q_abs = sp(torch.stack([1.0 + m00 + m11 + m22, 1.0 + m00 - m11 - m22, 1.0 - m00 + m11 - m22,
                            1.0 - m00 - m11 + m22], -1))
    cand = torch.stack([
        torch.stack([q_abs[..., 0] ** 2, m21 - m12, m02 - m20, m10 - m01], -1),
        torch.stack([m21 - m12, q_abs[..., 1] ** 2, m10 + m01, m02 + m20], -1),
        torch.stack([m02 - m20, m10 + m01, q_abs[..., 2] ** 2, m12 + m21], -1),
        torch.stack([m10 - m01, m20 + m02, m21 + m12, q_abs[..., 3] ** 2], -1)], -2)
    cand = cand / (2.0 * q_abs[..., None].clamp(min=0.1))
    idx = q_abs.argmax(-1)
    return torch.gather(cand, -2, idx[..., None, None].expand(idx.shape + (1, 4))).squeeze(-2)


def matrix_to_cam_pose(batch_matrices, RT=True):
    """common.py:155-167."""
    q = matrix_to_quaternion(batch_matrices[:, :3, :3])
    t = batch_matrices[:, :3, 3]
    return torch.cat([q, t], -1) if RT else torch.cat([t, q], -1)


def cam_pose_to_matrix(batch_poses):
    """common.py:169-181.  Poses on the GPU go through one kernel each way (ops.PoseToMatrixFn); host tensors (the
    reference converts keyframe poses on the CPU, too) through the same formula in tensor ops."""
    if batch_poses.is_cuda and batch_poses.dtype == torch.float32 and batch_poses.dim() == 2 and batch_poses.shape[1] == 7:
        return ops.PoseToMatrixFn.apply(batch_poses)
    c2w = torch.eye(4, device=batch_poses.device, dtype=batch_poses.dtype).unsqueeze(0).repeat(batch_poses.shape[0], 1, 1)
    c2w[:, :3, :3] = quaternion_to_matrix(batch_poses[:, :4])
    c2w[:, :3, 3] = batch_poses[:, 4:]
    return c2w
