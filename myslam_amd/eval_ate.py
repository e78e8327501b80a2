"""Absolute trajectory error of an estimated camera trajectory (the metric of
BASELINE.json configs[2]).  Same definition as the reference's evaluator (src/tools/eval_ate.py:66-100,135-246):
closed-form rigid alignment (Horn 1987, via the SVD of the cross-covariance) of the estimated camera centres onto the
ground-truth ones, then RMSE / mean / median of the residual distances.  numpy only, host side.
"""
import numpy as np


def align(model, data):
    """model, data: [3,n] estimated / ground-truth positions -> (rot [3,3], trans [3,1], residual distances [n])
    with rot @ model + trans ~ data in the least-squares sense (reflection-free)."""
    model = np.asarray(model, dtype=np.float64)
    data = np.asarray(data, dtype=np.float64)
    mc = model - model.mean(1, keepdims=True)
    dc = data - data.mean(1, keepdims=True)
    cov = mc @ dc.T                                     # sum of outer(model_i, data_i)
    U, _, Vh = np.linalg.svd(cov.T)
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vh) < 0:
        S[2, 2] = -1.0
    rot = U @ S @ Vh
    trans = data.mean(1, keepdims=True) - rot @ model.mean(1, keepdims=True)
    err = rot @ model + trans - data
    return rot, trans, np.sqrt((err * err).sum(0))


def evaluate(est_c2ws, gt_c2ws, do_align=True):
    """est_c2ws, gt_c2ws: [n,4,4] (numpy or torch) -> dict(rmse, mean, median, max) in the poses' length unit."""
    est = np.asarray([np.asarray(m)[:3, 3] for m in est_c2ws], dtype=np.float64).T
    gt = np.asarray([np.asarray(m)[:3, 3] for m in gt_c2ws], dtype=np.float64).T
    if do_align and est.shape[1] >= 3:
        _, _, e = align(est, gt)
    else:
        e = np.sqrt(((est - gt) ** 2).sum(0))
    return dict(rmse=float(np.sqrt((e * e).mean())), mean=float(e.mean()), median=float(np.median(e)), max=float(e.max()))
