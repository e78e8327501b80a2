"""Experiment (round 1): does a 2-D Hilbert ray order (single camera) beat the kernel's order on the GPU?
Computes the permutation on the CPU and injects it in place of eslam_ray_order - as all THREE per-orientation orders of the
ABI-5 buffer ([3][R] + the fan's extent, here 0 = bundle-major grid; ESLAM_TORCH_EXT=0: the injection hooks the Python glue)."""
import ctypes, os, sys, numpy as np, torch
os.environ.setdefault('ESLAM_TORCH_EXT', '0')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from myslam_amd import harness, ops, _hip
dev = torch.device('cuda:0')
wl = harness.make_workload('room0', 4096, 56, 8, device=dev)
lib = _hip.lib()

def hilbert_perm(bits=8):
    ro, rd = wl.rays_o.cpu().numpy().astype(np.float64), wl.rays_d.cpu().numpy().astype(np.float64)
    p1 = ro + rd / np.linalg.norm(rd, axis=1, keepdims=True)
    c = p1 - p1.mean(0)
    _, _, vt = np.linalg.svd(c, full_matrices=False)
    uv = c @ vt[:2].T
    lo, hi = uv.min(0), uv.max(0)
    q = np.clip(((uv - lo) / np.maximum(hi - lo, 1e-6) * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)
    x, y = q[:, 0].copy(), q[:, 1].copy()
    d = np.zeros(len(x), dtype=np.int64)
    s = 1 << (bits - 1)
    while s > 0:
        rx = ((x & s) > 0).astype(np.int64); ry = ((y & s) > 0).astype(np.int64)
        d += s * s * ((3 * rx) ^ ry)
        flip = (ry == 0) & (rx == 1)
        x = np.where(flip, s - 1 - x, x); y = np.where(flip, s - 1 - y, y)
        sw = ry == 0
        x, y = np.where(sw, y, x), np.where(sw, x, y)
        s >>= 1
    return as_orders(torch.from_numpy(np.argsort(d, kind='stable').astype(np.int32)).to(dev))


def as_orders(p):          # the buffer eslam_ray_order fills: three orders + 4 words (the fan's extent per plane: zeros)
    return torch.cat([p, p, p, torch.zeros(4, dtype=torch.int32, device=dev)])

def measure(label):
    buf = (ctypes.c_float * 12)()
    for _ in range(5): wl.step()
    acc = {}
    for _ in range(15):
        lib.eslam_profile_enable(1); wl.step(); torch.cuda.synchronize(); lib.eslam_profile_read(buf)
        for i in (0, 2, 4): acc.setdefault(i, []).append(buf[i])
    lib.eslam_profile_enable(0)
    print(label, {lib.eslam_profile_name(i).decode(): round(sorted(v)[7] * 1e3, 1) for i, v in acc.items()}, 'us', flush=True)

measure('kernel (per-orientation azimuth)')
perm = hilbert_perm()
side = torch.cuda.Stream()
orig = ops.ray_order_async
ops.ray_order_async = lambda ro, rd, planes=None: (perm, side)
import myslam_amd.src.utils.Renderer as R
measure('2-D Hilbert (injected)')
ops.ray_order_async = lambda ro, rd, planes=None: (as_orders(torch.arange(4096, dtype=torch.int32, device=dev)), side)
measure('unsorted (identity)')
