"""CPU simulation of the ray ORDER used for bundling in the scatter: for each candidate order of the bench rays
(room0, 4096x64), bundles of 32 consecutive rays x 12 planes -> number of distinct cells (= flush count before the
column carry) and the share of (bundle, plane) boxes too large for the in-LDS counting sort (> 8192 padded bins)."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _cpu_samples import bench_samples

# python tools/sim_order.py [scene rays n_strat n_imp]
R, ns, ni = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (4096, 56, 8)
S = ns + ni
sc, idx, ro, rd, z, pn = bench_samples(R, ns, ni, scene=sys.argv[1] if len(sys.argv) > 1 else "room0")

def spread(v, bits):
    out = np.zeros_like(v, dtype=np.uint64)
    for b in range(bits):
        out |= ((v >> b) & 1).astype(np.uint64) << (3 * b)
    return out

def spread2(v, bits):
    out = np.zeros_like(v, dtype=np.uint64)
    for b in range(bits):
        out |= ((v >> b) & 1).astype(np.uint64) << (2 * b)
    return out

p1 = ro + rd / np.linalg.norm(rd, axis=1, keepdims=True)

def morton3(bits):
    lo, hi = p1.min(0), p1.max(0)
    q = np.clip(((p1 - lo) / np.maximum(hi - lo, 1e-6) * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)
    return spread(q[:, 0], bits) | (spread(q[:, 1], bits) << 1) | (spread(q[:, 2], bits) << 2)

def pca2(bits, hilbert=False):
    """2-D order in the two dominant principal axes of the 1-m points (a camera's rays are a 2-D patch of a sphere)."""
    c = p1 - p1.mean(0)
    _, _, vt = np.linalg.svd(c, full_matrices=False)
    uv = c @ vt[:2].T
    lo, hi = uv.min(0), uv.max(0)
    q = np.clip(((uv - lo) / np.maximum(hi - lo, 1e-6) * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)
    if not hilbert:
        return spread2(q[:, 0], bits) | (spread2(q[:, 1], bits) << 1)
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
    return d

orders = {"given (random pixels)": np.arange(R), "3-D Morton 5 bits/axis (kernel)": np.argsort(morton3(5), kind='stable'),
          "3-D Morton 10 bits/axis": np.argsort(morton3(10), kind='stable'),
          "2-D Morton (PCA plane) 8 bits": np.argsort(pca2(8), kind='stable'),
          "2-D Hilbert (PCA plane) 8 bits": np.argsort(pca2(8, True), kind='stable')}
# one order PER PLANE ORIENTATION: rays sorted by the azimuth of their direction projected into that plane - rays of one
# bundle then lie on top of each other in the projection (whatever their angle out of the plane), which is what shares cells
def azimuth_orders():
    out = []
    for (a, b) in [(0, 1), (0, 2), (1, 2)]:
        out.append(np.argsort(np.arctan2(rd[:, b], rd[:, a]), kind='stable'))
    return out
orders["per-orientation azimuth (3 orders)"] = azimuth_orders()
B = 2048 // S
print(f"{'order':34s} {'distinct cells':>14s} {'carried flushes':>16s} {'boxes > 8192 bins':>18s}")
for name, order_any in orders.items():
    cells_tot = flush_tot = 0
    big = nbox = 0
    for d in range(2):
        for lvl in range(2):
            for o, (a, b) in enumerate([(0, 1), (0, 2), (1, 2)]):
                order = order_any[o] if isinstance(order_any, list) else order_any
                shp = sc.plane_shapes[3 * d + o][lvl]
                h, w = shp[2], shp[3]
                x0 = np.floor(np.clip((pn[..., a] + 1) / 2 * (w - 1), 0, w - 1)).astype(np.int64)[order]
                y0 = np.floor(np.clip((pn[..., b] + 1) / 2 * (h - 1), 0, h - 1)).astype(np.int64)[order]
                for i in range(0, R, B):
                    xs, ys = x0[i:i + B].ravel(), y0[i:i + B].ravel()
                    ex, ey = xs.max() - xs.min(), ys.max() - ys.min()
                    swap = ey > ex
                    m, M = (ys, xs) if swap else (xs, ys)
                    mext, Mext = m.max() - m.min() + 2, M.max() - M.min() + 1
                    nbox += 1
                    big += mext * Mext > 8192
                    cell = np.unique(M * 100000 + m)
                    cells_tot += len(cell)
                    # with the column carry a cell adjacent (along the minor axis) to its predecessor costs half a flush
                    adj = np.concatenate([[False], np.diff(cell) == 1])
                    flush_tot += len(cell) - 0.5 * adj.sum()
    print(f"{name:34s} {cells_tot:14d} {flush_tot:16.0f} {100*big/nbox:17.1f}%")
