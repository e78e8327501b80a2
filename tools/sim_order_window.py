"""CPU simulation of ray orders for a keyframe-WINDOW batch (rays from several cameras: src/Mapper.py:308-319), the case
tools/sim_order.py does not cover: 10 cameras on a ring (as harness.make_workload(cams=10)), 400 pixels each, 32+8 samples.
Bundles of 2048 // S consecutive rays x 12 planes -> distinct cells per bundle (= cell flushes before the column carry)."""
import math, os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from myslam_amd import scene as scn, synth

cams, per_cam, ns, ni, trunc = 10, 400, 32, 8, 0.06
S = ns + ni
sc = scn.make_scene("room0")
b = sc.bound.numpy().astype(np.float64)
half = 0.15 * (b[:, 1] - b[:, 0])
depth_img = synth.depth_image(sc.H, sc.W, 10)
ro, rd, gd, cam = [], [], [], []
idx_all = synth.hash_randint(sc.H * sc.W, (cams * per_cam,), 50_000)
for k in range(cams):
    a = 2.0 * math.pi * k / cams
    Rm = np.array([[math.cos(a), 0.0, math.sin(a)], [0.0, 1.0, 0.0], [-math.sin(a), 0.0, math.cos(a)]])
    t = b.mean(1) + np.array([math.sin(a), 0.0, math.cos(a)]) * half
    idx = idx_all[k * per_cam:(k + 1) * per_cam]
    u, v = (idx % sc.W).astype(np.float64), (idx // sc.W).astype(np.float64)
    dirs = np.stack([(u - sc.cx) / sc.fx, -(v - sc.cy) / sc.fy, -np.ones_like(u)], -1)
    rd.append(dirs @ Rm.T); ro.append(np.broadcast_to(t, dirs.shape)); gd.append(depth_img.reshape(-1)[idx].astype(np.float64))
    cam.append(np.full(per_cam, k))
ro, rd, gd, cam = np.concatenate(ro), np.concatenate(rd), np.concatenate(gd), np.concatenate(cam)
R = len(gd)
z_free = 1.2 * gd[:, None] * np.linspace(0.0, 1.0, ns)[None]
z_surf = gd[:, None] - 1.5 * trunc + 3 * trunc * np.linspace(0.0, 1.0, ni)[None]
z = np.sort(np.concatenate([z_free, z_surf], 1), 1)
pts = ro[:, None, :] + rd[:, None, :] * z[..., None]
inside = ((pts >= b[:, 0]) & (pts <= b[:, 1])).all(-1).all(-1)       # (stand-in for the AABB pre-filter)
pn = np.clip((pts - b[:, 0]) / (b[:, 1] - b[:, 0]) * 2 - 1, -1, 1)

def spread(v, bits):
    out = np.zeros_like(v, dtype=np.uint64)
    for i in range(bits):
        out |= ((v >> i) & 1).astype(np.uint64) << (3 * i)
    return out
p1 = ro + rd / np.linalg.norm(rd, axis=1, keepdims=True)
lo, hi = p1.min(0), p1.max(0)
q = np.clip(((p1 - lo) / np.maximum(hi - lo, 1e-6) * 16).astype(np.int64), 0, 15)
morton = np.argsort(spread(q[:, 0], 4) | (spread(q[:, 1], 4) << 1) | (spread(q[:, 2], 4) << 2), kind='stable')
def per_camera_azimuth():
    out = []
    for (a, c) in [(0, 1), (0, 2), (1, 2)]:
        ang = np.zeros(R)
        for k in range(cams):
            m = cam == k
            mean = rd[m].sum(0)
            ang[m] = np.arctan2(mean[a] * rd[m, c] - mean[c] * rd[m, a], mean[a] * rd[m, a] + mean[c] * rd[m, c])
        out.append(np.lexsort((ang, cam)))
    return out
orders = {"given (camera-major, random pixels)": np.arange(R), "3-D Morton 4 bits/axis of o + d (kernel)": morton,
          "per camera: azimuth per orientation": per_camera_azimuth()}
B = 2048 // S
print(f"{cams} cameras x {per_cam} rays x {S} samples; bundles of {B} rays")
for name, order_any in orders.items():
    cells_tot = 0
    for d in range(2):
        for lvl in range(2):
            for o, (a, c) in enumerate([(0, 1), (0, 2), (1, 2)]):
                order = order_any[o] if isinstance(order_any, list) else order_any
                shp = sc.plane_shapes[3 * d + o][lvl]
                h, w = shp[2], shp[3]
                x0 = np.floor((pn[..., a] + 1) / 2 * (w - 1)).astype(np.int64)[order]
                y0 = np.floor((pn[..., c] + 1) / 2 * (h - 1)).astype(np.int64)[order]
                for i in range(0, R, B):
                    cells_tot += len(np.unique(y0[i:i + B].ravel() * 100000 + x0[i:i + B].ravel()))
    print(f"{name:44s} distinct cells per (bundle, plane) summed: {cells_tot}")
