"""CPU simulation of the plane-gradient scatter: how many atomic flushes different merge strategies need
on the bench workload (room0, 4096x64).  Pure numpy; no GPU."""
import sys, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.abspath(__file__)))
from _cpu_samples import bench_samples

R, ns, ni = 4096, 56, 8
S = ns + ni
sc, _idx, _ro, _rd, _z, pn = bench_samples(R, ns, ni)
class _T:
    def __init__(self, a): self.a = a
    def numpy(self): return self.a
idx, rd = _T(_idx), _T(_rd)
names = ['geo-coarse', 'geo-fine', 'col-coarse', 'col-fine']
tot = {}
for d in range(2):
    for lvl in range(2):
        for o, (a, b) in enumerate([(0, 1), (0, 2), (1, 2)]):
            shp = sc.plane_shapes[3 * d + o][lvl]
            h, w = shp[2], shp[3]
            ix = np.clip((pn[..., a] + 1) / 2 * (w - 1), 0, w - 1)
            iy = np.clip((pn[..., b] + 1) / 2 * (h - 1), 0, h - 1)
            x0 = np.floor(ix).astype(np.int64); y0 = np.floor(iy).astype(np.int64)
            cell = y0 * w + x0                                   # [R,S]
            # strategy A: consecutive-run merge per ray (what scatter_kernel v1 does): flush = #runs, 2 instr each
            runs = 1 + (cell[:, 1:] != cell[:, :-1]).sum(1)
            # strategy B: unique cells per ray (ideal per-ray cache)
            uniq_cells = np.array([len(np.unique(c)) for c in cell])
            # strategy C: unique texel ROWS pairs per ray: unique (y, x0) pairs over both rows, as 256-B instr
            y1 = np.minimum(y0 + 1, h - 1)
            rows = np.stack([y0 * w + x0, y1 * w + x0], -1).reshape(R, -1)
            uniq_rowseg = np.array([len(np.unique(c)) for c in rows])
            # strategy D: unique texels per ray (128-B granules)
            x1 = np.minimum(x0 + 1, w - 1)
            tex = np.stack([y0 * w + x0, y0 * w + x1, y1 * w + x0, y1 * w + x1], -1).reshape(R, -1)
            uniq_tex = np.array([len(np.unique(c)) for c in tex])
            # strategy E: unique texels per group of 16 rays (consecutive rays, unsorted) and sorted by pixel
            key = names[2 * d + lvl]
            t = tot.setdefault(key, dict(runs=0, cells=0, rowseg=0, tex=0, tex16=0, tex16s=0, texall=0))
            t['runs'] += runs.sum(); t['cells'] += uniq_cells.sum(); t['rowseg'] += uniq_rowseg.sum(); t['tex'] += uniq_tex.sum()
            t['tex16'] += sum(len(np.unique(tex[i:i + 16])) for i in range(0, R, 16))
            order = np.argsort(idx.numpy() // sc.W // 40 * 1000 + idx.numpy() % sc.W // 40)   # 40x40 pixel tiles
            texs = tex[order]
            t['tex16s'] += sum(len(np.unique(texs[i:i + 16])) for i in range(0, R, 16))
            t['texall'] += len(np.unique(tex))
print(f"samples {R*S}, unmerged atomic instr (2 per sample per plane) = {R*S*2*12}")
print(f"{'plane class':12s} {'v1 runs x2':>12s} {'cells x2':>12s} {'rowsegs':>12s} {'texels/2':>12s} {'tex16/2':>12s} {'tex16sorted/2':>14s} {'all/2':>10s}")
s = np.zeros(7)
for k, t in tot.items():
    row = [t['runs'] * 2, t['cells'] * 2, t['rowseg'], t['tex'] / 2, t['tex16'] / 2, t['tex16s'] / 2, t['texall'] / 2]
    s += row
    print(f"{k:12s} " + " ".join(f"{v:12.0f}" for v in row))
print(f"{'total instr':12s} " + " ".join(f"{v:12.0f}" for v in s))
print("x256 B -> GB:", " ".join(f"{v*256/1e9:12.3f}" for v in s))
print("time @1.3TB/s (ms):", " ".join(f"{v*256/1.3e12*1e3:10.3f}" for v in s))

# ---- bundle statistics with a direction-Morton ray order (what scatter v2 would use) ----
def part1by2(v):
    v = v.astype(np.uint32) & 0x3FF
    v = (v | (v << 16)) & 0x030000FF
    v = (v | (v << 8)) & 0x0300F00F
    v = (v | (v << 4)) & 0x030C30C3
    v = (v | (v << 2)) & 0x09249249
    return v

dn = rd.numpy() / np.linalg.norm(rd.numpy(), axis=1, keepdims=True)
q = np.clip(np.floor((dn + 1) * 32), 0, 63).astype(np.uint32)
key = part1by2(q[:, 0]) | (part1by2(q[:, 1]) << 1) | (part1by2(q[:, 2]) << 2)
order = np.argsort(key, kind='stable')
for B in (16, 32, 64):
    tot_i, mx = 0, {}
    for d in range(2):
        for lvl in range(2):
            for o, (a, b) in enumerate([(0, 1), (0, 2), (1, 2)]):
                shp = sc.plane_shapes[3 * d + o][lvl]
                h, w = shp[2], shp[3]
                ix = np.clip((pn[..., a] + 1) / 2 * (w - 1), 0, w - 1)
                iy = np.clip((pn[..., b] + 1) / 2 * (h - 1), 0, h - 1)
                x0 = np.floor(ix).astype(np.int64); y0 = np.floor(iy).astype(np.int64)
                x1 = np.minimum(x0 + 1, w - 1); y1 = np.minimum(y0 + 1, h - 1)
                tex = np.stack([y0 * w + x0, y0 * w + x1, y1 * w + x0, y1 * w + x1], -1).reshape(R, -1)[order]
                u = [len(np.unique(tex[i:i + B])) for i in range(0, R, B)]
                tot_i += sum(u) / 2
                k = names[2 * d + lvl]
                mx[k] = max(mx.get(k, 0), max(u))
    print(f"bundle {B}: global atomic instr {tot_i:.0f} -> {tot_i*256/1.3e12*1e3:.3f} ms @1.3TB/s; max unique texels per (bundle,plane): {mx}")
