"""How many atomic flushes does the bundle scatter issue under different (rays x samples) tilings of the
Morton-ordered ray list?  Counts unique cells per tile (2 x 256-B atomic instr each, what scatter_sort_kernel does)
and unique texel-row-pairs with x-carry.  Pure numpy."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _cpu_samples import bench_samples
R, ns, ni = 4096, 56, 8
S = ns + ni
sc, _idx, _ro, _rd, _z, pn = bench_samples(R, ns, ni)
class _T:      # minimal stand-ins for the two tensors the code below reads with .numpy()
    def __init__(self, a): self.a = a
    def numpy(self): return self.a
ro, rd = _T(_ro), _T(_rd)
def part(v):
    v = v.astype(np.uint32) & 0x3FF
    v = (v | (v << 16)) & 0x030000FF; v = (v | (v << 8)) & 0x0300F00F; v = (v | (v << 4)) & 0x030C30C3; v = (v | (v << 2)) & 0x09249249
    return v
dn = rd.numpy() / np.linalg.norm(rd.numpy(), axis=1, keepdims=True)
p1 = ro.numpy() + dn
q = np.clip(np.floor((p1 - p1.min(0)) / (p1.max(0) - p1.min(0) + 1e-9) * 32), 0, 31).astype(np.uint32)
order = np.argsort(part(q[:, 0]) | (part(q[:, 1]) << 1) | (part(q[:, 2]) << 2), kind='stable')
cells = []
for d in range(2):
    for lvl in range(2):
        for o, (a, b) in enumerate([(0, 1), (0, 2), (1, 2)]):
            shp = sc.plane_shapes[3 * d + o][lvl]; h, w = shp[2], shp[3]
            x0 = np.floor(np.clip((pn[..., a] + 1) / 2 * (w - 1), 0, w - 1)).astype(np.int64)
            y0 = np.floor(np.clip((pn[..., b] + 1) / 2 * (h - 1), 0, h - 1)).astype(np.int64)
            cells.append(((y0 * w + x0)[order], w))
def count(tiles):
    """tiles: list of (ray_lo, ray_hi, s_lo, s_hi).  Returns (#cell flushes, #flushes with x-carry in half-instr units)."""
    tot = 0; carry = 0.0
    for cell, w in cells:
        for (r0, r1, s0, s1) in tiles:
            u = np.unique(cell[r0:r1, s0:s1])
            tot += len(u)
            # with x-carry: a cell whose left neighbour (cell-1) is also present costs 1 column instead of 2
            adj = np.isin(u - 1, u) & ((u % w) != 0)
            carry += (len(u) + (~adj).sum()) / 2.0     # in units of full 2-column flushes
    return tot, carry
def tiling(rb, sb_list):
    t = []
    for r0 in range(0, R, rb):
        for (s0, s1) in sb_list:
            t.append((r0, min(R, r0 + rb), s0, s1))
    return t
def mixed(near_s, near_rb, far_rb):
    return tiling(near_rb, [(0, near_s)]) + tiling(far_rb, [(near_s, S)])
for name, t in [('16 rays x 64 (current)', tiling(16, [(0, S)])), ('32 x 64', tiling(32, [(0, S)])), ('64 x 64', tiling(64, [(0, S)])),
                ('near 8: 128x8 | far 18x56', mixed(8, 128, 18)), ('near 16: 64x16 | far 21x48', mixed(16, 64, 21)),
                ('near 16: 128x16(2048) | far 42x48(2016)', mixed(16, 128, 42)),
                ('near 24: 42x24 | far 25x40', mixed(24, 42, 25))]:
    n, c = count(t)
    print(f"{name:42s} WGs/plane {len(t):5d}  cell flushes {n:8d} -> {n*512/1.3e12*1e3:.3f} ms   with x-carry {c:9.0f} -> {c*512/1.3e12*1e3:.3f} ms")
