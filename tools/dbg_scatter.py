"""Eager steps of one BASELINE.json workload for the rocprofv3 passes (kernel trace, PMC) and the ESLAM_SC_* scatter switches:
    python tools/dbg_scatter.py [scene rays n_strat n_imp zero_frac [lowp | camsN]]        (default: the bench workload, room0 4096 x 64;
    camsN: the batch of an N-camera keyframe window instead of one camera's rays)
Also launches one calibration read of known size (a float32 sum over 256 MiB) so that collect_traffic.py can turn the L2
request counters of the same pass into bytes."""
import ctypes, os, sys, time, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from myslam_amd import harness, _hip
dev = torch.device('cuda:0')
a = sys.argv[1:]
lowp = len(a) > 5 and a[5] == 'lowp'
cams = int(a[5][4:]) if len(a) > 5 and a[5].startswith('cams') else 1
wl = harness.make_workload(a[0] if a else 'room0', int(a[1]) if a else 4096, int(a[2]) if a else 56, int(a[3]) if a else 8,
                           device=dev, zero_frac=float(a[4]) if len(a) > 4 else 0.0, cams=cams,
                           focal_scale=float(os.environ.get('DBG_FOCAL_SCALE', '1')))
if cams > 1: print(f"{cams} cameras, {wl.R} rays kept of the batch")
step = wl.step
if lowp:
    from myslam_amd import lowp as lp, ops
    half = lp.HalfPlanes(wl.planes)
    def step():
        with ops.mixed_precision(half):
            return wl.step()
lib = _hip.lib()
buf = (ctypes.c_float * 12)()
cal = torch.ones(64 * 1024 * 1024, device=dev)            # 256 MiB: larger than the L2s, read exactly once by the reduction
for _ in range(3): step()
torch.cuda.synchronize()
ts = []
for _ in range(10):
    lib.eslam_profile_enable(1); step(); torch.cuda.synchronize(); lib.eslam_profile_read(buf); ts.append(buf[4])
lib.eslam_profile_enable(0)
for _ in range(3): float(cal.sum())
env = {k: v for k, v in os.environ.items() if k.startswith('ESLAM_SC')}
print(env, 'scatter ms median %.4f' % sorted(ts)[5], 'all kernels ms:', [round(x, 4) for x in buf])
