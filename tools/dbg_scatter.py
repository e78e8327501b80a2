"""Scatter experiments on the GPU box: time + table statistics for the settings given through ESLAM_SC_* env vars."""
import ctypes, os, sys, time, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from myslam_amd import harness, _hip
dev = torch.device('cuda:0')
# optional: scene rays n_strat n_imp zero_frac   (default: the bench workload)
a = sys.argv[1:]
wl = harness.make_workload(a[0] if a else 'room0', int(a[1]) if a else 4096, int(a[2]) if a else 56, int(a[3]) if a else 8,
                           device=dev, zero_frac=float(a[4]) if len(a) > 4 else 0.0)
lib = _hip.lib()
buf = (ctypes.c_float * 12)()
for _ in range(3): wl.step()
torch.cuda.synchronize()
ts = []
for _ in range(10):
    lib.eslam_profile_enable(1); wl.step(); torch.cuda.synchronize(); lib.eslam_profile_read(buf); ts.append(buf[4])
lib.eslam_profile_enable(0)
env = {k: v for k, v in os.environ.items() if k.startswith('ESLAM_SC')}
print(env, 'scatter ms median %.4f' % sorted(ts)[5], 'all kernels ms:', [round(x, 4) for x in buf])
