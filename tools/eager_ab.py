"""Median eager (no hipGraph) step time of the bench workload: 7 repeats of 200 iterations after 30 warm-up steps.
PIN=0,1 pins the process to those cores first; ESLAM_TORCH_STREAM_WAIT=1 uses torch's Stream.wait_stream for the side-stream fork / join."""
import sys, os, time
if os.environ.get('PIN'):
    os.sched_setaffinity(0, {int(c) for c in os.environ['PIN'].split(',')})
import torch
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/myslam_amd') else os.getcwd())
from myslam_amd import harness
dev = torch.device('cuda:0')
wl = harness.make_workload('room0', 4096, 56, 8, device=dev)
for _ in range(30): wl.step()
torch.cuda.synchronize()
ts = []
for rep in range(7):
    t0 = time.perf_counter()
    for _ in range(200): wl.step()
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 200 * 1e3)
print("pin", os.environ.get("PIN"), "affinity", len(os.sched_getaffinity(0)), os.environ.get("ESLAM_TORCH_STREAM_WAIT", "0"), "eager ms/step: median %.4f min %.4f" % (sorted(ts)[3], min(ts)))
