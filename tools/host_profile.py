"""Where does the host time of one eager mapping iteration go?  (cProfile over 300 steps on the GPU box)
    python tools/host_profile.py [nchw]            ESLAM_TORCH_EXT=0 in the environment: the Python glue instead of the compiled one"""
import cProfile, os, pstats, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from myslam_amd import harness, ops
nchw = len(sys.argv) > 1 and sys.argv[1] == 'nchw'
wl = harness.make_workload('room0', 4096, 56, 8, device=torch.device('cuda:0'), channels_last=not nchw)
for _ in range(20): wl.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(300): wl.step()
torch.cuda.synchronize()
print(f"glue: {'compiled (eslam_torch_ext)' if ops.torch_ext() is not None else 'python (ctypes)'}; planes {'NCHW' if nchw else 'channels-last'}; "
      f"eager step {1e3 * (time.perf_counter() - t0) / 300:.3f} ms wall")
pr = cProfile.Profile(); pr.enable()
for _ in range(300): wl.step()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats('cumulative').print_stats(28)
