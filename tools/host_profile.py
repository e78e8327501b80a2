"""Where does the host time of one eager mapping iteration go?  (cProfile over 300 steps on the GPU box)"""
import cProfile, pstats, sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from myslam_amd import harness
wl = harness.make_workload('room0', 4096, 56, 8, device=torch.device('cuda:0'))
for _ in range(20): wl.step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(300): wl.step()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats('cumulative').print_stats(28)
