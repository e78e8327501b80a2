import sys, os, torch
sys.path.insert(0, os.getcwd())
from myslam_amd import harness
wl = harness.make_workload('room0', 4096, 56, 8, device=torch.device('cuda:0'))
for _ in range(3): wl.step()
torch.cuda.synchronize()
