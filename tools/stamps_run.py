"""Three iterations of the bench workload: run with a -DBWD_STAMPS=1 or -DSC_STAMPS=1 build of the library (make variant,
ESLAM_HIP_LIB=...) to make the decoder backward / the scatter print their per-phase s_memtime cycle counts."""
import sys, os, torch
sys.path.insert(0, os.getcwd())
from myslam_amd import harness
wl = harness.make_workload('room0', 4096, 56, 8, device=torch.device('cuda:0'))
for _ in range(3): wl.step()
torch.cuda.synchronize()
