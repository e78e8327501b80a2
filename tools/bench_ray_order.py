"""eslam_ray_order alone (the three per-orientation counting sorts, one launch) on a few batch sizes: microseconds per call."""
import sys, time, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from myslam_amd import harness, ops, _hip
dev = torch.device('cuda:0')
for R, scene in ((4096, 'room0'), (4000, 'room0'), (5000, 'freiburg1_desk'), (2000, 'room0'), (1024, 'scene0000')):
    wl = harness.make_workload(scene, R, 56, 8, device=dev)
    ro, rd = wl.rays_o.detach(), wl.rays_d.detach()
    perm = torch.empty(_hip.ray_order_words(ro.shape[0]), dtype=torch.int32, device=dev)
    lib = _hip.lib()
    st = _hip.stream_handle(dev)
    for _ in range(5): lib.eslam_ray_order(_hip.ptr(ro), _hip.ptr(rd), ro.shape[0], _hip.ptr(perm), st)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): lib.eslam_ray_order(_hip.ptr(ro), _hip.ptr(rd), ro.shape[0], _hip.ptr(perm), st)
    torch.cuda.synchronize()
    print(scene, ro.shape[0], f"{(time.perf_counter() - t0) / 200 * 1e6:.1f} us per eslam_ray_order")
