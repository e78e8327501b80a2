"""A tracking iteration (2000 rays x 40, pose gradients only) replayed as a hipGraph 50 times: run under
`rocprofv3 --kernel-trace --stats` to see every kernel of the iteration (tools/bench_other_callers.py times it)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from myslam_amd import harness, losses
dev = torch.device('cuda:0')
wl = harness.make_workload('room0', 2000, 32, 8, device=dev, rays_grad=True)
planes = tuple([p.detach() for p in grp] for grp in wl.planes)
for p in wl.decoders.parameters(): p.requires_grad_(False)
def track():
    wl.rays_o.grad = None; wl.rays_d.grad = None
    d, c, s, z = wl.renderer.render_batch_ray(planes, wl.decoders, wl.rays_d, wl.rays_o, dev, wl.truncation, gt_depth=wl.gt_depth)
    losses.tracking_loss(d, c, s, z, wl.gt_depth, wl.gt_color, wl.truncation).backward()
g = harness.GraphedStep(track, [wl.rays_o, wl.rays_d])
torch.cuda.synchronize()
for _ in range(50): g()
torch.cuda.synchronize()
