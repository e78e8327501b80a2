"""The mapping iteration (sample + render fwd + loss + bwd, graph replay) on the other BASELINE.json configurations, one
GPU: scene0000 8192 x 96 with 10 % depth-less rays (configs[3] unsharded, and its 1/8 shard), freiburg1_desk 5000 x 56."""
import sys, time, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from myslam_amd import harness
dev = torch.device('cuda:0')
for scene, R, ns, ni, zf in (("room0", 4096, 56, 8, 0.0), ("scene0000", 8192, 88, 8, 0.1), ("scene0000", 1024, 88, 8, 0.1),
                             ("freiburg1_desk", 5000, 48, 8, 0.1), ("room0", 200, 24, 8, 0.0)):
    wl = harness.make_workload(scene, R, ns, ni, device=dev, zero_frac=zf)
    g = harness.GraphedStep(wl.step, wl.params())
    for _ in range(10): g()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): g()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 100 * 1e3
    print(f"{scene:15s} {wl.R:5d} rays x {wl.S:3d} ({100*zf:.0f} % depth-less), planes {wl.scene.plane_bytes/1e6:5.1f} MB: "
          f"{ms:.3f} ms/iteration = {wl.R*wl.S/ms*1e3:.3e} ray.samples/s", flush=True)
    del g, wl
    torch.cuda.empty_cache()
