"""Timings of the other callers of the path on the GPU box (not the headline metric): tracking iteration (pose gradients
only), whole-image render (render_img, Frame_Visualizer's caller), dense field query (Mesher.eval_points' caller)."""
import ctypes, sys, time, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from myslam_amd import harness, losses, _hip
dev = torch.device('cuda:0')
lib = _hip.lib()

def timed(fn, n=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

# tracking: Replica numbers (2000 rays x 40 samples, configs/Replica/replica.yaml:10-11,25-26), decoders frozen, planes detached
wl = harness.make_workload('room0', 2000, 32, 8, device=dev, rays_grad=True)
planes = tuple([p.detach() for p in grp] for grp in wl.planes)
for p in wl.decoders.parameters(): p.requires_grad_(False)
def track():
    wl.rays_o.grad = None; wl.rays_d.grad = None
    d, c, s, z = wl.renderer.render_batch_ray(planes, wl.decoders, wl.rays_d, wl.rays_o, dev, wl.truncation, gt_depth=wl.gt_depth)
    losses.tracking_loss(d, c, s, z, wl.gt_depth, wl.gt_color, wl.truncation).backward()
g = harness.GraphedStep(track, [wl.rays_o, wl.rays_d])
print(f"tracking iteration {wl.R} rays x {wl.S}: eager {timed(track):.3f} ms, graph replay {timed(g):.3f} ms")
buf = (ctypes.c_float * 12)(); lib.eslam_profile_enable(1); track(); torch.cuda.synchronize(); lib.eslam_profile_read(buf); lib.eslam_profile_enable(0)
print('  kernels ms:', {lib.eslam_profile_name(i).decode(): round(buf[i], 4) for i in range(12) if buf[i] >= 0})

# render_img: full Replica image 680 x 1200 = 816000 rays x 40 samples, no grad
r = wl.renderer
gt = torch.full((r.H, r.W), 1.5, device=dev)
c2w = wl.c2w.to(dev)
def img():
    r.render_img(wl.planes, wl.decoders, c2w, wl.truncation, dev, gt_depth=gt)
t = timed(img, n=5, warm=2)
print(f"render_img {r.H}x{r.W} x {wl.S} samples: {t:.1f} ms  ({r.H*r.W*wl.S/t*1e3:.3e} ray.samples/s)")
r.ray_batch_size = 10 ** 9           # one chunk instead of 82
t = timed(img, n=5, warm=2)
print(f"render_img, one chunk: {t:.1f} ms  ({r.H*r.W*wl.S/t*1e3:.3e} ray.samples/s)")

# dense field query: 500k points per call (Mesher.py:141)
pts = (torch.rand(500000, 3, device=dev) * (wl.scene.bound[:, 1] - wl.scene.bound[:, 0]).to(dev) + wl.scene.bound[:, 0].to(dev))
with torch.no_grad():
    t = timed(lambda: wl.decoders(pts, all_planes=wl.planes), n=10, warm=3)
print(f"Decoders.forward 500k points: {t:.3f} ms  ({500000/t*1e3:.3e} points/s)")

# mixed precision forward (configs[4]) vs float32 forward, inference, 4096 x 64 and a whole image in one chunk
from myslam_amd import lowp
wl2 = harness.make_workload('room0', 4096, 56, 8, device=dev)
ph = lowp.half_planes(wl2.planes)
rand = wl2._rand
def f32():
    with torch.no_grad():
        wl2.renderer.render_batch_ray(wl2.planes, wl2.decoders, wl2.rays_d, wl2.rays_o, dev, wl2.truncation, gt_depth=wl2.gt_depth, _rand=rand)
def f16():
    lowp.render_batch_ray_lowp(wl2.renderer, wl2.planes, ph, wl2.decoders, wl2.rays_d, wl2.rays_o, wl2.truncation, wl2.gt_depth, _rand=rand)
buf = (ctypes.c_float * 12)()
for name, fn in (('float32', f32), ('fp16 planes + bf16 MFMA', f16)):
    for _ in range(5): fn()
    ts = []
    for _ in range(20):
        lib.eslam_profile_enable(1); fn(); torch.cuda.synchronize(); lib.eslam_profile_read(buf); ts.append(buf[0])
    lib.eslam_profile_enable(0)
    print(f"forward kernel 4096x64 inference, {name}: {sorted(ts)[10]*1e3:.1f} us")
