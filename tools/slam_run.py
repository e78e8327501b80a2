"""BASELINE.json configs[2] in synthetic form: the tracking + mapping loop (myslam_amd/slam.py) on the HIP path over an
analytic RGB-D sequence in the Replica room0 geometry (680 x 1200 images, room0 bound and planes, the reference's
Replica iteration counts and pixel budgets), reporting ATE, render quality and where the time goes.
    python tools/slam_run.py [n_frames] [iters_first] [eager|graph|syncfree]
"""
import sys, time, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from myslam_amd import eval_ate, scene as scn, slam, synthscene

n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 41
iters_first = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
mode = sys.argv[3] if len(sys.argv) > 3 else 'eager'
dev = torch.device('cuda:0')
sc = scn.make_scene('room0')
cfg = slam.SlamConfig(iters_first=iters_first)          # configs/ESLAM.yaml + configs/Replica/replica.yaml values
t0 = time.perf_counter()
frames = synthscene.make_sequence(sc, n_frames, device=dev)
torch.cuda.synchronize()
print(f"sequence of {n_frames} frames {sc.H}x{sc.W} rendered in {time.perf_counter()-t0:.1f} s", flush=True)
torch.manual_seed(0)
if mode in ('graph', 'syncfree'):
    from myslam_amd.slam_graph import GraphedSlam
    s = GraphedSlam(sc, cfg, device=dev, seed=0, use_graphs=(mode == 'graph'))
else:
    s = slam.Slam(sc, cfg, device=dev, seed=0)
marks = []
def on_frame(s_, i):
    torch.cuda.synchronize()
    marks.append(time.perf_counter())
    if i % 8 == 0:
        e = float((s_.estimate_c2w_list[i][:3, 3] - s_.gt_c2w_list[i][:3, 3]).norm())
        print(f"frame {i}: {marks[-1]-t1:.1f} s, translation error {e*100:.2f} cm", flush=True)
t1 = time.perf_counter()
est = s.run(frames, on_frame=on_frame)
torch.cuda.synchronize()
total = time.perf_counter() - t1
ate = eval_ate.evaluate([e.cpu().numpy() for e in est], [f[3].cpu().numpy() for f in frames])
room = synthscene.AnalyticRoom(sc.bound)
pose = synthscene.trajectory(2 * n_frames, sc.bound, yaw_step_deg=0.75)[2 * n_frames // 2 + 1].to(dev)
gd, gc = synthscene.render_frame(room, sc, pose, dev)
q = s.render_quality(gc, gd, pose)
st = s.stats
first = marks[0] - t1
print(f"ATE rmse {ate['rmse']*100:.2f} cm (mean {ate['mean']*100:.2f}, max {ate['max']*100:.2f}); held-out view PSNR {q['psnr']:.2f} dB, depth L1 {q['depth_l1']*100:.2f} cm")
cap = st.get("capture_seconds", 0.0)
if cap:
    print(f"[{mode}] {st['graphs']} graph builds (warm-up + capture, one-time kernel loading included) {cap:.2f} s; "
          f"loop without them {total-cap:.2f} s = {n_frames/(total-cap):.1f} frames/s, "
          f"{(total-cap)/(st['tracking_iters']+st['mapping_iters'])*1e3:.3f} ms/iteration")
print(f"[{mode}] loop {total:.1f} s: first-frame mapping ({iters_first} iterations) {first:.1f} s = {first/iters_first*1e3:.2f} ms/iteration; "
      f"remaining {n_frames-1} frames {total-first:.1f} s = {(n_frames-1)/(total-first):.1f} frames/s "
      f"({st['tracking_iters']} tracking + {st['mapping_iters']-iters_first} mapping iterations, "
      f"{(total-first)/(st['tracking_iters']+st['mapping_iters']-iters_first)*1e3:.2f} ms/iteration)")
