"""Instrumented run of the Replica-sized tracking + mapping loop (tracking error jumps between frames 48 and 56):
per-frame translation / rotation error, tracking loss of the kept pose, kept rays, window size, joint_opt.
    python tools/slam_diag.py [n_frames] [iters_first] [scene_variant]
"""
import math, sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from myslam_amd import eval_ate, scene as scn, slam, synthscene

n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 81
iters_first = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
variant = sys.argv[3] if len(sys.argv) > 3 else 'r01'
dev = torch.device('cuda:0')
sc = scn.make_scene('room0')
cfg = slam.SlamConfig(iters_first=iters_first)
kw = {} if variant == 'r01' else dict(variant=variant)
frames = synthscene.make_sequence(sc, n_frames, device=dev, **kw)
torch.manual_seed(0)
s = slam.Slam(sc, cfg, device=dev, seed=0)
log = []
orig_track = s.track
def track(idx, gt_color, gt_depth):
    it0, r0 = s.stats["tracking_iters"], s.stats["tracking_rays"]
    out = orig_track(idx, gt_color, gt_depth)
    log.append((idx, (s.stats["tracking_rays"] - r0) / max(1, s.stats["tracking_iters"] - it0)))
    return out
s.track = track
def on_frame(s_, i):
    e = s_.estimate_c2w_list[i]; g = s_.gt_c2w_list[i]
    te = float((e[:3, 3] - g[:3, 3]).norm()) * 100
    cosang = float(((e[:3, :3].T @ g[:3, :3]).trace() - 1) / 2)
    re = math.degrees(math.acos(max(-1.0, min(1.0, cosang))))
    valid = float((frames[i][2] > 0).float().mean())
    dmean = float(frames[i][2][frames[i][2] > 0].mean()); dstd = float(frames[i][2][frames[i][2] > 0].std())
    kept = log[-1][1] if log and log[-1][0] == i else 0
    print(f"frame {i:3d}: t_err {te:6.2f} cm  r_err {re:5.2f} deg  kept_rays/iter {kept:6.0f}  depth mean {dmean:.2f} std {dstd:.2f}  "
          f"keyframes {len(s_.keyframe_list)}", flush=True)
est = s.run(frames, on_frame=on_frame)
ate = eval_ate.evaluate([e.cpu().numpy() for e in est], [f[3].cpu().numpy() for f in frames])
print(f"[{variant}] ATE rmse {ate['rmse']*100:.2f} cm (mean {ate['mean']*100:.2f}, max {ate['max']*100:.2f}) over {n_frames} frames")
