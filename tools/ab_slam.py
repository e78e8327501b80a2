"""Per-iteration losses of the toy tracking + mapping loop (deterministic given the seed, up to float-atomic order):
run once per library build (ESLAM_HIP_LIB) and diff the two logs to find the first call that differs."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from myslam_amd import scene as scn, slam, synthscene
dev = torch.device("cuda:0")
sc = scn.make_scene("toy")
cfg = slam.SlamConfig(tracking_pixels=500, tracking_iters=8, ignore_edge_H=10, ignore_edge_W=10, mapping_pixels=1000,
                      iters_first=100, iters=10, every_frame=4, keyframe_every=4)
frames = synthscene.make_sequence(sc, 9, device=dev)
torch.manual_seed(0)
s = slam.Slam(sc, cfg, device=dev, seed=0)
log = []
ml, tl = s.be.mapping_loss, s.be.tracking_loss
def mapping_loss(depth, color, sdf, z, gd, gc, *a, **k):
    v = ml(depth, color, sdf, z, gd, gc, *a, **k)
    log.append(("map", float(v), float(depth.sum()), float(color.sum()), float(sdf.sum()), int(gd.shape[0])))
    return v
def tracking_loss(depth, color, sdf, z, gd, gc, *a, **k):
    v = tl(depth, color, sdf, z, gd, gc, *a, **k)
    log.append(("trk", float(v), float(depth.sum()), float(color.sum()), float(sdf.sum()), int(gd.shape[0])))
    return v
s.be.mapping_loss, s.be.tracking_loss = mapping_loss, tracking_loss
s.run(frames)
for i, r in enumerate(log):
    print(i, r[0], "%.7g %.7g %.7g %.7g %d" % r[1:])
