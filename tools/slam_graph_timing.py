"""Replay time of the captured tracking / mapping iterations of slam_graph.GraphedSlam at Replica settings
(2000 tracking rays, 4000 mapping rays, 32+8 samples, room0 planes): what one iteration costs once it is a hipGraph."""
import sys, time, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from myslam_amd import scene as scn, slam, synthscene
from myslam_amd.slam_graph import GraphedSlam
dev = torch.device('cuda:0')
sc = scn.make_scene('room0')
s = GraphedSlam(sc, slam.SlamConfig(iters_first=200), device=dev, seed=0)
s.run(synthscene.make_sequence(sc, 45, device=dev))

def rep(g, n=200):
    for _ in range(10): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): g.replay()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

print(f"tracking iteration (2000 rays x 40, pose gradients + Adam): {rep(s._trk.graph):.3f} ms per replay")
for (b, joint, lrf), st in sorted(s._map.items()):
    print(f"mapping iteration, window of {b} frames, joint_opt={joint}: {rep(st.graph):.3f} ms per replay ({st.rays} rays x 40, Adam on planes + decoders{' + poses' if joint else ''})")
