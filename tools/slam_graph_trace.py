"""    python tools/slam_graph_trace.py [frames [map]]
The captured tracking iteration of slam_graph.GraphedSlam (Replica settings) replayed 100 times: run under
`rocprofv3 --kernel-trace --stats` to list every kernel of one whole tracking iteration - pixel pick, pose -> rays, pre-filter,
render, outlier mask, loss, backward to the pose, Adam, best-pose bookkeeping (tools/slam_graph_timing.py times the replay)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from myslam_amd import scene as scn, slam, synthscene
from myslam_amd.slam_graph import GraphedSlam
dev = torch.device('cuda:0')
sc = scn.make_scene('room0')
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 6
s = GraphedSlam(sc, slam.SlamConfig(iters_first=50), device=dev, seed=0)
s.run(synthscene.make_sequence(sc, frames, device=dev))
torch.cuda.synchronize()
if len(sys.argv) > 2 and sys.argv[2] == "map":      # 100 replays of the mapping iteration with the largest window instead
    key = max(s._map, key=lambda k: k[0])
    print("replaying the mapping iteration of a window of", key[0], "frames", flush=True)
    for _ in range(100): s._map[key].graph.replay()
else:
    for _ in range(100): s._trk.graph.replay()
torch.cuda.synchronize()
