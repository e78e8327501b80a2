"""Where the time of a ray-sharded step goes on ONE rank (world size 1: the collectives are device copies): GPU time per
phase from events on the main stream, and host wall time of the whole step.
    python tools/sharded_breakdown.py [rays] [n_strat]"""
import os, socket, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.distributed as dist
from myslam_amd import harness, parallel
R = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device('cuda:0')
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
wl = harness.make_workload("scene0000", R, 88, 8, device=dev, zero_frac=0.1)
m = parallel.ShardedMapper(wl)
for _ in range(3): m.step()
m.capture()
for _ in range(10): m.step()
torch.cuda.synchronize()
ga, gb, gc = m._graphs
names = ["clear", "graph_a(sample,fwd,mark,pack)", "sync all_reduce", "unpack", "graph_b(bwd)", "union nonzero + wait", "exchange(pack,all_reduce,unpack)"]
tot = [0.0] * len(names); host = 0.0; N = 50
for it in range(N):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(names) + 1)]
    t0 = time.perf_counter()
    ev[0].record()
    if not m.grads.clean:
        m.grads.zero_blocks_(m._last_idx, m._n_plane_elems) if m._last_idx is not None else m.grads.flat.zero_()
    ev[1].record()
    ga.replay(); ev[2].record()
    dist.all_reduce(m._sync); ev[3].record()
    parallel.sync_unpack(m._sync, m._pre.acc, m._gacc, m._touched); ev[4].record()
    e2 = torch.cuda.Event(); e2.record()
    gb.replay(); ev[5].record()
    cur = torch.cuda.current_stream(dev)
    with torch.cuda.stream(m._side):
        m._side.wait_event(e2)
        idx = m._touched.nonzero().squeeze(1)
    cur.wait_stream(m._side); idx.record_stream(cur); ev[6].record()
    m.grads.exchange_blocks(idx, m._n_plane_elems, None); m._last_idx = idx; m.grads.clean = False
    ev[7].record()
    torch.cuda.synchronize()
    host += time.perf_counter() - t0
    for k in range(len(names)): tot[k] += ev[k].elapsed_time(ev[k + 1])
print(f"scene0000 {wl.R} rays x {wl.S}: step wall {host / N * 1e3:.3f} ms; union {idx.numel()} blocks = {idx.numel() * 128 / 1e6:.2f} MB of {m._n_plane_elems * 4 / 1e6:.1f} MB")
for n, t in zip(names, tot): print(f"  {n:40s} {t / N * 1e3:8.1f} us (GPU, main stream)")
wf = harness.make_workload("scene0000", R, 88, 8, device=dev, zero_frac=0.1)
g = harness.GraphedStep(wf.step, wf.params())
for _ in range(10): g()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): g()
torch.cuda.synchronize(); print(f"  single-GPU step of the same rays: {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms")
dist.destroy_process_group()
