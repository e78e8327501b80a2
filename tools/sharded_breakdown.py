"""Where the time of a ray-sharded step goes on ONE rank (world size 1: the collective is a device copy): GPU time of the
two graphs and of the all-reduce from events on the main stream, host wall time of the whole step, and the same rays
through the plain single-GPU step.
    python tools/sharded_breakdown.py [rays] [scene] [n_strat]"""
import os, socket, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.distributed as dist
from myslam_amd import harness, parallel
R = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
scene = sys.argv[2] if len(sys.argv) > 2 else "scene0000"
ns = int(sys.argv[3]) if len(sys.argv) > 3 else 88
dev = torch.device('cuda:0')
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
wl = harness.make_workload(scene, R, ns, 8, device=dev, zero_frac=0.1, shard=(0, 1))
m = parallel.ShardedMapper(wl)
for _ in range(3): m.step()
m.capture()
for _ in range(20): m.step()
torch.cuda.synchronize()
gf, gb = m._graphs
names = ["graph front (clear, mark+list [side], set sizes, sample, fwd, bwd, pack)", "all_reduce [tail | marked texels]", "graph back (unpack)"]
tot = [0.0] * 3; N = 100
t0 = time.perf_counter()
for _ in range(N): m.step()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / N * 1e3
for it in range(N):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    ev[0].record(); gf.replay(); ev[1].record()
    n = m._marked()
    dist.all_reduce(m._buf[:m._tail_pad + 32 * n]); ev[2].record()
    gb.replay(); ev[3].record()
    torch.cuda.synchronize()
    for k in range(3): tot[k] += ev[k].elapsed_time(ev[k + 1])
sent, dense = m.last_exchange
print(f"{scene} {wl.R} rays x {wl.S}: step wall {wall:.3f} ms (graph replays + one all-reduce, back to back); "
      f"marked {n} texels = {n * 128 / 1e6:.2f} MB of {m._n_plane_elems * 4 / 1e6:.1f} MB; exchange {sent / 1e6:.2f} MB")
for nm, t in zip(names, tot): print(f"  {nm:80s} {t / N * 1e3:8.1f} us (GPU, main stream)")
# how tight the conservative marking is: non-zero 128-byte blocks of the gradient against marked ones
nz = int((m.grads.flat[:m._n_plane_elems].view(-1, 32) != 0).any(1).sum())
print(f"  texels with a non-zero gradient {nz}, marked {n} ({n / max(nz, 1):.2f}x)")
wf = harness.make_workload(scene, R, ns, 8, device=dev, zero_frac=0.1)
g = harness.GraphedStep(wf.step, wf.params())
for _ in range(10): g()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(100): g()
torch.cuda.synchronize(); print(f"  single-GPU step of the same rays: {(time.perf_counter() - t0) / 100 * 1e3:.3f} ms")
# one graph per step (back of the previous iteration + front of this one), then the all-reduce
mp = parallel.ShardedMapper(wl)
mp.step(); mp.capture(pipeline=True)
for _ in range(20): mp.step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): mp.step()
torch.cuda.synchronize(); w2 = (time.perf_counter() - t0) / 200 * 1e3
mp.flush(); torch.cuda.synchronize()
print(f"  pipelined (one graph per step + the all-reduce): step wall {w2:.3f} ms")
# where the HOST spends a pipelined step (no synchronisation inside: if these add up to the wall time, the step is host-bound)
gfp = mp._graphs[0]
th = [0.0, 0.0, 0.0]
for _ in range(200):
    a = time.perf_counter(); gfp.replay(); b = time.perf_counter(); n = mp._marked(); c = time.perf_counter()
    dist.all_reduce(mp._buf[:mp._tail_pad + 32 * n]); d = time.perf_counter()
    th[0] += b - a; th[1] += c - b; th[2] += d - c
torch.cuda.synchronize()
print(f"  host time per pipelined step: graph launch {th[0] / 200 * 1e6:.0f} us, wait for the list's length {th[1] / 200 * 1e6:.0f} us, "
      f"all_reduce call {th[2] / 200 * 1e6:.0f} us")
dist.destroy_process_group()
