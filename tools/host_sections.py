"""Host time per section of one eager mapping iteration (monkeypatched timers, no cProfile overhead)."""
import sys, time, collections, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from myslam_amd import harness, ops, _hip
acc = collections.defaultdict(float)
def wrap(obj, name, label=None):
    f = getattr(obj, name)
    label = label or name
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); acc[label] += time.perf_counter() - t; return r
    setattr(obj, name, staticmethod(g) if isinstance(obj, type) and name in ('forward', 'backward') else g)
wrap(ops, 'sample_z'); wrap(ops, 'ray_order_async')
wrap(ops.RenderFn, 'forward', 'RenderFn.forward'); wrap(ops.RenderFn, 'backward', 'RenderFn.backward')
wrap(ops.MappingLossFn, 'forward', 'Loss.forward'); wrap(ops.MappingLossFn, 'backward', 'Loss.backward')
wrap(_hip, 'make_planes'); wrap(_hip, 'make_decoders'); wrap(ops, 'decoder_params'); wrap(_hip, 'stream_handle')
lib = _hip.lib()
for fn in ('eslam_render_fwd', 'eslam_render_bwd', 'eslam_sample_z_all', 'eslam_loss_value', 'eslam_loss_grad', 'eslam_ray_order'):
    f = getattr(lib, fn)
    def mk(f, fn):
        def g(*a):
            t = time.perf_counter(); r = f(*a); acc['C:' + fn] += time.perf_counter() - t; return r
        return g
    setattr(lib, fn, mk(f, fn))
wrap(ops, '_alloc_plane_grads'); wrap(ops, '_split_dec_grads')
_te = torch.empty
def te(*a, **k):
    t = time.perf_counter(); r = _te(*a, **k); acc['torch.empty'] += time.perf_counter() - t; return r
torch.empty = te
wl = harness.make_workload('room0', 4096, 56, 8, device=torch.device('cuda:0'))
for _ in range(30): wl.step()
torch.cuda.synchronize(); acc.clear()
N = 300
t0 = time.perf_counter()
for _ in range(N): wl.step()
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"issue time per step {t_issue/N*1e3:.3f} ms")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]): print(f"  {k:20s} {v/N*1e6:8.1f} us")
