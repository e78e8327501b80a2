"""Optimiser step of a mapping iteration on the GPU box: fused multi-tensor Adam
(csrc/eslam_adam.hip) against torch.optim.Adam's GPU implementations on the room0 parameter set (12 planes, 6.8 M
elements + decoders), with the gradients one real backward leaves (sparse: one camera) and with dense gradients."""
import ctypes, sys, time, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from myslam_amd import harness, _hip, optim
dev = torch.device('cuda:0')
lib = _hip.lib()

def timed(fn, n=50, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

def kernel_ms(fn, slot=10, n=20):
    buf = (ctypes.c_float * 12)(); ts = []
    for _ in range(n):
        lib.eslam_profile_enable(1); fn(); torch.cuda.synchronize(); lib.eslam_profile_read(buf); ts.append(buf[slot])
    lib.eslam_profile_enable(0)
    return sorted(ts)[n // 2]

wl = harness.make_workload('room0', 4096, 56, 8, device=dev)
wl.step(); torch.cuda.synchronize()
params = wl.params()
n_el = sum(p.numel() for p in params)
nz = sum(int((p.grad != 0).sum()) for p in params)
print(f"parameter elements {n_el}, non-zero gradient elements after one backward {nz} ({100*nz/n_el:.1f} %)")
grads = [p.grad.clone(memory_format=torch.preserve_format) for p in params]

def groups():
    dec = list(wl.decoders.parameters())
    return [{"params": dec, "lr": 0.001}, {"params": wl.plane_list[:6], "lr": 0.005}, {"params": wl.plane_list[6:], "lr": 0.005}]

for label, dense in (("sparse gradients (one frame's rays)", False), ("dense gradients", True)):
    for p, g in zip(params, grads):
        p.grad = torch.randn_like(g, memory_format=torch.preserve_format) if dense else g.clone(memory_format=torch.preserve_format)
    res = {}
    for name, mk in (("hip fused", lambda: optim.Adam(groups())),
                     ("torch foreach", lambda: torch.optim.Adam(groups(), foreach=True)),
                     ("torch fused", lambda: torch.optim.Adam(groups(), fused=True)),
                     ("torch single-tensor", lambda: torch.optim.Adam(groups(), foreach=False))):
        o = mk()
        res[name] = timed(o.step)
        if name == "hip fused":
            k = kernel_ms(o.step)
            touched = n_el if dense else None
    print(f"{label}: " + ", ".join(f"{k_} {v:.3f} ms" for k_, v in res.items()) + f"; adam_step_kernel {k*1e3:.1f} us")
    # fresh optimiser (first step after construction: every untouched element is skipped)
    o = optim.Adam(groups()); o.step(); torch.cuda.synchronize()
    k1 = kernel_ms(lambda: optim.Adam(groups()).step())
    print(f"   first step of a fresh optimiser (state all zero): adam_step_kernel {k1*1e3:.1f} us")
bytes_dense = n_el * 28
print(f"dense algorithmic bytes {bytes_dense/1e6:.1f} MB (16 B read + 12 B written per element)")
