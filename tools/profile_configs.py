"""Per-kernel HIP-event times (eslam_profile_*) and step times of the mapping iteration on every BASELINE.json
configuration, one GPU: graph replay, eager launches, and the reference's NCHW plane layout.
    python tools/profile_configs.py [tag]            (tag only labels the output)
"""
import ctypes, json, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from myslam_amd import harness, _hip
dev = torch.device('cuda:0')
lib = _hip.lib()
NK = _hip.PROF_KERNELS

def timed(fn, n=100, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

def kernel_profile(step, iters=10):
    buf = (ctypes.c_float * NK)()
    sums, cnt = {}, {}
    for _ in range(iters):
        lib.eslam_profile_enable(1); step(); torch.cuda.synchronize(); lib.eslam_profile_read(buf)
        for i in range(NK):
            if buf[i] >= 0:
                n = lib.eslam_profile_name(i).decode(); sums[n] = sums.get(n, 0.) + buf[i]; cnt[n] = cnt.get(n, 0) + 1
    lib.eslam_profile_enable(0)
    return {k: round(sums[k] / cnt[k] * 1e3, 1) for k in sums}      # microseconds

tag = sys.argv[1] if len(sys.argv) > 1 else ''
only = sys.argv[2].split(',') if len(sys.argv) > 2 else None
rows = []
for name, scene, R, ns, ni, zf in (("room0_4096x64", "room0", 4096, 56, 8, 0.0), ("scene0000_8192x96", "scene0000", 8192, 88, 8, 0.1),
                                   ("scene0000_1024x96", "scene0000", 1024, 88, 8, 0.1), ("freiburg_5000x56", "freiburg1_desk", 5000, 48, 8, 0.1),
                                   ("room0_200x32", "room0", 200, 24, 8, 0.0),
                                   # in-between sizes (only when named): where the scatter's bundle size switches
                                   ("room0_2048x64", "room0", 2048, 56, 8, 0.0), ("room0_1024x64", "room0", 1024, 56, 8, 0.0),
                                   ("room0_3072x64", "room0", 3072, 56, 8, 0.0)):
    if (only and name not in only) or (not only and name in ("room0_2048x64", "room0_1024x64", "room0_3072x64")): continue
    wl = harness.make_workload(scene, R, ns, ni, device=dev, zero_frac=zf)
    for _ in range(5): wl.step()
    prof = kernel_profile(wl.step)
    eager = timed(wl.step, n=50)
    g = harness.GraphedStep(wl.step, wl.params())
    graph = timed(g)
    row = dict(tag=tag, config=name, rays=wl.R, S=wl.S, graph_ms=round(graph, 4), eager_ms=round(eager, 4),
               rs_per_s=wl.R * wl.S / graph * 1e3, kernels_us=prof)
    del g
    if name in ("room0_4096x64", "scene0000_1024x96"):
        wn = harness.make_workload(scene, R, ns, ni, device=dev, zero_frac=zf, channels_last=False)
        gn = harness.GraphedStep(wn.step, wn.params())
        row["nchw_graph_ms"] = round(timed(gn), 4)
        row["nchw_kernels_us"] = kernel_profile(wn.step)
        del gn, wn
    if name == "freiburg_5000x56":
        # BASELINE.json configs[4]: the same iteration on the mixed-precision kernels (fp16 plane copies, bf16 MFMA both ways)
        from myslam_amd import lowp, ops
        half = lowp.HalfPlanes(wl.planes)
        def lp_step():
            with ops.mixed_precision(half):
                return wl.step()
        for _ in range(5): lp_step()
        row["lowp_kernels_us"] = kernel_profile(lp_step)
        gl = harness.GraphedStep(lp_step, wl.params())
        row["lowp_graph_ms"] = round(timed(gl), 4)
        row["lowp_rs_per_s"] = wl.R * wl.S / row["lowp_graph_ms"] * 1e3
        row["lowp_refresh_ms"] = round(timed(lambda: half.refresh(wl.planes)), 4)
        del gl
    print(json.dumps(row), flush=True)
    del wl
    torch.cuda.empty_cache()
