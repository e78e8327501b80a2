#!/usr/bin/env python3
"""Bench of the ESLAM rendering hot path on MI355X (contract: one JSON line on stdout from rank 0).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one mapping iteration minus the optimiser update: depth-guided sampling ->
render_batch_ray forward -> mapping loss -> backward to the 12 planes and the decoders, float32, inputs resident in HBM.

  N = 1 (default)   BASELINE.json configs[1]: synthetic Replica room0, 4096 rays x 64 samples (56 stratified + 8 surface).
  N > 1 (default)   BASELINE.json configs[3], the split north_star names: ONE synthetic ScanNet scene0000 batch of
                    8192 rays x 96 samples (10 % depth-less rays), identical on every rank, each rank renders its
                    contiguous 1/N slice (parallel.shard_slice); the loss's global set sizes and the texels the batch can
                    touch are computed redundantly on every rank, ONE all-reduce of [tail | those texels' gradients] over RCCL / xGMI:
                    "scaling": "strong".  --strong runs the same workload at N = 1 (the unsharded step through the same
                    code); --weak lets the room0 batch grow with the job instead, 4096 x 64 per rank ("scaling": "weak").

value = ray.samples/s of the whole job = (rays of the job) * S / t_step, t_step = max over ranks of (wall time of K steps)/K.
roofline = per kernel, against the ceiling that actually bounds it (/opt/skills/guides/MI355X_MICROARCH.md): the forward
gather against the aggregate L2 rate, the scatter against the chip-wide float-atomic rate, the decoder backward against
HBM and the fp32 MFMA rate; durations are HIP events on the launch stream, bytes the rocprofv3 PMC counters of THIS build
(profiles/r03_traffic.json, keyed by the library's hash and the workload - stale numbers are dropped, not shown).
cpu_baseline = the CPU oracle (a restatement of the reference's PyTorch path, pinned to the reference by tests/golden)
timed on this box's host cores on the same workload - a reported baseline, not the target.
"""
import argparse
import ctypes
import hashlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SINGLE = dict(scene="room0", rays=4096, n_strat=56, n_imp=8, zero_frac=0.0,
              name="BASELINE configs[1]: Replica room0 (synthetic), 4096 rays x 64 samples (56+8)")
LOWP = dict(scene="freiburg1_desk", rays=5000, n_strat=48, n_imp=8, zero_frac=0.1,
            name="BASELINE configs[4]: TUM freiburg1_desk (synthetic), 5000 rays x 56 samples (48+8), 10 % depth-less rays, fp16 planes + "
                 "bf16 MFMA decoders (mixed-precision kernels forward and backward, float32 accumulation and float32 plane gradients)")
STRONG = dict(scene="scene0000", rays=8192, n_strat=88, n_imp=8, zero_frac=0.1,
              name="BASELINE configs[3]: ScanNet scene0000 (synthetic), ONE batch of 8192 rays x 96 samples (88+8), 10 % "
                   "depth-less rays, ray-sharded")
# ceilings, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0             # HBM3E spec (6.3 TB/s achievable by a streaming copy)
L2_PEAK_GBS = 34500.0             # aggregate L2 -> L1 rate ("L2 (per XCD)": ~34.5 TB/s)
L2_GATHER_MEASURED_GBS = (16800.0, 18800.0)    # guide's measured rate of an L2-resident row gather into LDS
ATOMIC_PEAK_GBS = 1300.0          # chip-wide global float atomics, added bytes
MFMA_F32_PEAK_TFLOPS = 157.3      # fp32-input MFMA = fp32 vector rate
# algorithmic work per ray.sample: 12 planes x 4 texels x 32 ch x 4 B gathered (+16 B of sample
# I/O); the same footprint scatter-added; decoder GEMMs 2624 MAC forward, x3 in the backward (recompute + dX + dW)
BYTES_FWD, BYTES_SCATTER, BYTES_STEP = 6160, 6144, 12304
FLOP_MLP_BWD = 3 * 2 * 2624


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-iters", type=int, default=20)
    ap.add_argument("--no-graph", action="store_true", help="issue every launch eagerly from Python (no hipGraph)")
    ap.add_argument("--weak", action="store_true", help="N > 1: every rank renders its own 4096 x 64 room0 batch")
    ap.add_argument("--strong", action="store_true", help="N = 1: run the configs[3] workload of the N > 1 mode unsharded")
    ap.add_argument("--no-extras", action="store_true", help="skip eager / NCHW / forward-only side measurements")
    ap.add_argument("--lowp", action="store_true", help="N = 1: BASELINE configs[4], the mixed-precision kernels on freiburg1_desk 5000 x 56")
    return ap.parse_args()


def lib_hash():
    from myslam_amd import _hip
    return hashlib.sha256(open(_hip.LIB_PATH, "rb").read()).hexdigest()[:16]


def kernel_profile(step_fn, iters):
    """Average HIP-event duration (ms) of every kernel of one step; events sit on the launch stream in the C-ABI."""
    from myslam_amd import _hip
    lib = _hip.lib()
    nk = _hip.PROF_KERNELS
    sums, cnt = {}, {}
    buf = (ctypes.c_float * nk)()
    for _ in range(iters):
        _hip.check(lib.eslam_profile_enable(1), "profile_enable")
        step_fn()
        torch.cuda.synchronize()
        _hip.check(lib.eslam_profile_read(buf), "profile_read")
        for i in range(nk):
            if buf[i] >= 0:
                n = lib.eslam_profile_name(i).decode()
                sums[n] = sums.get(n, 0.0) + buf[i]
                cnt[n] = cnt.get(n, 0) + 1
    lib.eslam_profile_enable(0)
    return {k: sums[k] / cnt[k] for k in sums}


def timed(fn, n, warm):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def timed_median(fn, n, warm, reps=5):
    """Median of `reps` timings of n calls each: a host-bound (eager) loop shares the box's cores with other tenants, and a
    single short sample of it was off by 2x from one run to the next."""
    for _ in range(warm):
        fn()
    return sorted(timed(fn, n, 0) for _ in range(reps))[reps // 2]


def cpu_baseline(wl, budget_s=25.0, max_iters=5):
    """Oracle on the host cores: forward + mapping loss + backward on the same rays / planes / decoders."""
    from oracle import eslam_oracle as orc
    orc.BILINEAR_IMPL = "grid_sample"          # the torch op the reference itself calls (decoders.py:79-81)
    # the GPU box gives a 1-GPU job a 16-CPU share of a 256-thread host: more torch threads than that oversubscribe
    ncpu = min(len(os.sched_getaffinity(0)), int(os.environ.get("BENCH_CPU_THREADS", "16")))
    torch.set_num_threads(max(1, ncpu))
    planes = tuple([p.detach().cpu().contiguous().requires_grad_(True) for p in grp] for grp in wl.planes)
    params = {k: v.detach().cpu().requires_grad_(True) for k, v in wl.decoders.state_dict().items() if k != "beta"}
    beta = wl.decoders.beta
    beta = beta.detach().cpu().requires_grad_(True) if torch.is_tensor(beta) else float(beta)
    ro, rd = wl.rays_o.detach().cpu(), wl.rays_d.detach().cpu()
    gd, gc = wl.gt_depth.cpu(), wl.gt_color.cpu()
    bound = wl.scene.bound
    times = []
    t_all = time.perf_counter()
    for it in range(max_iters):
        t_rand = torch.rand(wl.R, wl.S)
        t0 = time.perf_counter()
        depth, color, sdf, z = orc.render_batch_ray(planes, params, beta, bound, rd, ro, wl.truncation, gd,
                                                    wl.n_strat, wl.n_imp, t_rand, None, None)
        loss = orc.mapping_loss(depth, color, sdf, z, gd, gc, wl.truncation)
        loss.backward()
        times.append(time.perf_counter() - t0)
        for grp in planes:
            for p in grp:
                p.grad = None
        if time.perf_counter() - t_all > budget_s:
            break
    t = sorted(times)[len(times) // 2]
    return {"value": wl.R * wl.S / t, "unit": "ray.samples/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{len(times)} full iterations of the same {wl.R}x{wl.S} workload (median), torch CPU float32 with "
                      f"F.grid_sample as in the reference, "
                      f"{t * 1e3:.0f} ms/iter"}


def torch_gpu_baseline(wl, iters=10, warmup=3):
    """The same restatement of the reference's PyTorch-op path, run with torch ops on THIS GPU (boolean-mask loss, autograd
    through ~2000 small kernels): what a user of the reference gets on one MI355X without this library.  Context for the
    north_star's ">= 10x the reference single-GPU PyTorch path" target; not the metric."""
    from oracle import eslam_oracle as orc
    orc.BILINEAR_IMPL = "grid_sample"
    dev = wl.device
    planes = tuple([p.detach().contiguous().clone().requires_grad_(True) for p in grp] for grp in wl.planes)   # NCHW
    params = {k: v.detach().clone().requires_grad_(True) for k, v in wl.decoders.state_dict().items() if k != "beta"}
    beta = wl.decoders.beta
    beta = beta.detach().clone().requires_grad_(True) if torch.is_tensor(beta) else float(beta)
    bound = wl.scene.bound.to(dev)

    def one():
        t_rand = torch.rand(wl.R, wl.S, device=dev)
        depth, color, sdf, z = orc.render_batch_ray(planes, params, beta, bound, wl.rays_d.detach(), wl.rays_o.detach(),
                                                    wl.truncation, wl.gt_depth, wl.n_strat, wl.n_imp, t_rand, None, None)
        orc.mapping_loss(depth, color, sdf, z, wl.gt_depth, wl.gt_color, wl.truncation).backward()
        for grp in planes:
            for p in grp:
                p.grad = None
    for _ in range(warmup):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        one()
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / iters
    return {"value": wl.R * wl.S / t, "unit": "ray.samples/s", "ms_per_step": t * 1e3, "kind": "port",
            "what": "oracle (PyTorch-op restatement of the reference path) with device='cuda' on the same MI355X, eager"}


def main():
    # stdout carries exactly one JSON line.  Native libraries print there too (RCCL writes a version banner when its first
    # communicator is created), so the process-level fd 1 points at stderr until the line is ready.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    try:
        line = run()
    finally:
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        os.close(real_stdout)
    if line is not None:
        print(line, flush=True)


def step_statistics(step, n=100):
    """SURVEY.md section 8(d) asks for median and p10 / p90 beside the headline number.  Each step is bracketed by its own
    pair of events here (that adds a little launch gap per step, so the median sits a few microseconds above
    ms_per_step, which times K steps back to back)."""
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record()
        step()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return {"p10": round(ts[n // 10], 4), "p50": round(ts[n // 2], 4), "p90": round(ts[(9 * n) // 10], 4), "n": n}


def side_measurements(cfg, wl, dev):
    """What the same iteration costs a caller that does NOT replay a graph, or keeps the reference's NCHW planes
    (src/ESLAM.py:201-210) - VERDICT r01 weak #5 - and the forward-only rate SURVEY.md section 8(d) asks for."""
    from myslam_amd import harness, losses
    out = {}
    try:
        # a workload of its own: the headline's graph was captured on wl's parameters, which leaves their AccumulateGrad nodes bound to
        # the capture stream - eager steps on the default stream then pay a stream synchronisation per parameter (0.29 -> 0.57 ms)
        wl = harness.make_workload(cfg["scene"], cfg["rays"], cfg["n_strat"], cfg["n_imp"], device=dev, zero_frac=cfg["zero_frac"])
        out["eager_ms_per_step"] = round(timed_median(wl.step, 100, 30), 4)  # same step, every launch issued from Python

        def reference_shaped():      # the reference loop's own call sequence: render_batch_ray, then the loss, then backward
            for p in wl.params():
                p.grad = None
            d, c, s, z = wl.renderer.render_batch_ray(wl.planes, wl.decoders, wl.rays_d, wl.rays_o, wl.device, wl.truncation,
                                                      gt_depth=wl.gt_depth)
            losses.mapping_loss(d, c, s, z, wl.gt_depth, wl.gt_color, wl.truncation).backward()
        out["eager_separate_loss_ms_per_step"] = round(timed_median(reference_shaped, 100, 30), 4)
        wn = harness.make_workload(cfg["scene"], cfg["rays"], cfg["n_strat"], cfg["n_imp"], device=dev,
                                   zero_frac=cfg["zero_frac"], channels_last=False)
        out["nchw_eager_ms_per_step"] = round(timed_median(wn.step, 30, 10, reps=3), 4)     # (before the capture: see above)
        gn = harness.GraphedStep(wn.step, wn.params())
        out["nchw_ms_per_step"] = round(timed(gn, 50, 10), 4)            # reference-layout planes, graph replay
        del gn, wn

        # the same iteration in the trained-like state (planes x 60, SDF-head bias + 0.55: the compositing weights spread over ~20
        # samples of a ray instead of sitting on the first one, as in the reference's initial state the headline runs in)
        wt = harness.make_workload(cfg["scene"], cfg["rays"], cfg["n_strat"], cfg["n_imp"], device=dev,
                                   zero_frac=cfg["zero_frac"], state="trained")
        gt_ = harness.GraphedStep(wt.step, wt.params())
        out["trained_state_ms_per_step"] = round(timed(gt_, 50, 10), 4)
        del gt_, wt

        def fwd():
            with torch.no_grad():
                wl.renderer.render_batch_ray(wl.planes, wl.decoders, wl.rays_d, wl.rays_o, wl.device, wl.truncation,
                                             gt_depth=wl.gt_depth)
        g = harness.GraphedStep(fwd, [])
        ms = timed(g, 100, 10)
        out["forward_only"] = {"ms": round(ms, 4), "value": wl.R * wl.S / (ms * 1e-3), "unit": "ray.samples/s",
                               "what": "sample + render_batch_ray forward only (no_grad: nothing saved for backward), graph replay"}
    except Exception as e:       # report, never hide
        out["error"] = f"{type(e).__name__}: {e}"
    return out


def roofline(prof, n, ms_step, tag="4096x64"):
    """Per-kernel roofline entries + the contract's `roofline` object (the kernel with the longest duration).
    tag: which workload of profiles/r03_traffic.json the PMC bytes are taken from."""
    traffic = {}
    traffic_note = (f"no PMC entry '{tag}' for this build (profiles/r03_traffic.json missing, from another library hash, or another "
                    "ray count): traffic = null")
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "r03_traffic.json")))
        w = tj.get("workloads", {}).get(tag)
        if tj.get("lib_sha256_16") == lib_hash() and w and w.get("ray_samples") == n:
            traffic = w["kernels"]
            traffic_note = (f"profiles/r03_traffic.json['{tag}']: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC passes of this library build "
                            "on this workload (tools/run_profiles.sh)")
    except Exception:
        pass

    def entry(kernel, bound, alg_bytes, peak, unit="GB/s", extra=None):
        if kernel not in prof:
            return None
        t = prof[kernel] * 1e-3
        tr = traffic.get(kernel, {})
        e = {"kernel": kernel, "bound": bound, "avg_kernel_ms": round(prof[kernel], 5), "peak": peak, "unit": unit,
             "algorithmic_bytes_per_launch": alg_bytes, "traffic": tr.get("traffic_bytes")}
        if bound == "l2":
            # what the L2s actually served to the L1s: TCC_HIT + TCC_MISS requests x the request size calibrated in the same PMC
            # pass (tools/collect_traffic.py).  The algorithmic figure (every texel corner a 16-B-per-lane request) is kept beside it:
            # the difference is what the per-CU L1s absorbed.
            rb = tr.get("l2_request_bytes")
            e["l2_request_bytes"] = rb
            e["algorithmic_rate_GBs"] = alg_bytes / t / 1e9
            if rb is not None:
                e["achieved"] = rb / t / 1e9
            else:             # no PMC entry for this build: the event-timed rate of the algorithmic requests (an upper bound on the L2 rate)
                e["achieved"] = alg_bytes / t / 1e9
                e["achieved_is"] = "algorithmic request bytes / kernel time (no PMC entry for this library build)"
            e["peak_note"] = (f"aggregate L2 -> L1 rate; the guide's MEASURED rate of an L2-resident row gather is "
                              f"{L2_GATHER_MEASURED_GBS[0] / 1e3:.1f}-{L2_GATHER_MEASURED_GBS[1] / 1e3:.1f} TB/s")
        elif bound == "atomic":      # what the memory side adds: WRITE_SIZE of the atomics (exact for float atomics)
            ab = tr.get("atomic_bytes")
            e["achieved"] = None if ab is None else ab / t / 1e9
            e["algorithmic_rate_GBs"] = alg_bytes / t / 1e9
            # the same kernel against HBM on ALL its counter bytes (feature-gradient rows fetched + atomics): with one ray order per
            # plane orientation (round 3) the atomics halved (73 -> 36 MB at 4096 x 64) and the rows are no longer shared in L2
            # (95 -> 208 MB fetched): the kernel got faster and moved AWAY from the atomic ceiling - neither fraction is near 1,
            # it is bound by latency and instruction issue (profiles/r03_pmc_sq_summary_*.txt)
            tb = tr.get("traffic_bytes")
            e["hbm_frac_on_counter_bytes"] = None if tb is None else tb / t / 1e9 / HBM_PEAK_GBS
        elif bound == "hbm":
            tb = tr.get("traffic_bytes")
            e["achieved"] = None if tb is None else tb / t / 1e9
        e["frac"] = None if e.get("achieved") is None else e["achieved"] / peak
        if extra:
            e.update(extra)
        return e

    ks = [entry("render_fwd_kernel", "l2", BYTES_FWD * n, L2_PEAK_GBS),
          entry("scatter_sort_kernel", "atomic", BYTES_SCATTER * n, ATOMIC_PEAK_GBS),
          entry("mlp_bwd_kernel", "hbm", 2 * 512 * n, HBM_PEAK_GBS,
                extra={"mfma": {"flops_per_launch": FLOP_MLP_BWD * n, "peak_tflops": MFMA_F32_PEAK_TFLOPS,
                                "achieved_tflops": None if "mlp_bwd_kernel" not in prof else
                                FLOP_MLP_BWD * n / (prof["mlp_bwd_kernel"] * 1e-3) / 1e12,
                                "note": "useful flops (2624 MAC x 3); the 16x16x4 tiles are zero-padded on the 1- and 3-wide output layers"}})]
    ks = [k for k in ks if k]
    step_traffic = sum(v.get("traffic_bytes", 0) for v in traffic.values()) if traffic else None
    whole = {"bound": "hbm", "traffic": step_traffic, "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "achieved": None if step_traffic is None else step_traffic / (ms_step * 1e-3) / 1e9,
             "algorithmic_bytes": BYTES_STEP * n, "algorithmic_rate_GBs": BYTES_STEP * n / (ms_step * 1e-3) / 1e9,
             "note": "memory-side bytes of all kernels of a step (PMC) / ms_per_step; the algorithmic figure of SURVEY.md 8(d) "
                     "is kept as a labelled extra: the planes are L2 / Infinity-Cache resident, so it is NOT an HBM rate"}
    whole["frac"] = None if whole["achieved"] is None else whole["achieved"] / HBM_PEAK_GBS
    longest = max(ks, key=lambda e: e["avg_kernel_ms"]) if ks else None
    # the contract's object describes the longest kernel; without PMC bytes for THIS build (the scatter's and the decoder
    # backward's ceilings are priced on counter bytes) it describes the longest kernel whose rate the HIP events alone give
    measurable = [e for e in ks if e.get("frac") is not None]
    dom = max(measurable, key=lambda e: e["avg_kernel_ms"]) if measurable else longest
    top = None
    if dom is not None:
        top = {k: dom.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_kernel_ms",
                                       "algorithmic_bytes_per_launch")}
        if dom is not longest:
            top["note"] = (f"the longest kernel is {longest['kernel']} ({longest['avg_kernel_ms']} ms), whose ceiling is priced on PMC "
                           "bytes that were not collected for this library build and workload; reported instead: the longest kernel with an "
                           "event-timed rate")
        top["traffic_source"] = traffic_note
        top["kernels"] = ks
        top["whole_step"] = whole
    return top


def run():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from myslam_amd import harness
    strong = (world > 1 and not args.weak) or (world == 1 and args.strong)
    lowp = bool(args.lowp)
    if lowp and (world > 1 or strong):
        raise SystemExit("--lowp is the single-GPU configs[4] line")
    cfg = STRONG if strong else (LOWP if lowp else SINGLE)
    if strong:       # ONE batch, the same on every rank; this rank keeps its contiguous slice
        wl = harness.make_workload(cfg["scene"], cfg["rays"], cfg["n_strat"], cfg["n_imp"], device=dev,
                                   zero_frac=cfg["zero_frac"], seed=0, shard=(rank, world))
    elif world > 1:  # --weak: the batch grows with the job (4096 rays per rank), every rank still draws all of it and keeps its slice
        wl = harness.make_workload(cfg["scene"], cfg["rays"] * world, cfg["n_strat"], cfg["n_imp"], device=dev,
                                   zero_frac=cfg["zero_frac"], seed=0, shard=(rank, world))
    else:
        wl = harness.make_workload(cfg["scene"], cfg["rays"], cfg["n_strat"], cfg["n_imp"], device=dev,
                                   zero_frac=cfg["zero_frac"], seed=0)
    mapper = None
    if world > 1 or strong or os.environ.get("BENCH_FORCE_DP") == "1":   # BENCH_FORCE_DP: the sharded step on one rank
        from myslam_amd.parallel import ShardedMapper
        if world == 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        mapper = ShardedMapper(wl)
        step = mapper.step
    elif lowp:
        from myslam_amd import lowp as lp, ops as _ops
        half = lp.HalfPlanes(wl.planes)        # (no optimiser in the timed step: the half copies stay valid)

        def step():
            with _ops.mixed_precision(half):
                return wl.step()
    else:
        step = wl.step
    eager_step = step
    graphed = False
    if not args.no_graph:
        # capture the iteration into hipGraphs: the step is launch-bound when issued from Python.  With a mapper the
        # collective-free halves of the sharded step are captured separately and the one all-reduce stays eager.
        try:
            if mapper is not None:
                # ONE graph per step - [unpack of the previous iteration's all-reduced gradients, this iteration's sampler, forward,
                # backward, pack] - and the one all-reduce behind it (ShardedMapper.capture); flush() below drains the last iteration
                # INSIDE the timed region, so K timed steps hold K fronts and K + 1 backs
                mapper.capture(pipeline=True)
            else:
                step = harness.GraphedStep(eager_step, wl.params())
            graphed = True
        except Exception as e:       # report, never hide: the JSON line says which mode was timed
            print(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); timing eager launches", file=sys.stderr)
            step = eager_step

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if mapper is not None and graphed:
        mapper.flush()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        r_tot = torch.tensor([wl.R], device=dev, dtype=torch.float64)
        dist.all_reduce(r_tot)
        total_rays = int(r_tot.item())
    else:
        total_rays = wl.R
    ms_step = dt / args.steps * 1e3
    value = total_rays * wl.S / (dt / args.steps)

    out = None
    prof = None
    if mapper is not None:           # the sharded step contains collectives: every rank takes part in the profiled steps
        prof = kernel_profile(mapper._eager, max(3, args.profile_iters // 4))
    if rank == 0:
        pct = step_statistics(step) if world == 1 else None        # outside the timed region
        if prof is None:
            prof = kernel_profile(eager_step, args.profile_iters)   # eager: the HIP events sit inside the C-ABI calls
        n = wl.R * wl.S
        out = {
            "metric": "ray.samples/s (render+bwd)", "value": value, "unit": "ray.samples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True,
            "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "f16 planes + bf16 MFMA, f32 accumulation" if lowp else "f32", "data": "synthetic",
            "config": {"workload": cfg["name"] + "; mapping iteration: sample + render fwd + loss + bwd (planes+decoders), "
                                   "no optimiser",
                       "rays_of_the_job": total_rays, "rays_this_rank": wl.R, "samples_per_ray": wl.S,
                       "plane_bytes": wl.scene.plane_bytes, "planes_layout": "channels_last",
                       "parallelism": f"ray-sharded dp{world}" if mapper is not None else "single GPU",
                       "gradient_exchange": None if mapper is None else
                       ("[dense tail | texels the batch's rays can touch] of the flat gradient buffer, the list built on the device "
                        "from ray geometry (ESLAM_DP_COMPACT=0 for a dense all-reduce)" if mapper.compact
                        else "dense all-reduce of the flat gradient buffer"),
                       "collectives_per_step": None if mapper is None else 1,
                       "loss": "sums formed in the forward kernel's epilogue (eslam_render_fwd_loss), gradients formed inside "
                               "the backward kernel (eslam_render_bwd_loss)" +
                               ("; global set sizes computed redundantly on every rank from the whole batch (eslam_loss_set_sizes)"
                                if mapper is not None else ""),
                       "launch": ("hipGraph replay of the captured iteration" + (" (one graph per step: [unpack of the previous iteration, this iteration up to its pack], the one all-reduce eager behind it)" if mapper is not None
                                                                                              else ""))
                       if graphed else "eager launches from Python"},
            # (ray-sharded runs: this rank's kernels on this rank's shard; the PMC bytes are collected for the N = 1 workload only)
            "roofline": roofline(prof, n, ms_step, "5000x56_lowp" if lowp else ("8192x96" if strong else "4096x64")),
            "kernel_ms": {k: round(v, 4) for k, v in sorted(prof.items())},
            "step_ms_percentiles": pct,
        }
        if mapper is None and not lowp and not args.no_extras:
            out.update(side_measurements(cfg, wl, dev))
        if strong and not args.no_extras:
            # the SAME batch unsharded on this one GPU, through the single-GPU step: what a strong-scaling speed-up is against
            try:
                wf = harness.make_workload(cfg["scene"], cfg["rays"], cfg["n_strat"], cfg["n_imp"], device=dev,
                                           zero_frac=cfg["zero_frac"], seed=0)
                gf = harness.GraphedStep(wf.step, wf.params())
                ms1 = timed(gf, 50, 10)
                out["single_gpu_same_workload"] = {"ms_per_step": round(ms1, 4), "value": wf.R * wf.S / (ms1 * 1e-3),
                                                   "unit": "ray.samples/s", "rays": wf.R,
                                                   "what": "the whole batch on rank 0's GPU, single-GPU graph replay, measured "
                                                           "in this job after the timed region"}
                del gf, wf
            except Exception as e:
                out["single_gpu_same_workload"] = {"error": f"{type(e).__name__}: {e}"}
    if world > 1:
        dist.barrier()
    line = None
    if rank == 0:
        # reported on rank 0 at N=1 only (a host-side baseline does not change with the GPU count)
        out["cpu_baseline"] = cpu_baseline(wl) if (world == 1 and not strong and not lowp and not args.no_cpu_baseline) else None
        if out["cpu_baseline"] is not None:
            # same baseline leg, second device: the port's PyTorch ops run on this GPU instead of the host cores
            try:
                g = torch_gpu_baseline(wl)
                g["hip_path_speedup"] = out["value"] / g["value"]
                out["cpu_baseline"]["same_port_on_this_gpu"] = g
            except Exception as e:
                out["cpu_baseline"]["same_port_on_this_gpu"] = {"error": f"{type(e).__name__}: {e}"}
        line = json.dumps(out)
    if world > 1:
        dist.barrier()
    if world > 1 or mapper is not None:
        import torch.distributed as dist2
        if dist2.is_initialized():
            dist2.destroy_process_group()
    return line


if __name__ == "__main__":
    main()
