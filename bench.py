#!/usr/bin/env python3
"""Bench of the ESLAM rendering hot path on MI355X (contract: one JSON line on stdout from rank 0).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one mapping iteration minus the optimiser update: depth-guided sampling ->
render_batch_ray forward -> mapping loss -> backward to the 12 planes and the decoders, on BASELINE.json configs[1]:
synthetic Replica room0, 4096 rays x 64 samples (56 stratified + 8 surface), float32, inputs resident in HBM.
With N > 1 every rank renders its own 4096 rays (weak scaling) and the iteration adds the two collectives of
myslam_amd/parallel.py (16-float loss denominators, 27 MB gradient all-reduce over RCCL/xGMI).

value = ray.samples/s of the whole job = N * R_eff * S / t_step, t_step = max over ranks of (wall time of K steps)/K.
roofline = dominant kernel's ALGORITHMIC bytes per launch / its HIP-event-timed duration.
cpu_baseline = the CPU oracle (a restatement of the reference's PyTorch path, pinned to the reference by
tests/golden) timed on this box's host cores on the same workload - a reported baseline, not the target.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SCENE, RAYS, N_STRAT, N_IMP = "room0", 4096, 56, 8
HBM_PEAK_GBS = 8000.0                 # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
# algorithmic bytes per ray.sample: 12 planes x 4 texels x 32 ch x 4 B gathered (fwd),
# the same footprint scatter-added (bwd), + 16 B of per-sample I/O
BYTES_FWD, BYTES_SCATTER = 6160, 6144


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-iters", type=int, default=20)
    ap.add_argument("--no-graph", action="store_true", help="issue every launch eagerly from Python (no hipGraph)")
    return ap.parse_args()


def kernel_profile(step_fn, iters):
    """Average HIP-event duration (ms) of every kernel of one step; events sit on the launch stream in the C-ABI."""
    from myslam_amd import _hip
    lib = _hip.lib()
    sums, cnt = {}, {}
    buf = (ctypes.c_float * 12)()
    for _ in range(iters):
        _hip.check(lib.eslam_profile_enable(1), "profile_enable")
        step_fn()
        torch.cuda.synchronize()
        _hip.check(lib.eslam_profile_read(buf), "profile_read")
        for i in range(12):
            if buf[i] >= 0:
                n = lib.eslam_profile_name(i).decode()
                sums[n] = sums.get(n, 0.0) + buf[i]
                cnt[n] = cnt.get(n, 0) + 1
    lib.eslam_profile_enable(0)
    return {k: sums[k] / cnt[k] for k in sums}


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


def step_statistics(step, wl, n=100):
    """SURVEY.md section 8(d) asks for median and p10 / p90 and for the forward-only rate beside the headline number.
    Each step is bracketed by its own pair of events here (that adds a little launch gap per step, so the median sits a
    few microseconds above ms_per_step, which times K steps back to back)."""
    from myslam_amd import harness
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record()
        step()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    pct = {"p10": round(ts[n // 10], 4), "p50": round(ts[n // 2], 4), "p90": round(ts[(9 * n) // 10], 4), "n": n}

    def fwd():
        with torch.no_grad():
            wl.renderer.render_batch_ray(wl.planes, wl.decoders, wl.rays_d, wl.rays_o, wl.device, wl.truncation,
                                         gt_depth=wl.gt_depth)
    fwd_only = None
    try:
        g = harness.GraphedStep(fwd, [])
        for _ in range(10):
            g()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            g()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        fwd_only = {"ms": round(ms, 4), "value": wl.R * wl.S / (ms * 1e-3), "unit": "ray.samples/s",
                    "what": "sample + render_batch_ray forward only (no_grad: nothing saved for backward), graph replay"}
    except Exception as e:
        fwd_only = {"error": f"{type(e).__name__}: {e}"}
    return {"percentiles": pct, "forward_only": fwd_only}


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
    wl = harness.make_workload(SCENE, RAYS, N_STRAT, N_IMP, device=dev, seed=rank)
    mapper = None
    if world > 1 or os.environ.get("BENCH_FORCE_DP") == "1":      # BENCH_FORCE_DP: exercise the sharded step on one rank
        from myslam_amd.parallel import ShardedMapper
        if world == 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        mapper = ShardedMapper(wl)
        step = mapper.step
    else:
        step = wl.step
    eager_step = step
    graphed = False
    if not args.no_graph:
        # capture the iteration into hipGraphs: the step is launch-bound when issued from Python.  With N > 1 the two
        # collective-free phases of the sharded step are captured separately and the two RCCL all-reduces stay eager.
        try:
            if mapper is not None:
                mapper.capture()
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
    extras = None
    if rank == 0 and world == 1:
        extras = step_statistics(step, wl)       # outside the timed region: per-step percentiles, forward-only rate
    if rank == 0:
        prof = kernel_profile(wl.step, args.profile_iters)      # eager: the HIP events sit inside the C-ABI calls
        n = wl.R * wl.S
        alg = {"render_fwd_kernel": BYTES_FWD * n, "scatter_sort_kernel": BYTES_SCATTER * n}
        dom = max((k for k in prof if k in alg), key=lambda k: prof[k])
        achieved = alg[dom] / (prof[dom] * 1e-3) / 1e9
        other = [k for k in alg if k != dom][0]
        # HBM-side traffic per launch comes from the separate rocprofv3 --pmc passes of this build (profiles/): counters
        # cannot be read from inside this process
        traffic, traffic_note = None, None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
            if dom in tj and wl.R * wl.S == 4096 * 64:
                traffic = tj[dom]["traffic_bytes"]
                traffic_note = "profiles/r01_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload)"
        except Exception:
            pass
        out = {
            "metric": "ray.samples/s (render+bwd)", "value": value, "unit": "ray.samples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"Replica room0 (synthetic), {RAYS} rays x {wl.S} samples ({N_STRAT}+{N_IMP}) per GPU, "
                                   "mapping iteration: sample + render fwd + loss + bwd (planes+decoders), no optimiser",
                       "rays_after_aabb_filter": wl.R, "samples_per_ray": wl.S, "plane_bytes": wl.scene.plane_bytes,
                       "planes_layout": "channels_last", "parallelism": f"ray-sharded dp{world}",
                       "gradient_exchange": None if mapper is None else
                       ("block-sparse: union of touched texels (ESLAM_DP_COMPACT=0 for dense)" if mapper.compact
                        else "dense all-reduce of the 27 MB flat buffer"),
                       "loss": ("eslam_loss_reduce + all-reduce + eslam_loss_grad" if mapper is not None else
                                "separate eslam_loss_value launch" if harness._SEPARATE_LOSS else
                                "sums formed in the forward kernel's epilogue (eslam_render_fwd_loss)") +
                               "; gradients by eslam_loss_grad in the backward",
                       "launch": ("hipGraph replay of the captured iteration" + (" (2 graphs, all-reduces eager)" if mapper is not None
                                                                                              else ""))
                       if graphed else "eager launches from Python"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_note,
                         "note": "the 27 MB of planes are L2 / Infinity-Cache resident and duplicate texel contributions are "
                                 "merged on chip before they reach memory, so algorithmic bytes per second can exceed the HBM "
                                 "peak; traffic is what the memory side actually saw",
                         "algorithmic_bytes_per_launch": alg[dom], "avg_kernel_ms": prof[dom],
                         "second_kernel": {"kernel": other, "avg_kernel_ms": prof.get(other),
                                           "achieved": alg[other] / (prof[other] * 1e-3) / 1e9 if other in prof else None},
                         "whole_step": {"algorithmic_bytes": 12304 * n, "achieved": 12304 * n / (ms_step * 1e-3) / 1e9,
                                        "frac": 12304 * n / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS}},
            "kernel_ms": {k: round(v, 4) for k, v in sorted(prof.items())},
            "step_ms_percentiles": None if extras is None else extras["percentiles"],
            "forward_only": None if extras is None else extras["forward_only"],
        }
    if world > 1:
        dist.barrier()
    line = None
    if rank == 0:
        # reported on rank 0 at N=1 only (a host-side baseline does not change with the GPU count)
        out["cpu_baseline"] = cpu_baseline(wl) if (world == 1 and not args.no_cpu_baseline) else None
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
