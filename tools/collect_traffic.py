"""One workload's entry of profiles/r03_traffic.json from rocprofv3 --pmc passes over tools/dbg_scatter.py (eager steps).

    cd /tmp && export TMPDIR=/tmp         # on the GPU box, one pass per counter group (MI355X_MICROARCH.md, rocprofv3 PMC slots)
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out/pmc_fetch -- python3 tools/dbg_scatter.py <workload>
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d out/pmc_write -- python3 tools/dbg_scatter.py <workload>
    rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_ATOMIC_sum --output-format csv -d out/pmc_tcc -- python3 tools/dbg_scatter.py <workload>
    python3 tools/collect_traffic.py <ray_samples> out/pmc_fetch out/pmc_write out/pmc_tcc > entry.json
(tools/run_profiles.sh does this for the three workloads and merges the entries.)

Corrections (MI355X_MICROARCH.md, HBM): FETCH_SIZE / WRITE_SIZE are KB; FETCH_SIZE reports half the bytes of a wide (16 B per
lane) coalesced read - the forward's texel gather and the decoder backward's feature rows are such reads, so theirs is doubled;
the scatter reads one dword per lane (uncalibrated: reported raw); WRITE_SIZE is exact for 16-B streaming stores and for float
atomics.  L2 requests: TCC_HIT_sum + TCC_MISS_sum counts the requests the L1s (and the atomics) send to the L2s; their size is
CALIBRATED in the same pass on a read of known size (dbg_scatter.py sums 256 MiB of float32 once per launch: bytes / requests of
that kernel), and l2_request_bytes = requests x that size is what the forward's L2 -> L1 rate is priced on."""
import collections, csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ray_samples = int(sys.argv[1])
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[2:]:
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
def mean(k, c):
    v = acc[k].get(c)
    return None if not v else sum(v) / len(v)
def pick(prefix):
    ks = [k for k in acc if k.startswith(prefix)]
    return max(ks, key=lambda k: len(acc[k].get("FETCH_SIZE", [])) + len(acc[k].get("TCC_HIT_sum", []))) if ks else None
# request size: the reduction over 256 MiB (the kernel with the largest FETCH_SIZE among torch's reduce kernels)
req_bytes, cal = None, None
red = [k for k in acc if "reduce_kernel" in k and mean(k, "TCC_HIT_sum") is not None]
if red:
    k = max(red, key=lambda k: (mean(k, "TCC_HIT_sum") or 0) + (mean(k, "TCC_MISS_sum") or 0))
    n_req = mean(k, "TCC_HIT_sum") + mean(k, "TCC_MISS_sum")
    if n_req > 0:
        req_bytes = 256 * 1024 * 1024 / n_req
        cal = {"kernel": k[:80], "requests": n_req, "bytes_read": 256 * 1024 * 1024, "bytes_per_request": req_bytes,
               "fetch_size_reported_bytes": None if mean(k, "FETCH_SIZE") is None else mean(k, "FETCH_SIZE") * 1024}
out = {"ray_samples": ray_samples, "l2_request_calibration": cal, "kernels": {}}
for name, prefix, fetch_x2 in (("render_fwd_kernel", "void render_fwd_kernel", True), ("mlp_bwd_kernel", "void mlp_bwd_kernel", True),
                               ("scatter_sort_kernel", "void scatter_sort_kernel", False)):
    k = pick(prefix)
    if k is None:
        continue
    f, w = mean(k, "FETCH_SIZE"), mean(k, "WRITE_SIZE")
    e = {"kernel_name": k[:120], "dispatches": len(acc[k].get("FETCH_SIZE", []))}
    if f is not None:
        e["fetch_bytes"] = f * 1024 * (2 if fetch_x2 else 1)
        e["fetch_correction"] = "x2 (wide coalesced reads)" if fetch_x2 else "raw (dword reads, uncalibrated)"
    if w is not None:
        e["write_bytes"] = w * 1024
    if f is not None and w is not None:
        e["traffic_bytes"] = e["fetch_bytes"] + e["write_bytes"]
    h, m, a = mean(k, "TCC_HIT_sum"), mean(k, "TCC_MISS_sum"), mean(k, "TCC_EA0_ATOMIC_sum")
    if h is not None and m is not None:
        e["l2_hit_rate"] = h / (h + m)
        e["l2_requests"] = h + m
        if req_bytes:
            e["l2_request_bytes"] = (h + m) * req_bytes
    if a is not None:
        e["atomic_requests_64B"] = a
        if name == "scatter_sort_kernel":
            e["atomic_bytes"] = a * 64
    out["kernels"][name] = e
print(json.dumps(out, indent=1))
