"""profiles/r02_traffic.json from three rocprofv3 --pmc passes over tools/dbg_scatter.py (the bench workload, eager steps).

    cd /tmp && export TMPDIR=/tmp         # on the GPU box, one pass per counter group (MI355X_MICROARCH.md, rocprofv3 PMC slots)
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 tools/dbg_scatter.py
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 tools/dbg_scatter.py
    rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_ATOMIC_sum --output-format csv -d gpurun_out/pmc_tcc -- python3 tools/dbg_scatter.py
    python3 tools/collect_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_tcc > profiles/r02_traffic.json

Corrections (MI355X_MICROARCH.md, HBM): FETCH_SIZE / WRITE_SIZE are KB; FETCH_SIZE reports half the bytes of a wide (16 B per
lane) coalesced read - the forward's texel gather and the decoder backward's feature rows are such reads, so theirs is doubled;
the scatter reads one dword per lane (uncalibrated: reported raw); WRITE_SIZE is exact for 16-B streaming stores and for float
atomics.  The file is keyed by the library's hash: bench.py drops it when the loaded build differs.
"""
import collections, csv, glob, hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
def mean(k, c):
    v = acc[k].get(c)
    return None if not v else sum(v) / len(v)
def pick(prefix, must=None):
    ks = [k for k in acc if k.startswith(prefix) and (must is None or must in k)]
    return max(ks, key=lambda k: len(acc[k].get("FETCH_SIZE", []))) if ks else None
out = {"_how": __doc__.strip().split("\n\n")[0], "lib_sha256_16": hashlib.sha256(open(os.path.join(ROOT, "myslam_amd", "lib", "libeslam_hip.so"), "rb").read()).hexdigest()[:16],
       "ray_samples": 4096 * 64, "kernels": {}}
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
    if a is not None:
        e["atomic_requests_64B"] = a
        if name == "scatter_sort_kernel":
            e["atomic_bytes"] = a * 64
    out["kernels"][name] = e
print(json.dumps(out, indent=1))
