#!/bin/bash
# The round's profile collection on the GPU box (gpurun -- bash tools/run_profiles.sh <outdir under gpurun_out>):
# for each of the three BASELINE.json workloads the bench lines quote - configs[1] room0 4096 x 64, configs[3] scene0000
# 8192 x 96 (10 % depth-less rays), configs[4] freiburg1_desk 5000 x 56 on the mixed-precision kernels - the kernel-trace
# stats and the PMC passes (one counter group per pass: MI355X_MICROARCH.md "rocprofv3 PMC slots"), merged into
# profiles/r03_traffic.json for THIS library build; then the bench lines that read it.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-r03p}
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-graph --no-extras > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
find $OUT/stats -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $OUT/kernel_stats_bench_4096x64.csv
for wl in "4096x64:room0 4096 56 8 0.0" "8192x96:scene0000 8192 88 8 0.1" "5000x56_lowp:freiburg1_desk 5000 48 8 0.1 lowp"; do
  tag=${wl%%:*}; args=${wl#*:}
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$tag -- python3 $R/tools/dbg_scatter.py $args > $OUT/stats_$tag.out 2> $OUT/stats_$tag.err
  find $OUT/stats_$tag -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $OUT/kernel_stats_eager_$tag.csv
  for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_ATOMIC_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"; do
    name=$(echo $grp | cut -d' ' -f1)
    rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_${tag}_$name -- python3 $R/tools/dbg_scatter.py $args > $OUT/pmc_${tag}_$name.out 2> $OUT/pmc_${tag}_$name.err
  done
  (cd $R && python3 tools/collect_traffic.py $(python3 -c "a='$args'.split(); print(int(a[1])*(int(a[2])+int(a[3])))") $OUT/pmc_${tag}_FETCH_SIZE $OUT/pmc_${tag}_WRITE_SIZE $OUT/pmc_${tag}_TCC_HIT_sum > $OUT/traffic_$tag.json)
  (cd $R && python3 tools/pmc_summary.py $(find $OUT/pmc_${tag}_SQ_WAVE_CYCLES -name '*counter_collection.csv' | head -1) > $OUT/pmc_sq_summary_$tag.txt 2>&1)
done
cd $R
python3 - "$OUT" <<'PY'
import hashlib, json, os, sys
out = sys.argv[1]
root = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
lib = os.path.join(root, "myslam_amd", "lib", "libeslam_hip.so")
doc = {"_how": "tools/run_profiles.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC_HIT_sum TCC_MISS_sum TCC_EA0_ATOMIC_sum passes over tools/dbg_scatter.py "
               "(eager steps) per workload; corrections and the L2 request-size calibration: tools/collect_traffic.py",
       "lib_sha256_16": hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16], "workloads": {}}
for tag in ("4096x64", "8192x96", "5000x56_lowp"):
    p = os.path.join(out, f"traffic_{tag}.json")
    if os.path.exists(p):
        doc["workloads"][tag] = json.load(open(p))
json.dump(doc, open(os.path.join(out, "r03_traffic.json"), "w"), indent=1)
PY
cp $OUT/r03_traffic.json profiles/r03_traffic.json
python3 bench.py > $OUT/bench_4096x64.json 2> $OUT/bench.err
python3 bench.py --strong --steps 50 --warmup 10 > $OUT/bench_strong_n1.json 2> $OUT/bench_strong.err
python3 bench.py --lowp --steps 50 --warmup 10 --no-cpu-baseline > $OUT/bench_lowp_5000x56.json 2> $OUT/bench_lowp.err
# host time of the eager step: Python glue / compiled glue, channels-last / NCHW planes
ESLAM_TORCH_EXT=0 python3 tools/host_profile.py > $OUT/host_profile_python_glue.txt 2>/dev/null
python3 tools/host_profile.py > $OUT/host_profile_compiled_glue.txt 2>/dev/null
ESLAM_TORCH_EXT=0 python3 tools/host_profile.py nchw > $OUT/host_profile_python_glue_nchw.txt 2>/dev/null
python3 tools/host_profile.py nchw > $OUT/host_profile_compiled_glue_nchw.txt 2>/dev/null
# keep the merge small: drop the raw per-dispatch traces, keep summaries
find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*counter_collection.csv' -size +4M -delete; find $OUT -name '*.db' -delete
tail -c 1200 $OUT/bench_4096x64.json; echo; head -c 1500 $OUT/r03_traffic.json
