#!/bin/bash
# The round's profile collection on the GPU box (gpurun -- bash tools/run_profiles.sh <outdir under gpurun_out>):
# kernel-trace stats of the bench, the PMC passes (one counter group per pass: MI355X_MICROARCH.md "rocprofv3 PMC slots"),
# profiles/r02_traffic.json for THIS library build, then the default bench run that reads it.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-r02p}
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-graph --no-extras > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_ATOMIC_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"; do
  name=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_$name -- python3 $R/tools/dbg_scatter.py > $OUT/pmc_$name.out 2> $OUT/pmc_$name.err
done
cd $R
python3 tools/collect_traffic.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_TCC_HIT_sum > $OUT/r02_traffic.json
python3 tools/pmc_summary.py $(find $OUT/pmc_SQ_WAVE_CYCLES -name '*counter_collection.csv' | head -1) > $OUT/pmc_sq_summary.txt 2>&1
cp $OUT/r02_traffic.json profiles/r02_traffic.json
find $OUT/stats -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $OUT/kernel_stats_bench_4096x64.csv
python3 bench.py > $OUT/bench_4096x64.json 2> $OUT/bench.err
python3 bench.py --strong --steps 50 --warmup 10 > $OUT/bench_strong_n1.json 2> $OUT/bench_strong.err
# keep the merge small: drop the raw per-dispatch traces, keep summaries
find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*counter_collection.csv' -size +8M -delete
tail -c 1500 $OUT/bench_4096x64.json; echo; cat $OUT/r02_traffic.json | head -60
