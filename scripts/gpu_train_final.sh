#!/bin/bash
# training rows of the round (after the host-side changes): bench line with roofline + CPU leg, 2 ranks on one GPU, rocprofv3 stats
TAG=${1:-r02zb}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
timeout 600 python bench.py --workload train --steps 10 --warmup 3 2>&1 | tail -1 | tee $OUT/bench_train_$TAG.json | cut -c1-330
CBGX_DIST_BACKEND=gloo timeout 400 python bench.py --workload train --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | tee $OUT/bench_train_2rank_gloo_$TAG.json | cut -c1-330
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train_$TAG -o tr -- python $ROOT/bench.py --workload train --steps 4 --warmup 2 --no-cpu-baseline --no-roofline > $OUT/rocprof_train_$TAG.log 2>&1 ; tail -1 $OUT/rocprof_train_$TAG.log | cut -c1-200 )
f=$(find $OUT/prof_train_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats_train_$TAG.csv && python3 - <<PY
import csv
rows = list(csv.DictReader(open("$f")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms over 6 steps (+setup):", tot / 1e6, "launches", sum(int(r["Calls"]) for r in rows))
for r in rows[:10]:
    print("%-60s calls %6s total %8.2f ms avg %8.1f us  %5.1f%%" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
find $OUT/prof_train_$TAG -name "*kernel_trace.csv" -delete; find $OUT/prof_train_$TAG -name "*.db" -delete 2>/dev/null
python scripts/train_phase_times.py 2>&1 | tail -2
