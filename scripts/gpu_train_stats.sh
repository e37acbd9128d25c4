#!/bin/bash
# rocprofv3 kernel stats of the training bench, top rows only.  Usage: bash scripts/gpu_train_stats.sh <tag> [n rows]
TAG=${1:-ts}; ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train_$TAG -o tr -- python $ROOT/bench.py --workload train --steps 4 --warmup 2 --no-cpu-baseline --no-roofline > $OUT/rocprof_train_$TAG.log 2>&1 )
f=$(find $OUT/prof_train_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats_train_$TAG.csv && python3 - <<PY
import csv
rows = list(csv.DictReader(open("$f")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms over 6 steps (+setup):", round(tot / 1e6, 2), "launches", sum(int(r["Calls"]) for r in rows))
for r in rows[:${2:-16}]:
    print("%-56s calls %5s total %7.2f ms avg %7.1f us %5.1f%%" % (r["Name"].replace("cbgx::","").replace("void ","")[:56], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
find $OUT/prof_train_$TAG -name "*kernel_trace.csv" -delete; find $OUT/prof_train_$TAG -name "*.db" -delete 2>/dev/null
