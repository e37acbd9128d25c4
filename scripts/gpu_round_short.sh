#!/bin/bash
# Trimmed confirmation run: bench at four batch shapes, rocprofv3 kernel stats of the bench command, training bench.
TAG=${1:-r01}
OUT=${GRAFT_REPO_ROOT:-$(pwd)}/gpurun_out
mkdir -p $OUT
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
timeout 600 python bench.py 2>/dev/null | tail -1 > $OUT/bench_$TAG.json
timeout 300 python bench.py --pockets 10 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_p10_$TAG.json
timeout 300 python bench.py --pockets 1 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_p1_$TAG.json
timeout 300 python bench.py --pockets 1 --samples 1 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_p1s1_$TAG.json
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o bench -- python ${GRAFT_REPO_ROOT:-.}/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/rocprof_$TAG.log 2>&1 )
f=$(find $OUT/prof_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats_$TAG.csv
find $OUT/prof_$TAG -name "*kernel_trace.csv" -delete
timeout 300 python bench.py --workload train --steps 10 --warmup 3 2>/dev/null | tail -1 > $OUT/bench_train_$TAG.json
python - <<PY
import json
for n in ("bench", "bench_p10", "bench_p1", "bench_p1s1", "bench_train"):
    d = json.load(open("$OUT/%s_$TAG.json" % n))
    r = d["roofline"]
    print(n, d["value"], d["ms_per_step"], r["frac"], r["avg_launch_us"], d.get("cpu_baseline", {}).get("value"))
PY
head -4 $OUT/kernel_stats_$TAG.csv | cut -c1-160
