#!/bin/bash
# instruction-cache counters of the edge backward kernels in the training bench
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_icache_$TAG
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
CMD="python $ROOT/bench.py --workload train --steps 1 --warmup 1 --no-cpu-baseline --no-roofline"
timeout 300 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --kernel-include-regex "edge_backward_x2h|edge_backward_mfma|edge_mfma" --output-format csv -d $OUT/p1 -o pmc -- $CMD > $OUT/p1.log 2>&1
f=$(find $OUT/p1 -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && cp "$f" $OUT/pass1.csv && rm -rf $OUT/p1
python3 - <<PY
import csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open("$OUT/pass1.csv")):
    agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in d.items():
        v = sorted(v); print("   %-24s max %14.0f median %14.0f n=%d" % (c, v[-1], v[len(v)//2], len(v)))
PY
tail -3 $OUT/p1.log
