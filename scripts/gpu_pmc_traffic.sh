#!/bin/bash
# HBM traffic of the x2h edge kernel at the bench default: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes
# (kernel-trace only), as MI355X_MICROARCH.md's HBM section prescribes.  Usage: bash scripts/gpu_pmc_traffic.sh <tag>
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_traffic_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CMD="python $ROOT/bench.py --steps 3 --warmup 1 --streams 1 --no-cpu-baseline --no-secondary --no-roofline"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $C --kernel-include-regex "edge_mfma_kernel" --output-format csv -d $OUT/p_$C -o pmc -- $CMD > $OUT/$C.log 2>&1
  tail -1 $OUT/$C.log | cut -c1-160
  f=$(find $OUT/p_$C -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp "$f" $OUT/$C.csv && rm -rf $OUT/p_$C
done
python3 - <<PY
import csv, collections, json
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open("$OUT/%s.csv" % c)):
        agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        out.setdefault(k[:90], {})[c] = {"mean_KB": sum(v) / len(v), "max_KB": max(v), "n": len(v)}
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
