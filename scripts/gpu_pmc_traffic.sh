#!/bin/bash
# HBM traffic of the x2h edge kernel at the bench default: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes
# (kernel-trace only), as MI355X_MICROARCH.md's HBM section prescribes.  Usage: bash scripts/gpu_pmc_traffic.sh <tag>
# Round 4: the x2h edge stage of every layer is one launch of edge_x2h_dual_kernel; the launches that cover all N nodes of a batch
# are the layers 2..6 of each denoiser call (dispatch k of a call's nine: cached D1, cached D2, five full layers, pruned A2, A1),
# after the 3 launches per batch of the static-context construction.
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_traffic_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CMD="python $ROOT/bench.py --steps 3 --warmup 1 --streams 1 --no-cpu-baseline --no-secondary --no-roofline"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $C --kernel-include-regex "edge_x2h_dual_kernel" --output-format csv -d $OUT/p_$C -o pmc -- $CMD > $OUT/$C.log 2>&1
  tail -1 $OUT/$C.log | cut -c1-160
  f=$(find $OUT/p_$C -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp "$f" $OUT/$C.csv && rm -rf $OUT/p_$C
done
python3 - <<PY
import csv, collections, json
out = {}
N_BATCHES, STATIC_LAUNCHES, PER_CALL, FULL = 3, 3, 9, (2, 3, 4, 5, 6)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = sorted(csv.DictReader(open("$OUT/%s.csv" % c)), key=lambda r: int(r["Dispatch_Id"]))
    vals = [float(r["Counter_Value"]) for r in rows][N_BATCHES * STATIC_LAUNCHES:]
    full = [v for k, v in enumerate(vals) if k % PER_CALL in FULL]
    rest = [v for k, v in enumerate(vals) if k % PER_CALL not in FULL]
    out[c] = {"full_layer_mean_KB": sum(full) / max(len(full), 1), "full_layer_max_KB": max(full or [0]), "n_full": len(full),
              "listed_mean_KB": sum(rest) / max(len(rest), 1), "n_listed": len(rest)}
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
