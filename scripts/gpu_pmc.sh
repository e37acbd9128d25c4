#!/bin/bash
# PMC passes (rocprofv3 --pmc, one counter group per run, kernel-trace only) over a short bench run.
# Usage: bash scripts/gpu_pmc.sh <tag> [kernel-regex]
TAG=${1:-r01}
REGEX=${2:-"edge_mfma|node_proj|node_qmlp|node_qfold|knn_graph|edge_gate"}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > $OUT/counters_list.txt 2>&1
grep -c . $OUT/counters_list.txt
CMD="python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline"
i=0
while read -r GROUP; do
  [ -z "$GROUP" ] && continue
  i=$((i+1))
  echo "== pass $i: $GROUP"
  timeout 300 rocprofv3 --kernel-trace --pmc $GROUP --kernel-include-regex "$REGEX" --output-format csv -d $OUT/p$i -o pmc -- $CMD > $OUT/p$i.log 2>&1
  tail -1 $OUT/p$i.log | cut -c1-200
  f=$(find $OUT/p$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp "$f" $OUT/pass$i.csv && rm -rf $OUT/p$i
done <<'GROUPS'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES
SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_MISC SQ_LDS_ADDR_CONFLICT
FETCH_SIZE
WRITE_SIZE
TCC_HIT_sum TCC_MISS_sum
GRBM_GUI_ACTIVE
GROUPS
ls -la $OUT | head -20
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/pass*.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f.split("/")[-1])
    for k, d in agg.items():
        print("  ", k, {c: round(sum(v) / len(v), 1) for c, v in d.items()}, "n=", len(next(iter(d.values()))))
PY
