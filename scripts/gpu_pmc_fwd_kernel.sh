#!/bin/bash
# SQ counters of any kernel (regex) of the SAMPLING bench, one 200-graph batch, no stream overlap:
#   bash scripts/gpu_pmc_fwd_kernel.sh <regex> <tag>
RE=${1:-node_proj}; TAG=${2:-k}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/pmc_${TAG}; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
export CBGX_OVERLAP=0
CMD="python $ROOT/bench.py --steps 1 --warmup 1 --streams 1 --pockets 20 --no-cpu-baseline --no-secondary --no-roofline"
i=0
while read -r GROUP; do
  [ -z "$GROUP" ] && continue; i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $GROUP --kernel-include-regex "$RE" --output-format csv -d $OUT/p$i -o pmc -- $CMD > $OUT/p$i.log 2>&1
  f=$(find $OUT/p$i -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/pass$i.csv; rm -rf $OUT/p$i
done <<'GROUPS'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES
SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE
GROUPS
python3 - <<PY
import csv, glob, collections, json
summary = {}
for f in sorted(glob.glob("$OUT/pass*.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:70] + " grid " + r.get("Grid_Size", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        print(k)
        for c, v in d.items():
            v = sorted(v); print("   %-28s max %14.0f median %14.0f n=%d" % (c, v[-1], v[len(v)//2], len(v)))
            summary.setdefault(k, {})[c] = {"max": v[-1], "median": v[len(v)//2], "n": len(v)}
json.dump(summary, open("$OUT/summary.json", "w"), indent=1)
PY
rm -f $OUT/pass*.csv
