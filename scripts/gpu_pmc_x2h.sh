#!/bin/bash
# SQ counters of the forward x2h edge kernel at the bench default (99.5 k nodes per launch), three --pmc passes (kernel-trace only)
# Usage: bash scripts/gpu_pmc_x2h.sh <tag>
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_x2h_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CMD="python $ROOT/bench.py --steps 1 --warmup 1 --streams 1 --no-cpu-baseline --no-secondary --no-roofline"
i=0
while read -r GROUP; do
  [ -z "$GROUP" ] && continue
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $GROUP --kernel-include-regex "edge_x2h_dual_kernel" --output-format csv -d $OUT/p$i -o pmc -- $CMD > $OUT/p$i.log 2>&1
  f=$(find $OUT/p$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp "$f" $OUT/pass$i.csv && rm -rf $OUT/p$i
done <<'GROUPS'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES
SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_MISC SQ_LDS_ADDR_CONFLICT
GROUPS
python3 - <<PY
import csv, glob, collections, json
res = {}
for f in sorted(glob.glob("$OUT/pass*.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        # full launches only: the largest SQ_WAVES / instruction counts belong to the 99.5 k-node launches
        for c, v in d.items():
            v = sorted(v)
            res.setdefault(k, {})[c] = {"max": v[-1], "median": v[len(v) // 2], "n": len(v)}
json.dump(res, open("$OUT/summary.json", "w"), indent=1)
for k, d in res.items():
    print(k)
    for c, v in d.items():
        print("   %-32s max %14.0f median %14.0f n=%d" % (c, v["max"], v["median"], v["n"]))
PY
