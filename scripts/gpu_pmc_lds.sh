#!/bin/bash
# LDS bank conflicts of every libcbgx kernel: one --pmc pass (kernel-trace only) over the sampling line and one over the training line.
# Usage (repo root on the GPU box): bash scripts/gpu_pmc_lds.sh [tag]
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/pmc_lds_$TAG
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
C="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
timeout 300 rocprofv3 --kernel-trace --pmc $C --kernel-include-regex "cbgx" --output-format csv -d $OUT/s -o pmc -- python $ROOT/bench.py --steps 1 --warmup 1 --streams 1 --no-cpu-baseline --no-secondary --no-roofline > $OUT/sampling.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc $C --kernel-include-regex "cbgx" --output-format csv -d $OUT/t -o pmc -- python $ROOT/bench.py --workload train --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/training.log 2>&1
python3 - <<PY
import csv, glob, collections, json
out = {}
for leg in ("s", "t"):
    fs = glob.glob("$OUT/%s/**/*counter_collection.csv" % leg, recursive=True)
    if not fs: continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0][:70]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES": n[k] += 1
    rows = []
    for k, d in agg.items():
        act = d.get("SQ_LDS_IDX_ACTIVE", 0.0)
        rows.append({"kernel": k, "launches": n[k], "lds_active_cycles": act, "bank_conflict_cycles": d.get("SQ_LDS_BANK_CONFLICT", 0.0),
                     "conflict_share_of_lds_cycles": round(d.get("SQ_LDS_BANK_CONFLICT", 0.0) / act, 4) if act else 0.0,
                     "lds_active_per_busy_cycle": round(act / d["SQ_BUSY_CYCLES"], 4) if d.get("SQ_BUSY_CYCLES") else None})
    rows.sort(key=lambda r: -r["bank_conflict_cycles"])
    out["sampling" if leg == "s" else "training"] = rows
    for r in rows[:14]: print(leg, r["kernel"][:60].ljust(60), r["launches"], "conflict share", r["conflict_share_of_lds_cycles"], "conflict cycles %.3g" % r["bank_conflict_cycles"], "lds/busy", r["lds_active_per_busy_cycle"])
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
PY
rm -rf $OUT/s $OUT/t
