#!/bin/bash
# kernel trace (start order, grid sizes) of a few training steps: which launches sit between the library's kernels
# (fills, device copies, torch's small kernels).  Output: gpurun_out/trace_train_<tag>/ (kernel_trace.csv)
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/trace_train_$TAG
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o tr -- python $ROOT/bench.py --workload train --steps 2 --warmup 2 --no-cpu-baseline --no-roofline > $OUT/bench.log 2>&1
find $OUT -name "*kernel_trace.csv" | head -1 | xargs -I{} python - {} $OUT <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# keep the last quarter (the last timed step), compact columns
n = len(rows)
keep = rows[3 * n // 4:]
t0 = int(keep[0]["Start_Timestamp"])
with open(sys.argv[2] + "/last_step.tsv", "w") as f:
    for r in keep:
        nm = r["Kernel_Name"]
        nm = nm[:70]
        f.write(f"{(int(r['Start_Timestamp']) - t0) / 1e3:10.1f}\t{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f}\t{r.get('Grid_Size_X', r.get('Grid_Size', ''))}\t{nm}\n")
print(len(rows), "kernels;", len(keep), "kept")
P
find $OUT -name "*.csv" -size +20M -delete
ls -la $OUT
