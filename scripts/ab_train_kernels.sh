#!/bin/bash
# per-kernel average durations of the training step for every ab_libs/*.so (rocprofv3 --kernel-trace --stats, 6 steps): the A/B of a
# change to ONE node-level kernel, whose effect on the step is inside the box's +-1 % run-to-run spread.
# Usage (on the box): bash scripts/ab_train_kernels.sh "dgrad|fold_grad|q_backward"
cd ${GRAFT_REPO_ROOT:-/root/repo}; ROOT=$(pwd); export TMPDIR=/tmp
PAT=${1:-dgrad|fold_grad|q_backward|zero_rows}
for lib in ab_libs/*.so; do
D=/tmp/abk_$(basename $lib .so); rm -rf $D
( cd /tmp && CBGX_LIBRARY=$ROOT/$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o tr -- python $ROOT/bench.py --workload train --steps 6 --warmup 2 --no-cpu-baseline --no-roofline > /dev/null 2>&1 )
f=$(find $D -name "*kernel_stats.csv" | head -1)
python3 - "$f" "$PAT" "$lib" <<'P'
import csv, re, sys
f, pat, lib = sys.argv[1:4]
for r in csv.DictReader(open(f)):
    if re.search(pat, r["Name"]):
        print(lib, r["Name"].split("(")[0].replace("void ", "").replace("cbgx::", "")[:44], "calls", r["Calls"], "avg us", round(float(r["AverageNs"]) / 1e3, 1))
P
done
