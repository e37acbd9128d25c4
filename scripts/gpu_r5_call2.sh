#!/bin/bash
# Round 5, second GPU call: every GPU test on the new small-input schedule; small-batch rows per variant (waves per workgroup 2 / 4 / 8,
# fused node stage on / off); step timelines at 1 and 10 graphs; headline check.
# Before: rm -f ab_libs/*.so; python scripts/build_variant.py base; ... mw8 -DCBGX_EDGE_MIN_WAVES=8; ... mw2 -DCBGX_EDGE_MIN_WAVES=2
TAG=${1:-r05b}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
echo "== pytest -m gpu =="
timeout 1500 python -m pytest tests -q -m gpu -x -p no:faulthandler 2>&1 | grep -v "^$" | tail -15 | cut -c1-300 | tee $OUT/pytest_gpu_$TAG.log
echo "== small-batch rows =="
small_row() {  # lib pockets samples [env...]
  local lib=$1 p=$2 s=$3; shift 3
  env "$@" CBGX_LIBRARY=$(pwd)/$lib timeout 90 python bench.py --pockets $p --samples $s --steps 20 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['per_kernel_us_avg_and_launches']
print('$lib', '$*', '$p x $s graphs:', d['value'], {n: v for n, v in k.items() if v[1]})"
}
for rep in 1 2; do
for lib in ab_libs/base.so ab_libs/mw8.so ab_libs/mw2.so; do for cfg in "1 1" "1 10"; do set -- $cfg
  small_row $lib $1 $2 A=1
done; done
small_row ab_libs/base.so 1 1 CBGX_FUSE_ROWS=0
small_row ab_libs/base.so 1 10 CBGX_FUSE_ROWS=0
small_row ab_libs/base.so 4 10 A=1
small_row ab_libs/base.so 4 10 CBGX_FUSE_ROWS=0
done 2>&1 | tee $OUT/small_$TAG.log
echo "== headline (quick) =="
for lib in ab_libs/base.so ab_libs/mw8.so; do CBGX_LIBRARY=$(pwd)/$lib timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['per_kernel_us_avg_and_launches']
print('$lib', 'value', d['value'], {n: v for n, v in k.items() if v[1]})"; done | tee $OUT/ab_fwd_$TAG.log
echo "== timeline of one denoising step =="
for cfg in "1 1" "1 10"; do set -- $cfg
T=$OUT/tl_${TAG}_p$1s$2; mkdir -p $T
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv -d $T/t -o tr -- python $ROOT/bench.py --pockets $1 --samples $2 --steps 3 --warmup 2 --streams 1 --no-cpu-baseline --no-secondary --no-roofline > $T/run.log 2>&1 )
python3 - <<PY
import csv, glob, json
f = glob.glob("$T/t/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0].replace("void ", "").replace("cbgx::", "")[:40] for r in rows]
marks = [i for i, n in enumerate(names) if n.startswith("lig_proximity")]
a, b = marks[-2], marks[-1]
t0 = int(rows[a]["Start_Timestamp"])
seq = [{"k": names[i], "q": rows[i].get("Queue_Id"), "start_us": round((int(rows[i]["Start_Timestamp"]) - t0) / 1e3, 1),
        "us": round((int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])) / 1e3, 1),
        "wgs": int(rows[i].get("Grid_Size_X", rows[i].get("Grid_Size", 0))) // max(int(rows[i].get("Workgroup_Size_X", rows[i].get("Workgroup_Size", 1))), 1)}
       for i in range(a, b)]
json.dump({"step_us": round((int(rows[b]["Start_Timestamp"]) - t0) / 1e3, 1), "launches": len(seq), "seq": seq}, open("$T/step_timeline.json", "w"))
print("$cfg", "step", round((int(rows[b]["Start_Timestamp"]) - t0) / 1e3, 1), "us,", len(seq), "launches, busy", round(sum(s["us"] for s in seq), 1), "us")
PY
rm -rf $T/t
done 2>&1 | tee $OUT/timeline_$TAG.log
