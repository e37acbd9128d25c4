#!/bin/bash
# Round 5, first GPU call: (1) first evaluation of the two edge-kernel variants round 4 left unrun (-DCBGX_EDGE_DYN=2,
# -DCBGX_EDGE_SMALL_W4=1): parity on the combined build, small-batch rows / headline / training line per library;
# (2) the kernel timeline of ONE denoising step at 1 and 10 graphs (start offsets, durations, queue) with and without libcbgx's
# auxiliary stream -- what the latency path has to remove.
# Before the call, in the build container:
#   rm -f ab_libs/*.so; python scripts/build_variant.py base; python scripts/build_variant.py dyn2 -DCBGX_EDGE_DYN=2
#   python scripts/build_variant.py w4 -DCBGX_EDGE_SMALL_W4=1; python scripts/build_variant.py dynw4 -DCBGX_EDGE_DYN=2 -DCBGX_EDGE_SMALL_W4=1
TAG=${1:-r05a}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
echo "== parity on the combined variant =="
CBGX_LIBRARY=$(pwd)/ab_libs/dynw4.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_training.py -m gpu -q -x \
  -k "not rollout_200" -p no:faulthandler 2>&1 | grep -E "passed|failed|Error|^E |FAILED" | cut -c1-400 | head -20 | tee $OUT/pytest_dyn_$TAG.log
echo "== small-batch rows =="
small_row() {  # lib pockets samples [env...]
  local lib=$1 p=$2 s=$3; shift 3
  env "$@" CBGX_LIBRARY=$(pwd)/$lib timeout 90 python bench.py --pockets $p --samples $s --steps 20 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['per_kernel_us_avg_and_launches']
print('$lib', '$*', '$p x $s graphs:', d['value'], {n: v for n, v in k.items() if v[1]})"
}
for lib in ab_libs/base.so ab_libs/dyn2.so ab_libs/w4.so ab_libs/dynw4.so; do for cfg in "1 1" "1 10"; do set -- $cfg
  small_row $lib $1 $2 A=1
done; done 2>&1 | tee $OUT/small_dyn_$TAG.log
small_row ab_libs/base.so 1 1 CBGX_OVERLAP=0 | tee -a $OUT/small_dyn_$TAG.log
small_row ab_libs/base.so 1 10 CBGX_OVERLAP=0 | tee -a $OUT/small_dyn_$TAG.log
echo "== training line, headline: base vs dyn2 =="
for lib in ab_libs/base.so ab_libs/dyn2.so; do CBGX_LIBRARY=$(pwd)/$lib timeout 200 python bench.py --workload train --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lib train', d['value'], 'graph-steps/s; x2h backward', d['roofline']['avg_launch_us'], 'us')"; done | tee $OUT/ab_train_dyn_$TAG.log
for lib in ab_libs/base.so ab_libs/dyn2.so; do CBGX_LIBRARY=$(pwd)/$lib timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['per_kernel_us_avg_and_launches']
print('$lib', 'value', d['value'], {n: v for n, v in k.items() if v[1]})"; done | tee $OUT/ab_fwd_dyn_$TAG.log
echo "== timeline of one denoising step =="
for cfg in "1 1 1" "1 10 1" "1 1 0" "1 10 0"; do set -- $cfg
T=$OUT/tl_${TAG}_p$1s$2_ov$3; mkdir -p $T
( cd /tmp && CBGX_OVERLAP=$3 timeout 200 rocprofv3 --kernel-trace --output-format csv -d $T/t -o tr -- python $ROOT/bench.py --pockets $1 --samples $2 --steps 3 --warmup 2 --streams 1 --no-cpu-baseline --no-secondary --no-roofline > $T/run.log 2>&1 )
python3 - <<PY
import csv, glob, json
f = glob.glob("$T/t/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0].replace("void ", "").replace("cbgx::", "")[:40] for r in rows]
# the last complete denoising step: from the second-to-last lig_proximity launch to the last one
marks = [i for i, n in enumerate(names) if n.startswith("lig_proximity")]
a, b = marks[-2], marks[-1]
t0 = int(rows[a]["Start_Timestamp"])
seq = [{"k": names[i], "q": rows[i].get("Queue_Id"), "start_us": round((int(rows[i]["Start_Timestamp"]) - t0) / 1e3, 1),
        "us": round((int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])) / 1e3, 1),
        "wgs": int(rows[i].get("Grid_Size_X", rows[i].get("Grid_Size", 0))) // max(int(rows[i].get("Workgroup_Size_X", rows[i].get("Workgroup_Size", 1))), 1)}
       for i in range(a, b)]
json.dump({"columns": list(rows[0].keys()), "step_us": round((int(rows[b]["Start_Timestamp"]) - t0) / 1e3, 1), "launches": len(seq), "seq": seq},
          open("$T/step_timeline.json", "w"))
print("$cfg", "step", round((int(rows[b]["Start_Timestamp"]) - t0) / 1e3, 1), "us,", len(seq), "launches, busy",
      round(sum(s["us"] for s in seq), 1), "us")
PY
rm -rf $T/t
done 2>&1 | tee $OUT/timeline_$TAG.log
du -sh $OUT | tail -1
