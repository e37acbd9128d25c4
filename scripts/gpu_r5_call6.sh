#!/bin/bash
# Round 5, sixth GPU call: forward parity on the fused merge + gate kernel and the word-per-node list kernel; A/B of the merge + gate
# fusion (CBGX_MERGE_GATE=0: the two kernels) on the small rows and the headline; timelines
TAG=${1:-r05f}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
echo "== pytest -m gpu (forward / sampler files) =="
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config_sized.py tests/test_gpu_range.py -q -m gpu -x -p no:faulthandler 2>&1 | grep -v "^$" | tail -6 | cut -c1-300 | tee $OUT/pytest_gpu_$TAG.log
small_row() {  # pockets samples env
  local p=$1 s=$2; shift 2
  env "$@" timeout 90 python bench.py --pockets $p --samples $s --steps 20 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['per_kernel_us_avg_and_launches']
print('$*', '$p x $s graphs:', d['value'], {n: v for n, v in k.items() if v[1]})"
}
echo "== small-batch rows =="
for rep in 1 2; do for mg in 1 0; do small_row 1 1 CBGX_MERGE_GATE=$mg; small_row 1 10 CBGX_MERGE_GATE=$mg; small_row 4 10 CBGX_MERGE_GATE=$mg; done; done 2>&1 | tee $OUT/small_$TAG.log
echo "== headline A/B =="
for rep in 1 2; do for mg in 1 0; do CBGX_MERGE_GATE=$mg timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['per_kernel_us_avg_and_launches']
print('merge_gate=$mg', 'value', d['value'], {n: v for n, v in k.items() if v[1]})"; done; done | tee $OUT/ab_fwd_$TAG.log
echo "== timeline of one denoising step =="
for cfg in "1 1" "1 10"; do set -- $cfg
T=$OUT/tl_${TAG}_p$1s$2; mkdir -p $T
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv -d $T/t -o tr -- python $ROOT/bench.py --pockets $1 --samples $2 --steps 3 --warmup 2 --streams 1 --no-cpu-baseline --no-secondary --no-roofline > $T/run.log 2>&1 )
python3 - <<PY
import csv, glob, json, collections
f = glob.glob("$T/t/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0].replace("void ", "").replace("cbgx::", "")[:40] for r in rows]
marks = [i for i, n in enumerate(names) if n.startswith("graph_cache_begin")]
a, b = marks[-2], marks[-1]
t0 = int(rows[a]["Start_Timestamp"])
seq = [{"k": names[i], "start_us": round((int(rows[i]["Start_Timestamp"]) - t0) / 1e3, 1),
        "us": round((int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])) / 1e3, 1),
        "wgs": int(rows[i].get("Grid_Size_X", rows[i].get("Grid_Size", 0))) // max(int(rows[i].get("Workgroup_Size_X", rows[i].get("Workgroup_Size", 1))), 1)}
       for i in range(a, b)]
json.dump({"step_us": round((int(rows[b]["Start_Timestamp"]) - t0) / 1e3, 1), "launches": len(seq), "seq": seq}, open("$T/step_timeline.json", "w"))
agg = collections.defaultdict(list)
for s in seq: agg[s["k"]].append(s["us"])
print("$cfg", "step", round((int(rows[b]["Start_Timestamp"]) - t0) / 1e3, 1), "us,", len(seq), "launches", {k: (len(v), round(sum(v) / len(v), 1)) for k, v in agg.items()})
PY
rm -rf $T/t
done 2>&1 | tee $OUT/timeline_$TAG.log
