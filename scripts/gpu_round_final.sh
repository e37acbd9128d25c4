#!/bin/bash
# Evidence call of a round (one gpurun): all GPU tests (the driver's command with -x), smoke, the driver's default bench command
# (headline + secondary block + the CPU legs), rocprofv3 kernel stats of the sampling and the training command, PMC passes of the dominant
# forward kernel (instruction mix / waits, HBM traffic in separate FETCH_SIZE / WRITE_SIZE passes) and of the x2h edge backward.
# Usage (repo root on the GPU box): bash scripts/gpu_round_final.sh [tag]
TAG=${1:-r06z}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
echo "== host =="; rocminfo | grep -m2 -E "gfx|Compute Unit"; nproc; free -g | sed -n 2p
echo "== pytest -m gpu =="
timeout 1800 python -m pytest tests/ -x -q -m gpu --durations=6 -s -p no:faulthandler 2>&1 | grep -v "^$" | grep -E "passed|failed|Error|error|FAILED|roll-out|ReLU flip|worst relative|pocket frame|linker-256|^\| |^[0-9.]+s " | tail -60 | cut -c1-300 | tee $OUT/pytest_gpu_$TAG.log
echo "== smoke =="
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $OUT/smoke_$TAG.log
echo "== bench: the driver's command =="
( time timeout 1200 python bench.py ) 2>&1 | tail -5 | tee $OUT/bench_$TAG.json | cut -c1-300
grep -m1 '^{' $OUT/bench_$TAG.json | wc -c
echo "== bench: one stream (round-2 schedule) for comparison =="
timeout 600 python bench.py --streams 1 --no-cpu-baseline --no-secondary 2>&1 | tail -1 | tee $OUT/bench_1stream_$TAG.json | cut -c1-200
echo "== rocprofv3 kernel stats: sampling (one stream: per-kernel times without overlap), training =="
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o bench -- python $ROOT/bench.py --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-secondary > $OUT/rocprof_$TAG.log 2>&1 ; tail -1 $OUT/rocprof_$TAG.log | cut -c1-200 )
f=$(find $OUT/prof_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -10 "$f" | cut -c1-160 && cp "$f" $OUT/kernel_stats_$TAG.csv
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train_$TAG -o tr -- python $ROOT/bench.py --workload train --steps 4 --warmup 2 --no-cpu-baseline --no-roofline > $OUT/rocprof_train_$TAG.log 2>&1 ; tail -1 $OUT/rocprof_train_$TAG.log | cut -c1-200 )
f=$(find $OUT/prof_train_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f" | cut -c1-160 && cp "$f" $OUT/kernel_stats_train_$TAG.csv
find $OUT/prof_$TAG $OUT/prof_train_$TAG -name "*kernel_trace.csv" -delete; find $OUT/prof_$TAG $OUT/prof_train_$TAG -name "*.db" -delete 2>/dev/null
echo "== PMC: forward x2h (instruction mix, traffic) =="
bash scripts/gpu_pmc_x2h.sh $TAG 2>&1 | tail -30
bash scripts/gpu_pmc_traffic.sh $TAG 2>&1 | tail -12
echo "== PMC: x2h edge backward (training bench) =="
bash scripts/gpu_pmc_bwd.sh $TAG 2>&1 | tail -30
du -sh $OUT | tail -1
echo "== step timelines of the small inputs (kernel trace of one denoising step) =="
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
du -sh $OUT | tail -1
# optional tail: the headline once more on ab_libs/base.so (the commit before the build's last change), when that library travels
if [ -f $ROOT/ab_libs/base.so ]; then
echo "== headline on ab_libs/base.so, then on the tree again =="
for L in $ROOT/ab_libs/base.so $ROOT/cbgbench_amd/lib/libcbgx.so; do
CBGX_LIBRARY=$L timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['per_kernel_us_avg_and_launches']
print('$L'.split('/')[-1], d['value'], {n: v for n, v in k.items() if v[1]})"
done | tee $OUT/ab_fwd_$TAG.log
fi
echo "== training step: kernel trace with queue ids (critical path of the caller's stream) =="
bash scripts/gpu_trace_train.sh $TAG 2>&1 | head -12
cp $OUT/trace_train_$TAG/critical.txt $OUT/trace_train_critical_$TAG.txt 2>/dev/null
echo "== training line under the per-call knobs (edge rows, schedule) =="
for set in "" "CBGX_BX_EDGE_ROWS=1" "CBGX_TRAIN_FWD_OVERLAP=0 CBGX_TRAIN_ZERO_ROWS=0" "CBGX_TRAIN_OVERLAP=0"; do for rep in 1 2; do
env $set python bench.py --workload train --steps 20 --warmup 4 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']['per_kernel_us_avg_and_launches']
print('[$set] train', d['value'], 'graph-steps/s; x2h backward', d['roofline']['avg_launch_us'], 'us; gather', r.get('edge_rows_reduce'))"; done; done | tee $OUT/train_knobs_$TAG.log
