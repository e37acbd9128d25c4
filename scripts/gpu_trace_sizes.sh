#!/bin/bash
# per-launch durations of the node-stage and edge kernels grouped by launch size (kernel trace of one bench step, one caller stream)
# Usage (repo root on the GPU box): bash scripts/gpu_trace_sizes.sh [tag]
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/trace_$TAG
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o tr -- python $ROOT/bench.py --steps 1 --warmup 1 --streams 1 --pockets 20 --no-cpu-baseline --no-secondary --no-roofline > $OUT/run.log 2>&1
python3 - <<PY
import csv, glob, collections, json
f = glob.glob("$OUT/t/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print(list(rows[0].keys()))
agg = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:48]
    if not name.startswith("cbgx::"): continue
    grid = int(r.get("Grid_Size_X", r.get("Grid_Size", 0))); wg = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)))
    agg[(name, grid // max(wg, 1))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = []
for (name, wgs), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    out.append({"kernel": name, "workgroups": wgs, "launches": len(v), "us_median": round(v[len(v) // 2], 1), "us_min": round(v[0], 1), "us_max": round(v[-1], 1), "us_total": round(sum(v), 1)})
json.dump(out, open("$OUT/by_size.json", "w"), indent=1)
# the dispatches of the last denoising step in dispatch order (layer by layer: which launches are the full-N ones)
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0].replace("void ", "").replace("cbgx::", "")[:34] for r in rows]
last_knn = max(i for i, n in enumerate(names) if n.startswith("knn_graph"))
seq = [(names[i], round((int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])) / 1e3, 1)) for i in range(last_knn, len(rows))]
json.dump(seq, open("$OUT/last_step_sequence.json", "w"))
print(" ".join(f"{n}:{d}" for n, d in seq if not n.startswith(("build_", "mark_", "__amd"))))
for r in out[:40]: print(r)
PY
rm -rf $OUT/t
