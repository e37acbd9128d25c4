#!/bin/bash
# kernel trace of a few training steps with the queue of every dispatch: what is on the critical path (the caller's stream) and what
# runs beside it (the library's auxiliary stream).  Output: gpurun_out/trace_train_<tag>/{last_step.tsv, critical.txt}
TAG=${1:-r06}; MODEL=${2:-targetdiff}      # model class: targetdiff | diffbp | diffsbdd
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/trace_train_$TAG
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o tr -- python $ROOT/bench.py --workload train --model $MODEL --steps 3 --warmup 3 --no-cpu-baseline --no-roofline > $OUT/bench.log 2>&1
CSV=$(find $OUT -name "*kernel_trace.csv" | head -1)
python - $CSV $OUT <<'P'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# one step = from one knn_graph_reg_kernel to the next; keep the last complete one
starts = [k for k, r in enumerate(rows) if "knn_graph_reg_kernel" in r["Kernel_Name"]]
a, b = starts[-2], starts[-1]
keep = rows[a:b]
t0 = int(keep[0]["Start_Timestamp"])
qs = sorted({r["Queue_Id"] for r in keep}, key=lambda q: -sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in keep if r["Queue_Id"] == q))
qn = {q: k for k, q in enumerate(qs)}
def short(nm):
    nm = nm.replace("void ", "").replace("cbgx::", "")
    return nm.split("(")[0][:48]
with open(sys.argv[2] + "/last_step.tsv", "w") as f:
    for r in keep:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        f.write(f"{(s - t0) / 1e3:10.1f}\t{(e - s) / 1e3:8.1f}\tq{qn[r['Queue_Id']]}\t{r.get('Grid_Size_X', r.get('Grid_Size', ''))}\t{short(r['Kernel_Name'])}\n")
with open(sys.argv[2] + "/critical.txt", "w") as f:
    span = (int(rows[b]["Start_Timestamp"]) - t0) / 1e3
    f.write(f"step span {span:.1f} us, {len(keep)} dispatches, queues {len(qs)}\n")
    for q in qs:
        mine = [r for r in keep if r["Queue_Id"] == q]
        busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in mine) / 1e3
        f.write(f"\nq{qn[q]}: {len(mine)} dispatches, busy {busy:.1f} us ({100 * busy / span:.1f} % of the step)\n")
        agg = collections.OrderedDict()
        for r in mine:
            k = short(r["Kernel_Name"])
            c, t = agg.get(k, (0, 0.0))
            agg[k] = (c + 1, t + (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            f.write(f"   {t:9.1f} us  {c:4d} x {t / c:8.1f}  {k}\n")
        # idle gaps of this queue larger than 10 us
        gaps = []
        for u, v in zip(mine, mine[1:]):
            g = (int(v["Start_Timestamp"]) - int(u["End_Timestamp"])) / 1e3
            if g > 10: gaps.append((g, short(u["Kernel_Name"]), short(v["Kernel_Name"])))
        f.write(f"   gaps > 10 us: {len(gaps)}, total {sum(g for g, _, _ in gaps):.1f} us\n")
        for g, u, v in sorted(gaps, reverse=True)[:12]:
            f.write(f"      {g:8.1f} us between {u} and {v}\n")
print(open(sys.argv[2] + "/critical.txt").read())
P
find $OUT -name "*.csv" -size +20M -delete
ls -la $OUT
