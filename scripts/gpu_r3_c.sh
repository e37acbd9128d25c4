#!/bin/bash
# Round-3 call C: in-kernel query fold -- parity tests, quick sampling/training lines, kernel stats.
TAG=${1:-r03c}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
echo "== pytest =="
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_training.py tests/test_gpu_config_sized.py -q -m gpu -x --durations=4 2>&1 | grep -v "^$" | tail -25 | tee $OUT/pytest_gpu_$TAG.log
echo "== bench (no CPU leg, no secondary) =="
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>&1 | tail -1 | tee $OUT/bench_quick_$TAG.json | python -c "
import json,sys
o=json.loads(sys.stdin.read()); r=o['roofline']
print(o['value'], o['ms_per_step'], r['frac'], r['avg_launch_us']); print({k:(v['us_avg'],v['launches']) for k,v in r['per_kernel'].items() if v['launches']})"
echo "== training =="
timeout 600 python bench.py --workload train --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_train_$TAG.json | python -c "
import json,sys
o=json.loads(sys.stdin.read()); r=o['roofline']
print(o['value'], o['ms_per_step'], r['frac'], r['avg_launch_us']); print({k:(v['us_avg'],v['launches']) for k,v in r['per_kernel'].items() if v['launches']})"
echo "== rocprofv3 kernel stats (sampling) =="
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o bench -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/rocprof_$TAG.log 2>&1 ; tail -1 $OUT/rocprof_$TAG.log | cut -c1-120 )
f=$(find $OUT/prof_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 "$f" | cut -c1-150 && cp "$f" $OUT/kernel_stats_$TAG.csv
find $OUT/prof_$TAG -name "*kernel_trace.csv" -delete; find $OUT/prof_$TAG -name "*.db" -delete 2>/dev/null
