cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2 3; do for v in low default; do
CBGX_AUX_PRIORITY=$v python bench.py --workload train --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('train [CBGX_AUX_PRIORITY=$v]', d['value'])"; done; done
for v in low default; do
CBGX_AUX_PRIORITY=$v python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('headline [CBGX_AUX_PRIORITY=$v]', d['value'])"; done
