#!/bin/bash
# tools/sweep_env.sh "VAR=a VAR=b OTHER=c ..." : the headline bench once per assignment (plus the default first and last), one gpurun call
set -u
cd "$GRAFT_REPO_ROOT"
B="python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-roofline"
run() { env $1 timeout -k 10 200 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'])" || exit 1; }
run "SOD_DEFAULT=1"
for kv in $1; do run "$kv"; done
run "SOD_DEFAULT=1"
