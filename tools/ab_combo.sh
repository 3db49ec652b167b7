#!/bin/bash
# tools/ab_combo.sh "VAR1=a VAR2=b" [rounds] : the headline bench with and without a set of environment assignments, alternating
set -u
cd "$GRAFT_REPO_ROOT"
B="python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline"
for i in $(seq 1 ${2:-3}); do
  timeout -k 10 200 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default', d['value'])" || exit 1
  env $1 timeout -k 10 200 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'])" || exit 1
done
