#!/bin/bash
# tools/ab_combo.sh "VAR1=a VAR2=b" [rounds] [bench args] : the bench with and without a set of environment assignments, alternating 100-step runs
set -u
cd "$GRAFT_REPO_ROOT"
envs=$1; rounds=${2:-3}; shift 2 2>/dev/null || shift $#
B="python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline $*"
for i in $(seq 1 $rounds); do
  timeout -k 10 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default', d['value'])" || exit 1
  env $envs timeout -k 10 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$envs', d['value'])" || exit 1
done
