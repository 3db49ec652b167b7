#!/bin/bash
# tools/ab_env.sh VAR A B [rounds] [extra bench args] : the bench with VAR=A and VAR=B, alternating, in one gpurun call
set -u
cd "$GRAFT_REPO_ROOT"
var=$1; va=$2; vb=$3; rounds=${4:-2}; shift 4 2>/dev/null || shift $#
B="python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-roofline $*"
for i in $(seq 1 $rounds); do
  for v in "$va" "$vb"; do
    env "$var=$v" timeout -k 10 250 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$var=$v', d['value'])" || exit 1
  done
done
