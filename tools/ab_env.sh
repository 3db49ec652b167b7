#!/bin/bash
# tools/ab_env.sh VAR A B [rounds] : the headline bench with VAR=A and VAR=B, alternating, in one gpurun call
set -u
cd "$GRAFT_REPO_ROOT"
B="python bench.py --steps 80 --warmup 10 --no-cpu-baseline --no-roofline"
for i in $(seq 1 ${4:-2}); do
  for v in "$2" "$3"; do
    env "$1=$v" timeout -k 10 200 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1=$v', d['value'])" || exit 1
  done
done
