#!/bin/bash
# tools/ab_lib.sh <other-lib.so> [rounds] : the headline bench on the in-tree library and on another build of it, alternating, in one gpurun call
set -u
cd "$GRAFT_REPO_ROOT"
B="python bench.py --steps 80 --warmup 10 --no-cpu-baseline --no-roofline"
for i in $(seq 1 ${2:-2}); do
  timeout -k 10 200 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('in-tree', d['value'])" || exit 1
  SOD_HIP_LIB=$PWD/$1 timeout -k 10 200 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'])" || exit 1
done
