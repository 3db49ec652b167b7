#!/bin/bash
# round 5, fourth GPU call: whole GPU suite on the pruned library, then A/B of the adjacent hand-over batching
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5c4; mkdir -p $O
timeout -k 10 1700 python -m pytest tests -m gpu -x -q -s 2>&1 | grep -v "^$" | tail -40 > $O/gputest.log; tail -12 $O/gputest.log
B="python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline"
for rep in 1 2 3; do
timeout -k 10 300 $B 2>$O/ab.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default', d['value'])" || echo FAILED
done | tee $O/ab.txt
