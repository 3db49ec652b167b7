#!/bin/bash
# round 5, fifth GPU call: the GPU suite, then prefetch placement A/B
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5c5; mkdir -p $O
B="python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline"
run() { local name=$1; shift
  env "$@" timeout -k 10 300 $B 2>$O/ab_$name.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', d['value'])" || echo "$name FAILED"
}
for rep in 1 2 3; do
run pre_fwd SOD_X=0
run pre_bwd SOD_PREFETCH_AT=bwd
done 2>&1 | tee $O/ab.txt
SOD_PREFETCH_AT=bwd timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline > $O/trace.log 2>&1 || echo "trace failed"
python3 tools/trace_gaps.py $O/trace 3 > $O/occupancy_bwd.txt 2>&1; cat $O/occupancy_bwd.txt
python3 tools/kernel_sequence.py $O/trace > $O/sequence_bwd.txt 2>&1
rm -rf $O/trace
timeout -k 10 1700 python -m pytest tests -m gpu -x -q -s 2>&1 | grep -v "^$" | tail -40 > $O/gputest.log; tail -14 $O/gputest.log
