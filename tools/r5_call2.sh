#!/bin/bash
# round 5, second GPU call: 256-kernel tail threshold A/B, scheduler sanity, kernel-trace-only occupancy, then the GPU test suite
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5c2; mkdir -p $O
B="python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline"
run() { local name=$1; shift
  env "$@" timeout -k 10 300 $B 2>$O/ab_$name.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', d['value'], d['config'].get('host_enqueue_ms_per_step'), d['config']['device'].get('sclk_active'))" || echo "$name FAILED"
}
for rep in 1 2 3; do
run tail50 SOD_X=0
run tail30 SOD_CONV256_TAIL_MAX=30
run tail12 SOD_CONV256_TAIL_MAX=12
run tail5 SOD_CONV256_TAIL_MAX=5
done 2>&1 | tee $O/tail_ab.txt
timeout -k 10 300 $B --constant-lr 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('constant-lr', d['value'])" | tee -a $O/tail_ab.txt
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline > $O/trace.log 2>&1 || echo "trace failed"
python3 tools/classify_gaps.py $O/trace 3 > $O/gaps_kernel_trace_only.txt 2>&1; head -8 $O/gaps_kernel_trace_only.txt
python3 tools/trace_gaps.py $O/trace 3 > $O/occupancy.txt 2>&1; cat $O/occupancy.txt
rm -rf $O/trace
timeout -k 10 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/gputest.log; cat $O/gputest.log
