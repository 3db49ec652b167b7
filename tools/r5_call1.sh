#!/bin/bash
# round 5, first GPU call: driver-style baseline, small-op census, classified main-queue gaps, CU-mask sweep
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5c1; mkdir -p $O
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err || { echo "baseline failed"; tail -5 $O/bench_default.err; exit 1; }
python -c "import json; d=json.load(open('$O/bench_default.json')); print('baseline', d['value'], d['ms_per_step'], d['roofline']['backbone_convs'], d['roofline']['head_convs'])"
timeout -k 10 300 python tools/small_ops.py fcos > $O/small_ops.txt 2>&1 || echo "small_ops failed"
timeout -k 10 600 rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d $O/trace -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline > $O/trace.log 2>&1 || echo "trace failed"
python3 tools/classify_gaps.py $O/trace 3 > $O/gaps.txt 2>&1; tail -60 $O/gaps.txt
python3 tools/trace_gaps.py $O/trace 3 > $O/occupancy.txt 2>&1
python3 tools/kernel_sequence.py $O/trace > $O/sequence.txt 2>&1
rm -rf $O/trace
B="python bench.py --steps 60 --warmup 8 --no-cpu-baseline --no-roofline"
run() { # name, env assignments
  local name=$1; shift
  env "$@" timeout -k 10 200 $B 2>$O/ab_$name.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', d['value'])" || echo "$name FAILED"
}
for rep in 1 2; do
run default SOD_X=0
run wgrad64 SOD_CUMASK_WGRAD=0:64
run wgrad64_main192 SOD_CUMASK_WGRAD=0:64 SOD_CUMASK_MAIN=64:256
run wgrad96_main160 SOD_CUMASK_WGRAD=0:96 SOD_CUMASK_MAIN=96:256
run wgrad128_main128 SOD_CUMASK_WGRAD=0:128 SOD_CUMASK_MAIN=128:256
run wgrad32 SOD_CUMASK_WGRAD=0:32
run prefetch64 SOD_CUMASK_PREFETCH=0:64
run prefetch32 SOD_CUMASK_PREFETCH=0:32
run wgrad64_prefetch64_same SOD_CUMASK_WGRAD=0:64 SOD_CUMASK_PREFETCH=0:64
run tower128_main128 SOD_CUMASK_TOWER=0:128 SOD_CUMASK_MAIN=128:256
run tower128 SOD_CUMASK_TOWER=0:128
done 2>&1 | tee $O/cumask.txt
