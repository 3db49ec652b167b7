#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5c11; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_conv.py tests/test_gpu_losses.py tests/test_gpu_bottleneck.py -x -q 2>&1 | tail -8 > $O/test.log; cat $O/test.log
grep -q "failed\|error" $O/test.log && exit 1
B="python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline"
run() { local name=$1; shift
  env "$@" timeout -k 10 300 $B 2>$O/ab_$name.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', d['value'])" || echo "$name FAILED"
}
for rep in 1 2 3; do
run new SOD_X=0
run old SOD_HIP_LIB=$PWD/gpurun_abl/lib_prev.so
done 2>&1 | tee $O/ab.txt
