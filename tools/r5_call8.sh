#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5c8; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_conv.py -x -q -k "persistent_pointwise" 2>&1 | tail -15 > $O/pw_test.log; cat $O/pw_test.log
grep -q "passed" $O/pw_test.log || exit 1
grep -q "failed" $O/pw_test.log && exit 1
B="python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline"
run() { local name=$1; shift
  env "$@" timeout -k 10 300 $B 2>$O/ab_$name.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', d['value'])" || echo "$name FAILED"
}
for rep in 1 2 3; do
run pw1 SOD_X=0
run pw0 SOD_CONV_PW=0
done 2>&1 | tee $O/ab.txt
timeout -k 10 400 python bench.py --steps 24 --warmup 6 --no-cpu-baseline --dump-prof 90 > $O/dump.json 2> $O/dump.txt; grep "1, 1)" $O/dump.txt | head -40
