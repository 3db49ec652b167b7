#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5c12; mkdir -p $O
timeout -k 10 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > $O/gputest.log; cat $O/gputest.log
grep -q "failed\|error" $O/gputest.log && exit 1
run() { local name=$1; shift
  timeout -k 10 400 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline "$@" 2>$O/$name.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', d['value'], d['config']['base_lr'], d['config']['final_loss'])" || { echo "$name FAILED"; tail -3 $O/$name.err; }
}
run fcos
run retinanet --arch retinanet
run reppoints --arch reppoints
run rrcnn50 --arch rrcnn
run rrcnn101 --arch rrcnn --depth 101
run resnext50 --resnext
run fcos_r101 --depth 101
