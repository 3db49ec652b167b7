#!/bin/bash
# other configurations at the reference's LR schedule (warm-up), and the DeformConv-backward window table
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5c7; mkdir -p $O
run() { local name=$1; shift
  timeout -k 10 400 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline "$@" 2>$O/$name.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', d['value'], d['config']['base_lr'], d['config']['final_loss'])" || { echo "$name FAILED"; tail -3 $O/$name.err; }
}
run fcos
run retinanet --arch retinanet
run reppoints --arch reppoints
run reppoints_const --arch reppoints --constant-lr
run rrcnn50 --arch rrcnn
run rrcnn101 --arch rrcnn --depth 101
run resnext50 --resnext
timeout -k 10 400 python tools/bench_dcn_bwd_window.py > $O/dcn_window.txt 2>&1; tail -25 $O/dcn_window.txt
