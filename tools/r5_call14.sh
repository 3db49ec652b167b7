#!/bin/bash
# round 5, call 14: per-wave window counter + retuned window policy: parity, the window table with the counter ON, RepPoints at 1/4/8 px
set -o pipefail
mkdir -p gpurun_out/r5c14
timeout -k 10 300 python -m pytest tests/test_gpu_deform_conv.py tests/test_gpu_reppoints.py -x -q -m gpu > gpurun_out/r5c14/tests.log 2>&1 || { tail -30 gpurun_out/r5c14/tests.log; exit 1; }
tail -2 gpurun_out/r5c14/tests.log
timeout -k 10 300 python tools/bench_dcn_bwd_window.py > gpurun_out/r5c14/window.txt 2>&1 || { tail -20 gpurun_out/r5c14/window.txt; exit 1; }
cat gpurun_out/r5c14/window.txt
for px in 0 1 4 8; do
  timeout -k 10 300 python bench.py --arch reppoints --steps 40 --warmup 10 --no-cpu-baseline --no-roofline --reppoints-offset-px $px > gpurun_out/r5c14/rp_$px.json 2> gpurun_out/r5c14/rp_$px.err || { tail -20 gpurun_out/r5c14/rp_$px.err; exit 1; }
  python -c "import json; d=json.loads(open('gpurun_out/r5c14/rp_$px.json').read().strip().splitlines()[-1]); print('reppoints offset px', $px, d['value'], d['ms_per_step'])"
done
