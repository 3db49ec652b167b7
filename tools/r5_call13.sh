#!/bin/bash
# round 5, call 13: wave-cooperative out-of-window atomics of the fused DeformConv backward - parity, the window table, RepPoints bench
set -o pipefail
mkdir -p gpurun_out/r5c13
timeout -k 10 300 python -m pytest tests/test_gpu_deform_conv.py tests/test_gpu_reppoints.py -x -q -m gpu > gpurun_out/r5c13/tests.log 2>&1 || { tail -30 gpurun_out/r5c13/tests.log; exit 1; }
tail -2 gpurun_out/r5c13/tests.log
timeout -k 10 300 python tools/bench_dcn_bwd_window.py > gpurun_out/r5c13/window.txt 2>&1 || { tail -20 gpurun_out/r5c13/window.txt; exit 1; }
cat gpurun_out/r5c13/window.txt
timeout -k 10 300 python bench.py --arch reppoints --steps 60 --warmup 15 > gpurun_out/r5c13/reppoints.json 2> gpurun_out/r5c13/reppoints.err || { tail -20 gpurun_out/r5c13/reppoints.err; exit 1; }
python -c "import json; d=json.loads(open('gpurun_out/r5c13/reppoints.json').read().strip().splitlines()[-1]); print('reppoints', d['value'])"
