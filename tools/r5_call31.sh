#!/bin/bash
set -e -o pipefail
O=gpurun_out/r5c31; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_slender_ops.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
timeout -k 10 300 python tools/bench_f3.py > $O/bench_f3.txt 2>&1 || { tail -20 $O/bench_f3.txt; exit 1; }
cat $O/bench_f3.txt
