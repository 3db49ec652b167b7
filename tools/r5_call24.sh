#!/bin/bash
# round 5, call 24: is the host ahead of the device?  kernel trace + HIP runtime trace of the plain step and of the rehearsal step
set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5c24; mkdir -p $O
for mode in plain rehearsal; do
  extra=""; [ $mode = rehearsal ] && extra="--rccl-rehearsal"
  timeout -k 10 600 rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d $O/trace_$mode -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline $extra > $O/trace_$mode.log 2>&1
  python3 tools/classify_gaps.py $O/trace_$mode 3 > $O/gaps_$mode.txt 2>&1 || true
  head -16 $O/gaps_$mode.txt
  tail -2 $O/trace_$mode.log
  find $O/trace_$mode -name "*kernel_trace.csv" -exec cp {} $O/kernel_trace_$mode.csv \;
  find $O/trace_$mode -name "*hip_api_trace.csv" -exec cp {} $O/hip_api_trace_$mode.csv \;
  rm -rf $O/trace_$mode
done
ls -la $O
