#!/bin/bash
# round 5, call 21: where the collective path (one-rank RCCL rehearsal) loses 1.2 ms per step against the plain step: kernel traces of both
set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5c21; mkdir -p $O
for mode in plain rehearsal; do
  extra=""; [ $mode = rehearsal ] && extra="--rccl-rehearsal"
  timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$mode -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline $extra > $O/trace_$mode.log 2>&1
  python3 tools/classify_gaps.py $O/trace_$mode 3 > $O/gaps_$mode.txt 2>&1 || true
  python3 tools/trace_gaps.py $O/trace_$mode 3 > $O/occupancy_$mode.txt 2>&1 || true
  python3 tools/kernel_sequence.py $O/trace_$mode > $O/sequence_$mode.txt 2>&1 || true
  head -14 $O/gaps_$mode.txt; tail -4 $O/occupancy_$mode.txt
  find $O/trace_$mode -name "*kernel_trace.csv" -exec cp {} $O/kernel_trace_$mode.csv \;
  rm -rf $O/trace_$mode
done
ls -la $O
