#!/bin/bash
# tools/profile_gaps.sh <tag> [extra bench args] : kernel trace of the default multi-stream step -> idle / overlap summary
set -u
tag=$1; shift
# a rehearsal / data-parallel profile needs the rank's hardware-queue count exported HERE: rocprofv3's tool library starts the HIP runtime
# before python does, so bench.py's own in-process setting (utils/comm.py:prepare_rank_env) would come too late
case " $* " in *" --rccl-rehearsal "*) export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-6};; esac
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${tag}_trace -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-host-probe "$@" > gpurun_out/${tag}_trace.log 2>&1
python3 tools/trace_gaps.py gpurun_out/${tag}_trace 4 | tee gpurun_out/${tag}_gaps.txt
rm -rf gpurun_out/${tag}_trace
