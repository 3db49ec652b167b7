#!/bin/bash
# Profile of another architecture's step: tools/profile_arch.sh <tag> <arch> [extra bench args]  (through gpurun, repo root)
#   pass 1: single-stream kernel stats (true per-kernel durations)     -> profiles/<tag>_serial_kernel_stats.csv
#   pass 2/3: FETCH_SIZE / WRITE_SIZE PMC passes of the default run (each alone with --kernel-trace, as the pool requires)
#                                                                        -> profiles/<tag>_pmc.json  (<tag> = rN_<arch>: bench.py finds it by arch)
set -eu -o pipefail
tag=$1; arch=$2
# a rehearsal / data-parallel profile needs the rank's hardware-queue count exported HERE: rocprofv3's tool library starts the HIP runtime
# before python does, so bench.py's own in-process setting (utils/comm.py:prepare_rank_env) would come too late
case " $* " in *" --rccl-rehearsal "*) export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-6};; esac
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python3 bench.py --arch $arch ${3:-} --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-host-probe"
SOD_WGRAD_STREAM=0 SOD_TOWER_STREAMS=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_serial -- $B > gpurun_out/${tag}_serial.log 2>&1
python3 tools/summarize_profile.py ${tag}_serial gpurun_out/${tag}_serial "" "" 7
if [ "${SOD_PROFILE_PMC:-1}" = "1" ]; then
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -- $B > gpurun_out/${tag}_stats.log 2>&1
  timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_fetch -- $B > gpurun_out/${tag}_fetch.log 2>&1
  timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_write -- $B > gpurun_out/${tag}_write.log 2>&1
  # step time of the un-profiled command, for the step-level HBM rate in <tag>_pmc.json
  export SOD_PROFILE_MS_PER_STEP=$(timeout 300 $B 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
  python3 tools/summarize_profile.py ${tag} gpurun_out/${tag}_stats gpurun_out/${tag}_fetch gpurun_out/${tag}_write 7
  rm -rf gpurun_out/${tag}_stats gpurun_out/${tag}_fetch gpurun_out/${tag}_write
fi
mkdir -p gpurun_out/profiles_${tag} && cp profiles/${tag}* gpurun_out/profiles_${tag}/
rm -rf gpurun_out/${tag}_serial
tail -2 gpurun_out/${tag}_serial.log
