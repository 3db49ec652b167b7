#!/bin/bash
# Single-stream kernel-stats profile of another architecture's step: tools/profile_arch.sh <tag> <arch>  (through gpurun, repo root)
set -eu -o pipefail
tag=$1; arch=$2
# a rehearsal / data-parallel profile needs the rank's hardware-queue count exported HERE: rocprofv3's tool library starts the HIP runtime
# before python does, so bench.py's own in-process setting (utils/comm.py:prepare_rank_env) would come too late
case " $* " in *" --rccl-rehearsal "*) export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-6};; esac
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python3 bench.py --arch $arch ${3:-} --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-host-probe"
SOD_WGRAD_STREAM=0 SOD_TOWER_STREAMS=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_serial -- $B > gpurun_out/${tag}_serial.log 2>&1
python3 tools/summarize_profile.py ${tag}_serial gpurun_out/${tag}_serial "" "" 7
mkdir -p gpurun_out/profiles_${tag} && cp profiles/${tag}* gpurun_out/profiles_${tag}/
rm -rf gpurun_out/${tag}_serial
tail -2 gpurun_out/${tag}_serial.log
