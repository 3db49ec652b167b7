#!/bin/bash
# Single-stream kernel-stats profile of another architecture's step: tools/profile_arch.sh <tag> <arch>  (through gpurun, repo root)
set -eu -o pipefail
tag=$1; arch=$2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python3 bench.py --arch $arch ${3:-} --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-host-probe"
SOD_WGRAD_STREAM=0 SOD_TOWER_STREAMS=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_serial -- $B > gpurun_out/${tag}_serial.log 2>&1
python3 tools/summarize_profile.py ${tag}_serial gpurun_out/${tag}_serial "" "" 7
mkdir -p gpurun_out/profiles_${tag} && cp profiles/${tag}* gpurun_out/profiles_${tag}/
rm -rf gpurun_out/${tag}_serial
tail -2 gpurun_out/${tag}_serial.log
