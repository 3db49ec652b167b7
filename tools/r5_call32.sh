#!/bin/bash
# round 5, call 32: single-stream kernel profiles of the other three configurations with the round-5 library (profiles/r5_<arch>_serial_*)
set -e -o pipefail
bash tools/profile_arch.sh r5_retinanet retinanet
bash tools/profile_arch.sh r5_reppoints reppoints
bash tools/profile_arch.sh r5_rrcnn rrcnn
