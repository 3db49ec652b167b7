#!/bin/bash
# round 5, call 17: what the step loses while RCCL-like channel kernels hold CUs (one-GPU rehearsal with emulated occupancy)
set -e -o pipefail
O=gpurun_out/r5c17; mkdir -p $O
run() { local name=$1; shift
  timeout -k 10 400 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-roofline --rccl-rehearsal "$@" > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; exit 1; }
  python -c "import json; d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'], d['config'].get('exposed_comm_ms_per_step'))"
}
run none
run occ16_300 --rehearsal-occupancy 16:300
run occ32_300 --rehearsal-occupancy 32:300
run occ64_300 --rehearsal-occupancy 64:300
run occ32_100 --rehearsal-occupancy 32:100
run none2
