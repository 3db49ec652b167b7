#!/bin/bash
# round 5, call 28: GPU_MAX_HW_QUEUES sweep, rehearsal and plain
set -e -o pipefail
O=gpurun_out/r5c28; mkdir -p $O
run() { local name=$1; shift
  env "$@" timeout -k 10 400 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-roofline $EXTRA > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; exit 1; }
  python -c "import json; d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); c=d['config']; print('$name', d['value'], d['ms_per_step'], c.get('exposed_comm_ms_per_step'))" | tee -a $O/table.txt
}
EXTRA="--rccl-rehearsal"
run reh_q5 GPU_MAX_HW_QUEUES=5
run reh_q6 GPU_MAX_HW_QUEUES=6
run reh_q7 GPU_MAX_HW_QUEUES=7
run reh_q6_occ GPU_MAX_HW_QUEUES=6
EXTRA="--rccl-rehearsal --rehearsal-occupancy 32:300"
run reh_q6_occ32 GPU_MAX_HW_QUEUES=6
run reh_q4_occ32 SOD_X=0
EXTRA=""
run plain_q5 GPU_MAX_HW_QUEUES=5
run plain_q6 GPU_MAX_HW_QUEUES=6
run plain_q7 GPU_MAX_HW_QUEUES=7
run plain_q2 GPU_MAX_HW_QUEUES=2
run plain_q4 SOD_X=0
