#!/bin/bash
# round 5, call 27: hardware queues - do the communication streams share a HW queue with the compute streams? (GPU_MAX_HW_QUEUES, default 4)
set -e -o pipefail
O=gpurun_out/r5c27; mkdir -p $O
run() { local name=$1; shift
  env "$@" timeout -k 10 400 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-roofline $EXTRA > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; exit 1; }
  python -c "import json; d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); c=d['config']; print('$name', d['value'], d['ms_per_step'], c.get('exposed_comm_ms_per_step'))" | tee -a $O/table.txt
}
EXTRA="--rccl-rehearsal"
run reh_q4 SOD_X=0
run reh_q8 GPU_MAX_HW_QUEUES=8
run reh_q6 GPU_MAX_HW_QUEUES=6
run reh_q16 GPU_MAX_HW_QUEUES=16
EXTRA=""
run plain_q4 SOD_X=0
run plain_q8 GPU_MAX_HW_QUEUES=8
EXTRA="--rccl-rehearsal"
run reh_q4b SOD_X=0
run reh_q8b GPU_MAX_HW_QUEUES=8
