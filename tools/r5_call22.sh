#!/bin/bash
set -e -o pipefail
O=gpurun_out/r5c22; mkdir -p $O
run() { local name=$1; shift
  env "$@" timeout -k 10 400 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-roofline --rccl-rehearsal > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; exit 1; }
  python -c "import json; d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); c=d['config']; print('$name', d['value'], d['ms_per_step'], c.get('exposed_comm_ms_per_step'))" | tee -a $O/table.txt
}
run c10d SOD_X=0
run skip SOD_DEBUG_SKIP_ALLREDUCE=1
run avoid_record TORCH_NCCL_AVOID_RECORD_STREAMS=1
run c10d_b SOD_X=0
run skip_b SOD_DEBUG_SKIP_ALLREDUCE=1
