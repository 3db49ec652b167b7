#!/bin/bash
set -e -o pipefail
O=gpurun_out/r5c26; mkdir -p $O
run() { local name=$1; shift
  timeout -k 10 400 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline "$@" > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; exit 1; }
  python -c "import json; d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); c=d['config']; print('$name', d['value'], d['ms_per_step'], 'host unblocked', c.get('host_ms_per_step_unblocked'), 'enqueue', c.get('host_enqueue_ms_per_step'), 'lead', c.get('host_lead_ms_min_median'), c.get('exposed_comm_ms_per_step'))" | tee -a $O/table.txt
}
run plain
run rehearsal --rccl-rehearsal
run plain2
