#!/bin/bash
# round 5, call 18: tapered last bucket under emulated RCCL occupancy (one-GPU rehearsal) + the 2-rank gloo DDP tests' GPU twin
set -e -o pipefail
O=gpurun_out/r5c18; mkdir -p $O
run() { local name=$1; shift
  timeout -k 10 400 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-roofline --rccl-rehearsal "$@" > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; exit 1; }
  python -c "import json; d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'], d['config'].get('exposed_comm_ms_per_step'), d['config'].get('bucket_mb_each'))"
}
run tail0_300 --rehearsal-occupancy 32:300 --bucket-tail-mb 0
run tail6_300 --rehearsal-occupancy 32:300
run tail0_100 --rehearsal-occupancy 32:100 --bucket-tail-mb 0
run tail6_100 --rehearsal-occupancy 32:100
run tail0_300b --rehearsal-occupancy 32:300 --bucket-tail-mb 0
run tail6_300b --rehearsal-occupancy 32:300
run tail2_300 --rehearsal-occupancy 32:300 --bucket-tail-mb 2
