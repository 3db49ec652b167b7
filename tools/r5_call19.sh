#!/bin/bash
# round 5, call 19: gradient-bucket defaults under emulated 8-rank RCCL occupancy (one-GPU rehearsal): wire format x bucket size x bus bandwidth
set -e -o pipefail
O=gpurun_out/r5c19; mkdir -p $O
run() { local name=$1; shift
  timeout -k 10 400 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-roofline --rccl-rehearsal "$@" > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; exit 1; }
  python -c "import json; d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'], d['config'].get('exposed_comm_ms_per_step'), d['config'].get('bucket_mb_each'), d['config'].get('wire_dtype'))" | tee -a $O/table.txt
}
run norccl_equiv
run fp32_b32_300 --rehearsal-occupancy 32:300
run bf16_b32_300 --rehearsal-occupancy 32:300 --wire bf16
run fp32_b16_300 --rehearsal-occupancy 32:300 --bucket-mb 16
run fp32_b64_300 --rehearsal-occupancy 32:300 --bucket-mb 64
run bf16_b16_300 --rehearsal-occupancy 32:300 --bucket-mb 16 --wire bf16
run fp32_b32_150 --rehearsal-occupancy 32:150
run bf16_b32_150 --rehearsal-occupancy 32:150 --wire bf16
run fp32_b32_300b --rehearsal-occupancy 32:300
