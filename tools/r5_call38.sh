#!/bin/bash
# round 5, call 38: longer runs - does the rate hold over 300 steps (clocks, allocator), do RepPoints' learned offsets stay cheap over 300 steps of its schedule
set -e -o pipefail
O=gpurun_out/r5c38; mkdir -p $O
run() { local name=$1; shift
  timeout -k 10 500 python bench.py --no-cpu-baseline --no-roofline "$@" > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; exit 1; }
  python -c "import json; d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); c=d['config']; print('$name', d['value'], d['ms_per_step'], c.get('final_loss'), c.get('final_grad_norm'))" | tee -a $O/table.txt
}
run fcos_40 --steps 40 --warmup 8
run fcos_300 --steps 300 --warmup 8
run reppoints_40 --arch reppoints --steps 40 --warmup 8
run reppoints_300 --arch reppoints --steps 300 --warmup 8
