#!/bin/bash
# round 5, call 29: GPU_MAX_HW_QUEUES set in-process by bench.py (before the first HIP call) - does the runtime see it?  + other architectures
set -e -o pipefail
O=gpurun_out/r5c29; mkdir -p $O
run() { local name=$1; shift
  timeout -k 10 400 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-roofline --rccl-rehearsal "$@" > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; exit 1; }
  python -c "import json; d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); c=d['config']; print('$name', d['value'], d['ms_per_step'], c.get('exposed_comm_ms_per_step'), c.get('hw_queues'))" | tee -a $O/table.txt
}
run fcos_default
run fcos_q4 --hw-queues 4
run retinanet_default --arch retinanet
run retinanet_q4 --arch retinanet --hw-queues 4
run reppoints_default --arch reppoints
run reppoints_q4 --arch reppoints --hw-queues 4
run rrcnn_default --arch rrcnn
run rrcnn_q4 --arch rrcnn --hw-queues 4
run retinanet_q5 --arch retinanet --hw-queues 5
run reppoints_q5 --arch reppoints --hw-queues 5
run rrcnn_q5 --arch rrcnn --hw-queues 5
