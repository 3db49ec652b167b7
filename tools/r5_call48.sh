#!/bin/bash
# round 5, call 48: runtime switches - kernel arguments in device memory (HIP_FORCE_DEV_KERNARG), alternating 100-step runs
set -e -o pipefail
O=gpurun_out/r5c48; mkdir -p $O
run() { local name=$1; shift
  env "$@" timeout -k 10 400 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --no-host-probe > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; exit 1; }
  python -c "import json; d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'])" | tee -a $O/table.txt
}
for rep in 1 2 3; do
run base_$rep SOD_X=0
run devkernarg_$rep HIP_FORCE_DEV_KERNARG=1
done
run devkernarg0 HIP_FORCE_DEV_KERNARG=0
