#!/bin/bash
# round 5, call 15: final profile of the headline step (profiles/r5b_*) + a driver-style bench run
set -e -o pipefail
bash tools/profile_step.sh ${1:-r5b}
mkdir -p gpurun_out/r5c15${1:+_$1}
timeout -k 10 900 python bench.py > gpurun_out/r5c15${1:+_$1}/bench.json 2> gpurun_out/r5c15${1:+_$1}/bench.err
python -c "import json; d=json.loads(open('gpurun_out/r5c15${1:+_$1}/bench.json').read().strip().splitlines()[-1]); print('driver-style', d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline'])"
