#!/bin/bash
# round 5, call 52: the FCOS loss node returns three outputs (no SelectBackward launches at the forward / backward junction): parity + A/B
set -e -o pipefail
O=gpurun_out/r5c52; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_f32_mode.py tests/test_gpu_losses.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
run() { local name=$1; shift
  timeout -k 10 400 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --no-host-probe "$@" > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; exit 1; }
  python -c "import json; d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'], d['config']['final_loss'])" | tee -a $O/table.txt
}
run new_1; run new_2; run new_3
