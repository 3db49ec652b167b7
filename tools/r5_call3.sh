#!/bin/bash
# round 5, third GPU call: batched side-stream hand-overs A/B, the kernel trace with them, the rest of the GPU suite
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5c3; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_bottleneck.py tests/test_gpu_conv.py -x -q 2>&1 | tail -5 > $O/gputest_a.log; cat $O/gputest_a.log
grep -q "failed" $O/gputest_a.log && exit 1
B="python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline"
run() { local name=$1; shift
  env "$@" timeout -k 10 300 $B 2>$O/ab_$name.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', d['value'])" || echo "$name FAILED"
}
for rep in 1 2 3 4; do
run batch1 SOD_X=0
run batch0 SOD_WGRAD_BATCH=0
done 2>&1 | tee $O/batch_ab.txt
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline > $O/trace.log 2>&1 || echo "trace failed"
python3 tools/classify_gaps.py $O/trace 3 > $O/gaps.txt 2>&1; head -12 $O/gaps.txt
python3 tools/trace_gaps.py $O/trace 3 > $O/occupancy.txt 2>&1; cat $O/occupancy.txt
rm -rf $O/trace
timeout -k 10 1500 python -m pytest tests/test_gpu_parity100.py tests/test_gpu_pointset.py tests/test_gpu_rcnn.py tests/test_gpu_reppoints.py tests/test_gpu_retinanet.py tests/test_gpu_slender_ops.py tests/test_gpu_f32_mode.py -x -q -s 2>&1 | grep -v "^$" | tail -40 > $O/gputest_b.log; cat $O/gputest_b.log
