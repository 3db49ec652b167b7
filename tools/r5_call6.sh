#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5c6; mkdir -p $O
timeout -k 10 400 python bench.py --steps 24 --warmup 6 --no-cpu-baseline --dump-prof 90 > $O/dump.json 2> $O/dump.txt; tail -95 $O/dump.txt
