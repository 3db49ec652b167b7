#!/bin/bash
# round 5, call 30: full GPU suite after the DeformConv / arena / bench changes, then smoke() and a 2-rank shared-GPU gloo bench
set -e -o pipefail
O=gpurun_out/r5c30; mkdir -p $O
timeout -k 10 1500 python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1 || { tail -40 $O/gputest.log; exit 1; }
tail -3 $O/gputest.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
