#!/bin/bash
# round 5, call 34: instruction mix of the fused DeformConv backward (in-window offsets) from the SQ counters
set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5c34; mkdir -p $O
cat > $O/one.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd())
from slenderobjdet_amd.layers import functional as HF
dev = torch.device("cuda:0")
N, H, W, C, K = 16, 100, 168, 256, 256
torch.manual_seed(0)
x = torch.randn(N, H, W, C, device=dev).relu().bfloat16()
dy = (torch.randn(N, H, W, K, device=dev) * 0.1).bfloat16()
w = (torch.randn(K, 1, 1, 9 * C, device=dev) * 0.02)
_, wt = HF.weight_prep(w)
off = torch.randn(N, H, W, 18, device=dev) * 0.5
doff = torch.zeros_like(off)
for _ in range(3):
    HF.deform_conv_bwd_fused(dy, wt, x, off, None, (3, 3), 1, 1, 1, 1, doff, None)
torch.cuda.synchronize()
PY
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ATOMIC_RETURN SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
  tag=$(echo $pass | tr ' ' '_' | cut -c1-40)
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $O/pmc_$tag -- python3 $O/one.py > $O/pmc_$tag.log 2>&1 || { echo "pass failed: $pass"; tail -3 $O/pmc_$tag.log; continue; }
  f=$(find $O/pmc_$tag -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "dcn_bwd_fused" in r["Kernel_Name"]]
agg = collections.defaultdict(list)
for r in rows:
    agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
n = None
for k, v in agg.items():
    print(f"{k:34s} per launch {sum(v) / len(v):16.0f}   ({len(v)} launches)")
PY
  rm -rf $O/pmc_$tag
done
