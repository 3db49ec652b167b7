#!/usr/bin/env python
"""Classify the inter-kernel gaps of the busiest (main) queue of a training step.

Input: a rocprofv3 run with --kernel-trace --hip-runtime-trace (CSV).  For the last K steps (delimited by sgd_kernel) every gap
> MIN_US between two consecutive kernels of the main queue is put into one of

  host      the launch call of the next kernel RETURNED after the previous kernel had ended (minus SLACK): the queue ran dry because the
            host had not enqueued the kernel yet;
  xstream   the kernel was enqueued in time, and a kernel of ANOTHER queue ends inside the gap or within SLACK of its end: the main queue
            was waiting for a cross-stream event (hipStreamWaitEvent);
  small     the kernel was enqueued in time and the gap is bracketed by a copy / fill / tiny elementwise kernel (< 12 us): dispatch
            latency of a dependent small launch;
  device    none of the above: barrier packet + cache write-back between two dependent kernels.

Prints the per-class sums per step and the top (previous kernel -> next kernel) pairs.  Usage: tools/classify_gaps.py <dir> [K]"""
import collections
import csv
import glob
import os
import sys

MIN_US, SLACK_US = 3.0, 4.0


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    n = n.split("(")[0]
    for a, b in (("at::native::vectorized_elementwise_kernel", "aten_elt"), ("at::native::", "aten::"), ("__amd_rocclr_", "rocclr_")):
        n = n.replace(a, b)
    return n[:64]


def main(d, K=3):
    kf = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    af = glob.glob(os.path.join(d, "**", "*hip_api_trace.csv"), recursive=True)
    api_end = {}
    if af:
        for r in csv.DictReader(open(af[0])):
            fn = r.get("Function", "")
            if "Launch" in fn or "Memcpy" in fn or "Memset" in fn:
                api_end[r["Correlation_Id"]] = int(r["End_Timestamp"])
    rows = []
    for r in csv.DictReader(open(kf)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0"), r.get("Correlation_Id", "")))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if "sgd_kernel" in r[2]]
    if len(marks) < K + 1:
        print("not enough steps", len(marks))
        return
    tot = collections.Counter()
    pairs = collections.defaultdict(lambda: [0.0, 0, collections.Counter()])
    for s in range(len(marks) - K - 1, len(marks) - 1):
        seg = rows[marks[s] + 1: marks[s + 1] + 1]
        perq = collections.Counter()
        for a, b, _, q, _ in seg:
            perq[q] += b - a
        qmain = max(perq, key=perq.get)
        other_ends = sorted(b for a, b, _, q, _ in seg if q != qmain)
        cls_sum = collections.Counter()
        last = None
        for a, b, n, q, cid in seg:
            if q != qmain:
                continue
            if last is not None:
                gap = (a - last[1]) / 1e3
                if gap > MIN_US:
                    t_api = api_end.get(cid)
                    small = (b - a) < 12e3 or (last[1] - last[0]) < 12e3
                    if t_api is not None and t_api > last[1] - SLACK_US * 1e3:
                        c = "host"
                    elif any(last[1] - SLACK_US * 1e3 <= e <= a + 1e3 for e in other_ends):
                        c = "xstream"
                    elif small:
                        c = "small"
                    else:
                        c = "device"
                    cls_sum[c] += gap
                    p = pairs[(short(last[2]), short(n))]
                    p[0] += gap; p[1] += 1; p[2][c] += 1
            last = (a, b, n)
        span = (seg[-1][1] - rows[marks[s]][1]) / 1e6
        print(f"step span {span:.3f} ms, main queue {qmain}: gaps > {MIN_US} us sum {sum(cls_sum.values()) / 1e3:.3f} ms  "
              + "  ".join(f"{k} {v / 1e3:.3f}" for k, v in sorted(cls_sum.items())))
        tot.update(cls_sum)
    print(f"mean per step: " + "  ".join(f"{k} {v / K / 1e3:.3f} ms" for k, v in sorted(tot.items())) + ("" if af else "  (no HIP API trace: 'host' cannot be told)"))
    print("top (previous -> next) pairs on the main queue, us per step:")
    for (p, n), (g, c, cl) in sorted(pairs.items(), key=lambda kv: -kv[1][0])[:40]:
        print(f"  {g / K:8.1f} us  x{c / K:5.1f}  {dict(cl)}  {p}  ->  {n}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 3)
