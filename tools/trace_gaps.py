#!/usr/bin/env python
"""GPU occupancy of the timed steps from a rocprofv3 kernel trace: for the last K steps (delimited by sgd_kernel launches) the span,
the union of kernel-busy intervals (any queue), the idle remainder, the sum of kernel durations (overlap factor) and per-queue busy time;
plus the histogram of idle gaps.  Usage: tools/trace_gaps.py <trace_dir> [K]"""
import csv
import glob
import os
import sys


def main(d, K=3):
    f = glob.glob(os.path.join(d, "*", "*_kernel_trace.csv"))[0]
    rows = []
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if "sgd_kernel" in r[2]]
    if len(marks) < K + 1:
        print("not enough steps", len(marks))
        return
    for s in range(len(marks) - K - 1, len(marks) - 1):
        seg = rows[marks[s] + 1: marks[s + 1] + 1]
        t0, t1 = rows[marks[s]][1], seg[-1][1]
        busy, cur_s, cur_e, gaps = 0, None, None, []
        for a, b, _, _ in seg:
            if cur_e is None:
                cur_s, cur_e = a, b
                gaps.append(a - t0)
            elif a > cur_e:
                busy += cur_e - cur_s
                gaps.append(a - cur_e)
                cur_s, cur_e = a, b
            else:
                cur_e = max(cur_e, b)
        busy += cur_e - cur_s
        tot = sum(b - a for a, b, _, _ in seg)
        perq = {}
        for a, b, _, q in seg:
            perq[q] = perq.get(q, 0) + b - a
        big = sorted(gaps, reverse=True)[:8]
        print(f"step: span {(t1 - t0) / 1e6:.3f} ms  busy(any queue) {busy / 1e6:.3f}  idle {(t1 - t0 - busy) / 1e6:.3f}  sum of kernels {tot / 1e6:.3f}  "
              f"kernels {len(seg)}  gaps>0: {sum(1 for g in gaps if g > 0)}  mean gap {sum(gaps) / max(1, len(gaps)) / 1e3:.2f} us  largest {[round(g / 1e3, 1) for g in big]}")
        print("   per queue busy ms:", {q: round(v / 1e6, 3) for q, v in sorted(perq.items())})
        # gaps on the busiest queue (dependent launches)
        qmain = max(perq, key=perq.get)
        last, qg = None, []
        for a, b, n, q in seg:
            if q != qmain:
                continue
            if last is not None:
                qg.append(a - last)
            last = b
        qg.sort()
        n = len(qg)
        print(f"   busiest queue {qmain}: {n + 1} kernels, sum of its inter-kernel gaps {sum(qg) / 1e6:.3f} ms, median {qg[n // 2] / 1e3:.2f} us, p90 {qg[int(n * 0.9)] / 1e3:.2f} us")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 3)
