#!/usr/bin/env python
"""Print the kernel sequence of the LAST training step from a rocprofv3 --kernel-trace CSV (step boundary = sgd_kernel), with
start offsets, durations and the gap to the previous kernel's end on the same queue: shows where small launches and bubbles sit."""
import csv
import glob
import os
import sys


def main(d, pattern=None):
    f = glob.glob(os.path.join(d, "*", "*_kernel_trace.csv"))[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    sgd = [i for i, r in enumerate(rows) if "sgd_kernel" in r["Kernel_Name"]]
    a, b = sgd[-2] + 1, sgd[-1] + 1
    t0 = int(rows[a]["Start_Timestamp"])
    last_end = {}
    for r in rows[a:b]:
        q = r.get("Queue_Id", "0")
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
        last_end[q] = e
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
        if pattern is None or pattern in name:
            print(f"{(s - t0) / 1e3:9.1f} us  q{q:>3s}  dur {(e - s) / 1e3:7.1f}  gap {gap:6.1f}  {name}")
    print("# step span %.3f ms, %d kernels" % ((int(rows[b - 1]["End_Timestamp"]) - t0) / 1e6, b - a))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
