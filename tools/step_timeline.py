"""Text timeline of ONE training step from a rocprofv3 kernel trace (--kernel-trace --output-format csv): per time slot and hardware queue,
the share of the slot the queue was busy and the kernel that filled most of it.  python tools/step_timeline.py <trace dir> [slot_us=250]
The step is the interval between the ends of the last two optimizer kernels."""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
slot = float(sys.argv[2]) if len(sys.argv) > 2 else 250.0
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [int(r["End_Timestamp"]) for r in rows if "sgd_kernel" in r["Kernel_Name"] or "adam" in r["Kernel_Name"].lower()]
t0, t1 = ends[-2], ends[-1]


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("sodconv::", "").replace("void ", "")
    n = n.split("(")[0]
    return n[:34]


queues = collections.OrderedDict()
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e <= t0 or s >= t1:
        continue
    queues.setdefault(r.get("Queue_Id", "0"), []).append((max(s, t0), min(e, t1), short(r["Kernel_Name"])))
order = sorted(queues, key=lambda q: -sum(e - s for s, e, _ in queues[q]))
print(f"step {1e-6 * (t1 - t0):.3f} ms; queues by busy time: " + ", ".join(f"{q}: {1e-6 * sum(e - s for s, e, _ in queues[q]):.2f} ms / {len(queues[q])} kernels" for q in order))
nslot = int((t1 - t0) / (slot * 1e3)) + 1
for i in range(nslot):
    a, b = t0 + i * slot * 1e3, min(t0 + (i + 1) * slot * 1e3, t1)
    cells = []
    for q in order:
        busy = collections.Counter()
        for s, e, n in queues[q]:
            o = min(e, b) - max(s, a)
            if o > 0:
                busy[n] += o
        tot = sum(busy.values())
        cells.append(f"{100 * tot / (b - a):3.0f}% {busy.most_common(1)[0][0] if busy else '':34s}")
    print(f"{1e-3 * (a - t0):7.0f} us | " + " | ".join(cells))

if len(sys.argv) > 3:      # python tools/step_timeline.py <trace dir> <slot_us> <pattern>: every launch matching the pattern, with its queue's neighbours
    pat = sys.argv[3]
    for q in order:
        ks = sorted(queues[q])
        for i, (s, e, n) in enumerate(ks):
            if pat in n:
                prev = ks[i - 1][2] if i else ""
                nxt = ks[i + 1][2] if i + 1 < len(ks) else ""
                print(f"queue {q} at {1e-3 * (s - t0):8.1f} us: {n} {1e-3 * (e - s):7.1f} us | after {prev} | before {nxt}")
