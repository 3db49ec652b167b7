"""Prints which kernels surround the runtime's buffer-copy launches inside one training step (from a rocprofv3 kernel trace CSV)."""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][-70:] for r in rows]
idx = [i for i, n in enumerate(names) if "sgd_kernel" in n]
a, b = idx[-2], idx[-1]
seq = names[a + 1:b + 1]
print(len(seq), "kernels in the step")
pat = sys.argv[2] if len(sys.argv) > 2 else "copyBuffer"
out = [(seq[i - 1][-45:] if i else "", seq[i + 1][-45:] if i + 1 < len(seq) else "") for i, n in enumerate(seq) if pat in n]
print(len(out), pat)
for k, v in collections.Counter(out).most_common(30):
    print(v, k)

if len(sys.argv) > 3:       # per-kernel launch counts and time inside the step
    dur = collections.defaultdict(lambda: [0, 0.0])
    for r in rows[a + 1:b + 1]:
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][-70:]
        d = dur[n]
        d[0] += 1
        d[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot = sum(v[1] for v in dur.values())
    span = (int(rows[b]["End_Timestamp"]) - int(rows[a]["End_Timestamp"])) / 1e3
    print("sum of kernel durations %.1f us, wall span of the step %.1f us" % (tot, span))
    for n, (c, t) in sorted(dur.items(), key=lambda kv: -kv[1][1]):
        print("%4d %9.1f us  %s" % (c, t, n))
