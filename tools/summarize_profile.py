#!/usr/bin/env python
"""Condense rocprofv3 outputs (kernel stats CSV + separate FETCH_SIZE / WRITE_SIZE PMC passes) into the small tracked
files under profiles/: <tag>_kernel_stats.csv (top kernels) and <tag>_pmc.json (per-launch HBM traffic, FETCH_SIZE doubled
as MI355X_MICROARCH.md §HBM prescribes for gfx950 wide coalesced reads; counters are in KiB)."""
import collections
import csv
import glob
import json
import os
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    return name.split("(")[0].strip()


def main(tag, stats_dir, fetch_dir, write_dir, steps):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out_dir = os.path.join(root, "profiles")
    os.makedirs(out_dir, exist_ok=True)
    rows = list(csv.DictReader(open(glob.glob(os.path.join(stats_dir, "*", "*_kernel_stats.csv"))[0])))
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    # The number of steps the TRACE holds, not the number the caller believes it asked for: the optimizer kernel runs exactly once per step
    # (round 5's three architecture footers divided a 10-step trace by 7).  Falls back to the argument when no optimizer kernel is listed.
    opt = [int(r["Calls"]) for r in rows if any(t in r["Name"] for t in ("sgd_kernel", "adam_kernel", "adagrad_kernel"))]
    asked = steps
    if opt and max(opt) != steps:
        steps = max(opt)
    with open(os.path.join(out_dir, f"{tag}_kernel_stats.csv"), "w") as f:
        pre = "SOD_WGRAD_STREAM=0 " if tag.endswith("_serial") else ""
        post = " ; one stream, no kernel overlap: true per-kernel durations" if pre else " ; default two-stream run: backward kernels share the GPU"
        f.write("# %srocprofv3 --kernel-trace --stats -- python bench.py --steps %d --warmup 2 --no-cpu-baseline --no-roofline ; all %d steps incl. warm-up%s\n"
                % (pre, steps - 2, steps, post))
        f.write("kernel,calls,total_ms,avg_us,min_us,max_us,percent\n")
        for r in rows[:40]:
            f.write("%s,%s,%.3f,%.2f,%.2f,%.2f,%s\n" % (short(r["Name"]).replace(",", ";"), r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                     float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
        f.write("# total GPU kernel time %.3f ms over %d steps = %.3f ms/step%s\n" % (total / 1e6, steps, total / 1e6 / steps,
                "" if steps == asked else " (step count taken from the optimizer kernel's %d calls; the caller said %d)" % (steps, asked)))

    def load(d):
        agg = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(glob.glob(os.path.join(d, "*", "*_counter_collection.csv"))[0])):
            k = short(r["Kernel_Name"])
            agg[k][0] += float(r["Counter_Value"])
            agg[k][1] += 1
        return agg

    pmc = {}
    if fetch_dir and write_dir:
        fe, wr = load(fetch_dir), load(write_dir)
        for k in fe:
            if k in wr and fe[k][1] > 0:
                fetch_b = fe[k][0] / fe[k][1] * 1024.0 * 2.0      # gfx950: FETCH_SIZE reports 1/2 of wide coalesced reads
                write_b = wr[k][0] / wr[k][1] * 1024.0
                pmc[k] = {"launches": fe[k][1], "fetch_bytes_per_launch": round(fetch_b), "write_bytes_per_launch": round(write_b),
                          "hbm_bytes_per_launch": round(fetch_b + write_b)}
        # ms_per_step: the step time of the kernel-stats run of the same command (span of its kernel trace is not kept; the bench line's
        # ms_per_step of that run is passed in the environment by tools/profile_step.sh when known)
        ms = os.environ.get("SOD_PROFILE_MS_PER_STEP")
        json.dump({"note": "per-launch averages over a bench.py run; FETCH_SIZE x2 correction applied (gfx950)", "steps": steps,
                   "ms_per_step": float(ms) if ms else None, "kernels": pmc},
                  open(os.path.join(out_dir, f"{tag}_pmc.json"), "w"), indent=1, sort_keys=True)
    print("wrote", tag, len(rows), "kernels;", len(pmc), "with PMC")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None, sys.argv[4] if len(sys.argv) > 4 else None,
         int(sys.argv[5]) if len(sys.argv) > 5 else 7)
