#!/usr/bin/env python3
"""profiles/<tag>_steady_step.csv: per-step kernel time of the bench in steady state = (stats of the 16-step
run - stats of the 6-step run) / 10 (bench.py's default frames per step), from tools/steady_profile.sh."""
import csv, glob, os, sys
tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def load(s):
    fs = glob.glob(os.path.join(root, "gpurun_out", f"steady_{tag}", f"s{s}", "**", "*kernel_stats.csv"), recursive=True)
    d = {}
    for r in csv.DictReader(open(max(fs, key=os.path.getmtime))):
        d[r["Name"]] = (int(r["Calls"]), float(r["TotalDurationNs"]))
    return d
a, b = load(6), load(16)
rows = []
for n, (c, t) in b.items():
    c0, t0 = a.get(n, (0, 0.0))
    if c - c0 > 0:
        rows.append((n, (c - c0) / 10.0, (t - t0) / 10.0 / 1e3))
rows.sort(key=lambda r: -r[2])
tot = sum(r[2] for r in rows)
out = os.path.join(root, "profiles", f"{tag}_steady_step.csv")
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "CallsPerStep", "MicrosecondsPerStep", "Percent"])
    for n, c, t in rows:
        w.writerow([n, f"{c:.2f}", f"{t:.1f}", f"{100 * t / tot:.2f}"])
print(f"steady step: {tot / 1e3:.2f} ms of kernels -> {out}")
for n, c, t in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    print(f"{t:9.1f} us {c:7.2f} calls  {n[:120]}")
