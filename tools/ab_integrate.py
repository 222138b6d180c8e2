#!/usr/bin/env python3
"""A/B harness for integrate-kernel variants.  Kernel times vary by +-10 % between PROCESSES on the same box
(placement of the volume in HBM), so every library is probed in `--rounds` fresh processes, interleaved, and
the minimum and the median of the per-process averages are reported.

    python tools/ab_integrate.py --rounds 5 exp/libhive_A.so exp/libhive_B.so [--env HIVE_TSDF_FAST_AXIS=x]
"""
import argparse
import os
import re
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+", help="library paths relative to the repo root ('-' = the product library)")
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--dense", action="store_true", help="also run the dense calibration frame")
ap.add_argument("--env", action="append", default=[])
args = ap.parse_args()


def run(lib, script, extra):
    env = dict(os.environ)
    for kv in args.env:
        k, v = kv.split("=", 1)
        env[k] = v
    if lib != "-":
        env["HIVE_AMD_LIB"] = os.path.join(ROOT, lib)
    else:
        env.pop("HIVE_AMD_LIB", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", script)] + extra, env=env, capture_output=True, text=True).stdout
    m = re.findall(r"kernel avg ([0-9.]+) us", out)
    return float(m[-1]) if m else float("nan")


room = {lib: [] for lib in args.libs}
dense = {lib: [] for lib in args.libs}
for r in range(args.rounds):
    for lib in args.libs:
        room[lib].append(run(lib, "probe_integrate.py", ["--frames", "20", "--reps", "2"]))
        if args.dense:
            dense[lib].append(run(lib, "probe_dense.py", ["--frames", "4"]))
    print(f"round {r} done", flush=True)
for lib in args.libs:
    line = f"{lib:28s} room min {min(room[lib]):6.1f} med {statistics.median(room[lib]):6.1f} us"
    if args.dense:
        line += f" | dense min {min(dense[lib]):6.1f} med {statistics.median(dense[lib]):6.1f} us"
    print(line + "   " + " ".join(f"{v:.0f}" for v in room[lib]))
