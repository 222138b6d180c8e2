#!/usr/bin/env python3
"""gpurun_out/prof_r02 -> profiles/r02_kernel_stats.csv, profiles/r02_integrate_pmc.json, profiles/integrate_traffic.json
(HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE, KiB as reported: the gfx950 correction of MI355X_MICROARCH.md, HBM section)."""
import csv, glob, json, os, shutil
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out", "prof_r02"), os.path.join(root, "profiles")
stats = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(max(stats, key=os.path.getmtime), os.path.join(dst, "r02_kernel_stats.csv"))
stats = glob.glob(os.path.join(src, "trace_1024", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(max(stats, key=os.path.getmtime), os.path.join(dst, "r02_kernel_stats_1024cubed_1080p.csv"))
    log = os.path.join(src, "trace_1024.log")
    if os.path.exists(log):
        shutil.copy(log, os.path.join(dst, "r02_probe_1024cubed_1080p.log"))
def mean_counter(sub, counter):
    vals = []
    files = glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True)
    for f in sorted(files, key=os.path.getmtime)[-1:]:  # the newest run only (gpurun merges into gpurun_out without deleting older runs)
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"]
            timed = "false, false" in name or "Lb0ELb0E" in name or "integrate_multi_kernel" in name  # the timed variants (no COUNT, no ACCUM)
            if row["Counter_Name"] == counter and ("integrate_kernel" in name or "integrate_multi_kernel" in name) and timed:
                vals.append(float(row["Counter_Value"]))
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)
out = {"kernel": "integrate_multi_kernel<RM> (up to 4 frames per launch) / integrate_kernel<4, RM, COUNT=false, ACCUM=false>", "correction": "reads = 2 x FETCH_SIZE (gfx950 tallies the 128-B requests of a 16 B/lane stream at 64 B), writes = WRITE_SIZE; KiB"}
traffic = {"source": "profiles/r02_integrate_pmc.json"}
for scene, key in (("bench", "hbm_bytes_per_launch"), ("room", "hbm_bytes_per_launch_room")):
    fetch, nf = mean_counter(f"pmc_fetch_{scene}", "FETCH_SIZE")
    write, nw = mean_counter(f"pmc_write_{scene}", "WRITE_SIZE")
    out[scene] = {"launches": nf, "FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write}
    if fetch is not None and write is not None:
        out[scene]["hbm_bytes_per_launch"] = (2.0 * fetch + write) * 1024.0
        traffic[key] = out[scene]["hbm_bytes_per_launch"]
json.dump(out, open(os.path.join(dst, "r02_integrate_pmc.json"), "w"), indent=1)
if len(traffic) > 1:
    json.dump(traffic, open(os.path.join(dst, "integrate_traffic.json"), "w"))
print(json.dumps(out, indent=1))
