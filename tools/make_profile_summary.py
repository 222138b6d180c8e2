#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag>/ into profiles/<tag>_*.{csv,json} (the files the judge reads).

  profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary of the bench command
  profiles/<tag>_integrate_pmc.json FETCH_SIZE / WRITE_SIZE of integrate_kernel per launch (KiB as
                                    reported), and the HBM bytes per launch with the gfx950 correction of
                                    /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE reports half of the
                                    bytes of a wide (16 B/lane) coalesced read, WRITE_SIZE is exact.
  profiles/integrate_traffic.json   what bench.py reads for roofline.traffic
"""
import csv, glob, json, os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
def newest(sub, pattern):
    """gpurun merges every call's files into the same directory: only the latest run counts"""
    files = glob.glob(os.path.join(src, sub, "**", pattern), recursive=True)
    return [max(files, key=os.path.getmtime)] if files else []
stats = newest("trace", "*kernel_stats.csv")
if stats:
    shutil.copy(stats[0], os.path.join(dst, f"{tag}_kernel_stats.csv"))
def mean_counter(sub, counter, pat):
    vals = []
    for f in newest(sub, "*counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter and pat in row["Kernel_Name"] and "true" not in row["Kernel_Name"].split(">")[0].split(",")[2]:
                vals.append(float(row["Counter_Value"]))
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)
fetch, nf = mean_counter("pmc_fetch", "FETCH_SIZE", "integrate_kernel<")
write, nw = mean_counter("pmc_write", "WRITE_SIZE", "integrate_kernel<")
out = {"kernel": "integrate_kernel (timed variants, COUNT=false)", "launches_fetch": nf, "launches_write": nw,
       "FETCH_SIZE_KiB_per_launch": fetch, "WRITE_SIZE_KiB_per_launch": write}
if fetch is not None and write is not None:
    out["hbm_bytes_per_launch"] = (2.0 * fetch + write) * 1024.0
    out["correction"] = "reads = 2 x FETCH_SIZE (gfx950 counts 128-B requests as 64 B for 16 B/lane streams), writes = WRITE_SIZE"
json.dump(out, open(os.path.join(dst, f"{tag}_integrate_pmc.json"), "w"), indent=1)
if "hbm_bytes_per_launch" in out:
    json.dump({"hbm_bytes_per_launch": out["hbm_bytes_per_launch"], "source": f"profiles/{tag}_integrate_pmc.json"},
              open(os.path.join(dst, "integrate_traffic.json"), "w"))
print(json.dumps(out))
