#!/bin/bash
# Per-kernel average durations of the TSDF leg (prep, work list, sort, sweep) on tools/probe_sweep_ab.py's room scene: rocprofv3 --kernel-trace --stats.
# Usage (GPU box): [PROBE_CONFIGS=...] tools/kernel_stats_probe.sh <tag> [lib]   -> gpurun_out/kstats_<tag>.txt
OUT=$GRAFT_REPO_ROOT/gpurun_out/kstats_$1
[ -n "$2" ] && export HIVE_AMD_LIB=$GRAFT_REPO_ROOT/$2
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PROBE_CONFIGS="${PROBE_CONFIGS:-SORT=1}" timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/probe_sweep_ab.py 16 room > $OUT.log 2>&1
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY' | tee $OUT.txt
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r["Name"] for k in ("integrate_multi", "build_worklist_multi", "sort_worklist", "prep_frame")):
        print(f"{r['Name'][:58]:60s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs']) / 1e3:8.1f}")
PY
rm -rf $OUT
