#!/bin/bash
# LDS bank-conflict / MFMA-busy counters of the fused depth head for one library (tools/pmc_head.sh <lib-or-dash>)
cd /tmp && export TMPDIR=/tmp
[ "$1" != "-" ] && export HIVE_AMD_LIB=$GRAFT_REPO_ROOT/$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmchead
rm -rf $OUT && mkdir -p $OUT
python3 $GRAFT_REPO_ROOT/tools/probe_head.py 2>&1 | tail -1
timeout -k 10 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/p -- python3 $GRAFT_REPO_ROOT/tools/probe_head.py > $OUT/p.log 2>&1
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "head_conv" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
print("conflict fraction %.3f  mfma busy %.3f" % (m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"], m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8 * 1024)))
PY
