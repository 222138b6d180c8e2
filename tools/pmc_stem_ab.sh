#!/bin/bash
# LDS bank-conflict counters of the stem convolution for two library builds (round 5: the four-channel padded patch): bash tools/pmc_stem_ab.sh <libA.so> <libB.so>
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_stem
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for l in $1 $2; do
  n=$(basename $l .so)
  HIVE_AMD_LIB=$GRAFT_REPO_ROOT/$l timeout -k 10 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/$n -- python3 $GRAFT_REPO_ROOT/tools/probe_stem.py > $OUT/$n.log 2>&1 || echo "$n failed"
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$OUT/*/")):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "stem_conv_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    m = {k: sum(v) / len(v) for k, v in acc.items()}
    if m:
        print(d.rstrip("/").split("/")[-1], {k: round(v) for k, v in m.items()}, "conflict fraction", round(m["SQ_LDS_BANK_CONFLICT"] / max(m["SQ_LDS_IDX_ACTIVE"], 1), 3))
PY
