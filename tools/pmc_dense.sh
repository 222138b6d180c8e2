#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the integrate kernel: calibration frames with a known byte count (every voxel /
# the first 30 % of every z row updated) and the room scene.  Usage (on the GPU box): tools/pmc_dense.sh <outdir-name>
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {  # name, command...
  name=$1; shift
  "$@" 2>&1 | grep -v amdgpu > $OUT/plain_$name.log
  timeout -k 10 90 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmcA_$name -- "$@" > $OUT/pmcA_$name.log 2>&1
  timeout -k 10 90 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmcB_$name -- "$@" > $OUT/pmcB_$name.log 2>&1
}
run dense1.0 python3 $GRAFT_REPO_ROOT/tools/probe_dense.py --frames 4 --fraction 1.0
run dense0.3 python3 $GRAFT_REPO_ROOT/tools/probe_dense.py --frames 4 --fraction 0.3
run room python3 $GRAFT_REPO_ROOT/tools/probe_integrate.py --reps 1 --frames 10
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$OUT/pmc?_*/")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "integrate_kernel" in r["Kernel_Name"] and "Lb1E" not in r["Kernel_Name"].split("EEv")[0][-12:-4]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        print(d.rstrip("/").split("/")[-1], {k: (len(v), sum(v) / len(v)) for k, v in acc.items()})
PY
rm -rf $OUT/pmc?_*/
