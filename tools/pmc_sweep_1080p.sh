#!/bin/bash
# The fused sweep's counters at BASELINE config 4's geometry (1920 x 1080 into 1024^3, room scene, eight consecutive frames = two sweeps of four per pass; tools/probe_sweep_1080p.py):
# separate rocprofv3 --pmc runs (no tracing beside them), per-launch averages of integrate_multi_kernel.  Usage (GPU box): tools/pmc_sweep_1080p.sh <outdir-name>
#   -> gpurun_out/<outdir-name>/summary.txt
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pass() { n=$1; shift; timeout -k 10 200 rocprofv3 --pmc "$@" --output-format csv -d $OUT/p$n -- python3 $GRAFT_REPO_ROOT/tools/probe_sweep_1080p.py pmc > $OUT/p$n.log 2>&1 || echo "pass $n failed"; }
pass A SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE
pass B SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_ANY
pass D TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum
pass E TCP_TOTAL_ACCESSES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum
pass F FETCH_SIZE
pass G WRITE_SIZE
pass H TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
python3 - > $OUT/summary.txt <<PY
import csv, glob, collections
for d in sorted(glob.glob("$OUT/p?/")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "integrate_multi_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            print(f"{k:44s} launches={len(v):3d} avg per launch={sum(v) / max(len(v), 1):18.1f}")
PY
cat $OUT/summary.txt
rm -rf $OUT/p?/
