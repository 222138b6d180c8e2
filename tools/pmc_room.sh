#!/bin/bash
# SQ / TA / TCP counter passes for the integrate kernel on the room scene.  Usage: tools/pmc_room.sh <outdir-name> [lib]
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
[ -n "$2" ] && export HIVE_AMD_LIB=$GRAFT_REPO_ROOT/$2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P="python3 $GRAFT_REPO_ROOT/tools/probe_integrate.py --reps 1 --frames 10"
pass() { n=$1; shift; timeout -k 10 90 rocprofv3 --pmc "$@" --output-format csv -d $OUT/p$n -- $P > $OUT/p$n.log 2>&1 || echo "pass $n failed"; }
pass A SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE
pass B SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL
pass C TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum
pass F TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum
pass D TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum
pass E TCP_TOTAL_ACCESSES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$OUT/p?/")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "integrate_kernel" in n and ("false, false" in n or "Lb0ELb0E" in n):
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            print(f"{k:44s} n={len(v):3d} avg={sum(v) / len(v):16.1f}")
PY
rm -rf $OUT/p?/
