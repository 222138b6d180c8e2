#!/bin/bash
# PMC passes for the integrate kernel (run on the GPU box via gpurun). Usage: tools/pmc_probe.sh <outdir-name>
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P="python3 $GRAFT_REPO_ROOT/tools/probe_integrate.py --reps 1 --frames 10"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $P > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc1 -- $P > $OUT/pmc1.log 2>&1
rocprofv3 --pmc FETCH_SIZE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/pmc2 -- $P > $OUT/pmc2.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc3 -- $P > $OUT/pmc3.log 2>&1
echo done
