#!/bin/bash
# kernel-level steady-state profile of the bench step: 43 steps so that per-step kernels have >= 40 calls
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/steady_$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 40 --warmup 3 --batch ${2:-8} > $OUT/trace.log 2>&1
find $OUT/trace -name "*kernel_trace.csv" -delete
tail -1 $OUT/trace.log | cut -c1-200
