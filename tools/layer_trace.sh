#!/bin/bash
# Launch-ordered kernel trace of ONE network forward at the bench batch (no TSDF overlap): which layer costs what.
# Usage (GPU box): tools/layer_trace.sh [batch] [dtype]   ->  gpurun_out/layer_trace/forward.csv (name, us, grid, in launch order)
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/layer_trace
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/raw -- python3 $GRAFT_REPO_ROOT/tools/layer_trace.py run ${1:-107} ${2:-bf16} > $OUT/run.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/layer_trace.py parse $OUT
find $OUT/raw -name "*kernel_trace.csv" -delete
