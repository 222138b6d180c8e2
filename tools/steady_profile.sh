#!/bin/bash
# Steady-state kernel profile of the bench step: two rocprofv3 --kernel-trace --stats runs that differ only in
# --steps (S1 < S2); tools/steady_diff.py subtracts them, which removes MIOpen's find step and every other
# one-off of the warm-up.  Usage (GPU box): tools/steady_profile.sh <tag> [batch]
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/steady_$1
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $GRAFT_REPO_ROOT/bench.py --timed-only --steps 4 --warmup 2 --batch ${2:-107} > $OUT/warm.log 2>&1
for S in 6 16; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s$S -- python3 $GRAFT_REPO_ROOT/bench.py --timed-only --steps $S --warmup 2 --batch ${2:-107} > $OUT/s$S.log 2>&1
  find $OUT/s$S -name "*kernel_trace.csv" -delete
done
tail -1 $OUT/s16.log | cut -c1-200
