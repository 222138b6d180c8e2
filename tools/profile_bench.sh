#!/bin/bash
# rocprofv3 passes over the default bench command (run on the GPU box: gpurun -- bash tools/profile_bench.sh <tag>)
#   1. --kernel-trace --stats        -> per-kernel time
#   2. --pmc FETCH_SIZE              -> HBM read bytes   (separate passes: TCC has 4 slots, FETCH_SIZE takes 3)
#   3. --pmc WRITE_SIZE              -> HBM write bytes
set -e
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 6 --warmup 2"
# first an un-profiled run: MIOpen's find step (cudnn.benchmark) caches its choices in the user db, so the
# profiled runs below show the steady-state kernels and not the search
$CMD > $OUT/warm.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write.log 2>&1
# keep the run small enough to travel back (<= 64 MiB): the stats summary, and only the integrate rows of the PMC passes
find $OUT/trace -name "*kernel_trace.csv" -delete
for d in pmc_fetch pmc_write; do
  for f in $(find $OUT/$d -name "*counter_collection.csv"); do
    (head -1 $f; grep integrate_kernel $f) > $f.tmp && mv $f.tmp $f
  done
done
du -sh $OUT
echo profile done
