#!/bin/bash
# Round-2 profile set (run on the GPU box: gpurun -- bash tools/profile_r02.sh):
#   1. rocprofv3 --kernel-trace --stats of the default bench command        -> gpurun_out/prof_r02/trace
#   2. FETCH_SIZE / WRITE_SIZE of integrate_kernel, separate passes, on the bench (DPT-fed) scene and on the room scene
#      (analytic depth; tools/probe_integrate.py)                             -> gpurun_out/prof_r02/pmc_*
# tools/profile_r02_summary.py condenses them into profiles/r02_*.
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r02
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $GRAFT_REPO_ROOT/bench.py --timed-only --steps 6 --warmup 2"
ROOM="python3 $GRAFT_REPO_ROOT/tools/probe_integrate.py --reps 1 --frames 30 --no-mesh"  # 12 degrees apart, as the bench's room sample
$BENCH > $OUT/warm.log 2>&1   # MIOpen's find results are cached: the profiled runs show steady-state kernels
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $OUT/trace.log 2>&1 || echo "trace failed"
find $OUT/trace -name "*kernel_trace.csv" -delete
for pass in fetch:FETCH_SIZE write:WRITE_SIZE; do
  n=${pass%%:*}; c=${pass#*:}
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_${n}_bench -- $BENCH > $OUT/pmc_${n}_bench.log 2>&1 || echo "$n bench failed"
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_${n}_room -- $ROOM > $OUT/pmc_${n}_room.log 2>&1 || echo "$n room failed"
done
# 3. BASELINE config 4's TSDF side: 1920 x 1080 frames into 1024^3 (5 mm), kernel trace                -> gpurun_out/prof_r02/trace_1024
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_1024 -- python3 $GRAFT_REPO_ROOT/tools/probe_integrate.py --height 1080 --width 1920 --voxel 0.005 --frames 6 --reps 2 --no-mesh > $OUT/trace_1024.log 2>&1 || echo "1024 trace failed"
find $OUT/trace_1024 -name "*kernel_trace.csv" -delete
for f in $(find $OUT -name "*counter_collection.csv"); do (head -1 $f; grep -E "integrate_kernel|integrate_multi_kernel" $f) > $f.tmp && mv $f.tmp $f; done
du -sh $OUT; echo profile done
