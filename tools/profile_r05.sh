#!/bin/bash
# Round-5 profile set (GPU box: gpurun -- bash tools/profile_r05.sh [parts]); parts = any of: trace timed pmc valu small fp16 layers (default: trace timed pmc valu small)
#   fp16 : the default bench command with --dtype fp16 (the reference's model.half())                                        -> .../fp16.log
#   layers: launch-ordered kernel times of one forward at the bench batch (tools/layer_trace.sh)                              -> gpurun_out/layer_trace/forward.csv
#   trace: rocprofv3 --kernel-trace --stats of the default bench command, and of the same with --no-overlap  -> gpurun_out/prof_r05/trace{,_no_overlap}
#   pmc  : FETCH_SIZE / WRITE_SIZE of the integrate kernels, separate passes, bench scene (DPT depth) and room scene (analytic depth,
#          consecutive frames: tools/probe_integrate.py --yaw-step 2.4)                          -> gpurun_out/prof_r05/pmc_{fetch,write}_{bench,room}
#   valu : SQ_INSTS_VALU / SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES / SQ_WAVE_CYCLES / SQ_WAVES / GRBM_GUI_ACTIVE of the same          -> .../pmc_valu_*
#   small: kernel trace of tools/probe_small_kernels.py (marching cubes 512^3, unproject, project_bbox, grid_mesh, ...)           -> .../small
# tools/profile_r05_summary.py condenses them into profiles/r04_*.
PARTS=${*:-trace timed pmc valu small}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r05
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $GRAFT_REPO_ROOT/bench.py --timed-only --no-overlap --steps 3 --warmup 1"  # counters of the sweep ALONE (the timed job overlaps it with the network)
ROOM="python3 $GRAFT_REPO_ROOT/tools/probe_integrate.py --reps 1 --frames 32 --yaw-step 2.4 --no-mesh"
has() { [[ " $PARTS " == *" $1 "* ]]; }
if has trace; then
  rm -rf $OUT/trace
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $OUT/trace.log 2>&1 || echo "trace failed"
  find $OUT/trace -name "*kernel_trace.csv" -delete
  # the same command with the sweeps on the network's stream: every kernel's duration is its own (what roofline.avg_launch_us is compared with)
  rm -rf $OUT/trace_no_overlap
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_no_overlap -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-overlap > $OUT/trace_no_overlap.log 2>&1 || echo "trace (no overlap) failed"
  find $OUT/trace_no_overlap -name "*kernel_trace.csv" -delete
fi
pmc() {  # name, counters...
  n=$1; shift
  for scene in bench room; do
    rm -rf $OUT/pmc_${n}_$scene
    if [ $scene = bench ]; then CMD=$BENCH; else CMD=$ROOM; fi
    timeout -k 10 200 rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc_${n}_$scene -- $CMD > $OUT/pmc_${n}_$scene.log 2>&1 || echo "$n $scene failed"
  done
}
if has timed; then  # ONE un-mixed row for the sweep: only the timed job, sweeps on the network's stream -> profiles/r05_kernel_stats_timed_only.csv
  rm -rf $OUT/timed_only
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/timed_only -- python3 $GRAFT_REPO_ROOT/bench.py --timed-only --no-overlap --steps 20 --warmup 5 > $OUT/timed_only.log 2>&1 || echo "timed-only trace failed"
  find $OUT/timed_only -name "*kernel_trace.csv" -delete
fi
if has pmc; then pmc fetch FETCH_SIZE; pmc write WRITE_SIZE; fi
if has valu; then pmc valu SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE; fi
if has small; then
  rm -rf $OUT/small
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/small -- python3 $GRAFT_REPO_ROOT/tools/probe_small_kernels.py > $OUT/small.json 2> $OUT/small.err || echo "small failed"
  find $OUT/small -name "*kernel_trace.csv" -delete
fi
if has fp16; then timeout -k 10 400 python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --dtype fp16 > $OUT/fp16.log 2>&1 || echo "fp16 failed"; fi
if has layers; then timeout -k 10 300 bash $GRAFT_REPO_ROOT/tools/layer_trace.sh 107 bf16 || echo "layers failed"; fi
for f in $(find $OUT -name "*counter_collection.csv"); do (head -1 $f; grep -E "integrate_kernel|integrate_multi_kernel" $f) > $f.tmp && mv $f.tmp $f; done
du -sh $OUT; echo profile done
