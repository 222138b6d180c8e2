#!/bin/bash
# A/B of two library builds on the WHOLE timed job (network + sweeps on the side stream), alternating processes on one box: bash tools/ab_step.sh <libA.so> <libB.so> [rounds]
cd $GRAFT_REPO_ROOT
for i in $(seq 1 ${3:-3}); do
  for l in $1 $2; do
    echo -n "$(basename $l)  "; HIVE_AMD_LIB=$PWD/$l python bench.py --timed-only --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(round(d['value'],1),'frames/s', round(d['ms_per_step'],3),'ms/step, sweep in job', round(d['avg_integrate_us'],1),'us')"
  done
done
