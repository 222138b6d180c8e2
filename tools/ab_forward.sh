#!/bin/bash
# A/B of two builds of the library on ONE box: alternating processes (clock drift and box-to-box spread cancel), each printing the median forward time.
# Usage (GPU box): bash tools/ab_forward.sh <libA.so> <libB.so> [rounds=3] [batch=107] [dtype=bf16]
cd $GRAFT_REPO_ROOT
A=$1; B=$2; R=${3:-3}; BATCH=${4:-107}; DT=${5:-bf16}
for i in $(seq 1 $R); do
  HIVE_AMD_LIB=$PWD/$A python tools/forward_ms.py $BATCH $DT || exit 1
  HIVE_AMD_LIB=$PWD/$B python tools/forward_ms.py $BATCH $DT || exit 1
done
