#!/bin/bash
# A/B of environment switches, alternating processes on one box: tools/ab_env.sh <batch> <dtype> <rounds> "VAR=a" "VAR=b" ...   (use "X=" for the default)
b=$1; d=$2; r=$3; shift 3
for i in $(seq $r); do
  for kv in "$@"; do
    echo -n "$kv  "; env $kv python tools/forward_ms.py $b $d 40 2>&1 | grep -v amdgpu.ids
  done
done
