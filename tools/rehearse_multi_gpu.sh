#!/bin/bash
# Rehearsal of bench.py's N > 1 control flow on a ONE-GPU box: two ranks share the card, the collectives go through gloo
# (host-staged; hive_amd/distributed.py), so what is exercised is the sharding, the merge modes and the JSON line -- not RCCL
# and not the timing.  Usage (GPU box): bash tools/rehearse_multi_gpu.sh
cd $GRAFT_REPO_ROOT
export HIVE_DIST_BACKEND=gloo
for mode in "--merge sum" "--scaling strong --merge sum" "--merge exact"; do
  echo "== $mode"
  timeout -k 10 280 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
      bench.py --gpus 2 --steps 2 --warmup 1 --batch 10 --frames 24 --voxel 0.04 --no-cpu-baseline $mode 2>&1 | grep -v "amdgpu.ids\|Gloo\|UserWarning\|warnings.warn" | tail -3 | cut -c1-900
done
