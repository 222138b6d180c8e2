# A/B of the merged q | k + v^T launch (HIVE_QKV_MERGE: 0 = two launches, 1 = merged where the tiles fit one workgroup per CU, 2 = also two per CU), alternating processes on one box
set -e
for r in 1 2 3; do
  for m in 0 1 2; do
    echo -n "merge=$m "; HIVE_QKV_MERGE=$m python tools/forward_ms.py 2 fp16 40 2>&1 | grep -v amdgpu.ids
  done
done
