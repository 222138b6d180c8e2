#!/bin/bash
# Rehearsal of bench.py's N > 1 control flow on a ONE-GPU box: two ranks share the card, the collectives go through gloo
# (host-staged; hive_amd/distributed.py), so what is exercised is the sharding, the merge modes, both scaling legs and the JSON line -- not RCCL
# and not the timing.  Usage (GPU box): bash tools/rehearse_multi_gpu.sh  -> the full JSON lines on stdout, a one-line digest of each on stderr
cd $GRAFT_REPO_ROOT
export HIVE_DIST_BACKEND=gloo
for mode in "--merge sum" "--scaling strong --merge sum" "--merge exact"; do
  echo "== python bench.py --gpus 2 --steps 2 --warmup 1 --batch 10 --frames 24 --voxel 0.04 --no-cpu-baseline $mode   (no launcher: bench.py starts its own ranks)"
  timeout -k 10 280 python bench.py --gpus 2 --steps 2 --warmup 1 --batch 10 --frames 24 --voxel 0.04 --no-cpu-baseline $mode 2>&1 | grep '^{"metric"' | tee /tmp/rehearse_line.json
  python3 - >&2 <<'PY'
import json
try:
    d = json.load(open("/tmp/rehearse_line.json"))
    other = d.get("strong") or d.get("weak") or {}
    chk = d.get("timed_volume_check", {})
    print("   digest:", {"metric": d["metric"], "scaling": d["scaling"], "value": round(d["value"], 1), "frames_total": d["config"]["frames_total"],
                         "other_leg": {k: other.get(k) for k in ("scaling", "value", "frames_total")} if other else None,
                         "timed_volume_check": {k: chk.get(k) for k in ("pass", "frames_integrated", "frames_per_rank", "weight_sum", "expected_weight_sum")}})
except Exception as e:
    print("   digest: no JSON line", e)
PY
done
