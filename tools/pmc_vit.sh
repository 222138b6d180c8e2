#!/bin/bash
# MFMA utilisation / LDS counters of the hand-written MFMA kernels (ViT GEMMs, attention, fused depth head) at the
# bench batch.  Usage (GPU box): tools/pmc_vit.sh <tag>   ->  gpurun_out/pmc_vit_<tag>.json
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcvit_$1
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P="python3 $GRAFT_REPO_ROOT/bench.py --timed-only --steps 2 --warmup 1 --batch ${2:-96}"
$P > $OUT/warm.log 2>&1
pass() { n=$1; shift; timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/p$n -- $P > $OUT/p$n.log 2>&1 || echo "pass $n failed"; }
pass A SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAVES
pass B SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
pass C SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD
python3 - <<PY
import csv, glob, collections, json
kernels = {"gemm256p_kernel<0": "gemm q|k (bias; 256 x 256 tiles)", "gemm256p_kernel<1": "gemm fc1 + GELU (256 x 256 tiles)", "gemm256p_kernel<2": "gemm proj / fc2 + residual (256 x 256 tiles)",
           "gemm_kernel<0": "gemm (bias; 128 x 128 tiles)", "gemm_kernel<1": "gemm + GELU (128 x 128 tiles)", "gemm_kernel<2": "gemm + residual (128 x 128 tiles)",
           "gemm256p_kernel<3>": "gemm v^T (256 x 256 tiles)", "gemm_kernel<3": "gemm v^T (128 x 128 tiles)", "attention_kernel": "attention", "head_conv_kernel": "fused depth head",
           "conv_kernel<256, 256, 0>": "conv -> 256-channel tiles (decoder RCUs, layer_rn)", "conv_kernel<256, 256, 1>": "conv + GroupNorm statistics (ResNetV2, 256-channel tiles)",
           "conv_kernel<256, 256, 2>": "conv + GroupNorm apply, second pass (ResNetV2 conv3 / downsample)", "conv_kernel<256, 128, 0>": "conv 3x3 256 -> 128 (output_conv[0])"}
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p?/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        for k, label in kernels.items():
            if k in r["Kernel_Name"]:
                acc[label][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for label, cs in acc.items():
    m = {k: sum(v) / len(v) for k, v in cs.items()}
    m["launches_sampled"] = len(next(iter(cs.values())))
    # SQ_* cycle counters are summed over the shader engines' SQs; MFMA busy is in cycles, SQ_BUSY_CYCLES per SE (32 on the chip)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m:
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; MFMA busy cycles over all 1024 SIMDs' matrix cores (4 per CU)
        m["mfma_busy_fraction"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8.0 * 256 * 4)
    if "SQ_LDS_BANK_CONFLICT" in m and m.get("SQ_LDS_IDX_ACTIVE"):
        m["lds_bank_conflict_fraction"] = m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"]
    out[label] = m
json.dump(out, open("$GRAFT_REPO_ROOT/gpurun_out/pmc_vit_$1.json", "w"), indent=1)
for label, m in out.items():
    print(label, {k: (round(v, 4) if v < 10 else int(v)) for k, v in m.items()})
PY
rm -rf $OUT/p?/
