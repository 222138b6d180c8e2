#!/bin/bash
# MFMA utilisation / LDS counters of the hand-written MFMA kernels (ViT GEMMs, attention, fused depth head) at the
# bench batch.  Usage (GPU box): tools/pmc_vit.sh <tag>   ->  gpurun_out/pmc_vit_<tag>.json
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcvit_$1
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P="python3 $GRAFT_REPO_ROOT/bench.py --timed-only --no-overlap --steps 2 --warmup 1 --batch ${2:-107}"
$P > $OUT/warm.log 2>&1
pass() { n=$1; shift; timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/p$n -- $P > $OUT/p$n.log 2>&1 || echo "pass $n failed"; }
pass A SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAVES
pass B SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
pass C SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD
python3 - <<PY
import csv, glob, collections, json, re
# kernel families by name pattern (rocprofv3 prints some template instantiations demangled, some mangled: both forms)
kernels = [(r"gemm256p_kernel(<[^,>]*, 0>|I[^L]*Li0E)", "gemm q|k (bias; 256 x 256 tiles)"), (r"gemm256p_kernel(<[^,>]*, 1>|I[^L]*Li1E|<bool _Accum, int, E>)", "gemm fc1 + GELU, readout (256 x 256 tiles)"),  # (rocprofv3 garbles this one instantiation's demangled name)
           (r"gemm256p_kernel(<[^,>]*, 2>|I[^L]*Li2E)", "gemm proj / fc2 + residual (256 x 256 tiles)"), (r"gemm256p_kernel(<[^,>]*, 3>|I[^L]*Li3E)", "gemm v^T (256 x 256 tiles)"),
           (r"gemm_kernel", "gemm (128 x 128 tiles)"), (r"attention_kernel", "attention"), (r"head_conv_kernel", "fused depth head"),
           (r"conv_kernel(<[^,>]*, 256, 256, 0>|I[^L]*Li256ELi256ELi0E)", "conv -> 256-channel tiles (decoder RCUs, layer_rn)"),
           (r"conv_kernel(<[^,>]*, 256, 256, 1>|I[^L]*Li256ELi256ELi1E)", "conv + GroupNorm statistics (ResNetV2, 256-channel tiles)"),
           (r"conv_kernel(<[^,>]*, 256, 256, 2>|I[^L]*Li256ELi256ELi2E)", "conv + GroupNorm apply, second pass (ResNetV2 conv3 / downsample)"),
           (r"conv_kernel(<[^,>]*, 256, 128, 0>|I[^L]*Li256ELi128ELi0E)", "conv 3x3 256 -> 128 (output_conv[0])"), (r"stem_conv_kernel", "ResNetV2 stem 7x7/2"), (r"bneck_conv3x3_kernel", "64-channel bottleneck 3x3 with GroupNorm on load (bneck.hip)")]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p?/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        for pat, label in kernels:
            if re.search(pat, r["Kernel_Name"]):
                acc[label][r["Counter_Name"]].append(float(r["Counter_Value"]))
                break
out = {}
for label, cs in acc.items():
    m = {k: sum(v) / len(v) for k, v in cs.items()}
    m["launches_sampled"] = len(next(iter(cs.values())))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m:
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; MFMA busy cycles over all 1024 SIMDs' matrix cores (4 per CU)
        m["mfma_busy_fraction"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8.0 * 256 * 4)
    if "SQ_LDS_BANK_CONFLICT" in m and m.get("SQ_LDS_IDX_ACTIVE"):
        m["lds_bank_conflict_fraction"] = m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"]
    out[label] = m
out = {"mfma_busy": {k: round(v["mfma_busy_fraction"], 4) for k, v in out.items() if "mfma_busy_fraction" in v}, "command": "$P (rocprofv3 --pmc, three passes)",
       "note": "mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 matrix cores): 16 busy cycles per v_mfma_f32_16x16x32, 32 per 32x32x16",
       "kernels": out}
json.dump(out, open("$GRAFT_REPO_ROOT/gpurun_out/pmc_vit_$1.json", "w"), indent=1)
for label, v in out["mfma_busy"].items():
    print(f"{v:6.3f}  {label}")
PY
