#!/usr/bin/env python3
"""Register / spill table of the kernels in an ISA listing (hipcc -S --cuda-device-only): name, VGPRs, spilled VGPRs / SGPRs, scratch bytes.
Usage: python tools/kernel_regs.py file.s [name filter]"""
import re
import sys

text = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for blk in text.split("  - .agpr_count:")[1:]:
    get = lambda k: (re.search(r"\." + k + r":\s*(\S+)", blk) or [None, "?"])[1]
    name = get("name")
    if flt in name:
        print(f"{name[:100]:100s} vgpr {get('vgpr_count'):>4s} vspill {get('vgpr_spill_count'):>4s} sspill {get('sgpr_spill_count'):>4s} scratch {get('private_segment_fixed_size'):>5s}")
