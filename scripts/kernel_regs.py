#!/usr/bin/env python3
"""Register / LDS budget of the kernels in a device-only assembly listing (hipcc --cuda-device-only -S): name, SGPRs, VGPRs, spills, LDS,
and the waves per SIMD the VGPR count allows on gfx950 (512 VGPRs per SIMD lane, allocated in blocks of 8).
usage: kernel_regs.py <file.s> [substring]"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
key = sys.argv[2] if len(sys.argv) > 2 else ""
rows = []
for b in txt.split("  - .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, b) or [None, "-"])[1]
    name = g("name")
    if key and key not in name:
        continue
    rows.append((name, g("sgpr_count"), g("vgpr_count"), g("sgpr_spill_count"), g("vgpr_spill_count"), g("group_segment_fixed_size")))
names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
for r, n in zip(rows, names):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"\(.*", "", n); n = n.replace("void ", "")
    v = int(r[2]); vb = (v + 7) // 8 * 8
    print("%-44s sgpr %3s vgpr %3s (waves/SIMD %d) sgpr-spill %3s vgpr-spill %3s lds %6s" % (n[:44], r[1], r[2], min(8, 512 // max(vb, 1)), r[3], r[4], r[5]))
