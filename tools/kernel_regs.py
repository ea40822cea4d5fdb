#!/usr/bin/env python3
"""VGPR / spill / scratch / LDS of every kernel of one .hip file (hipcc -S, gfx950): python tools/kernel_regs.py csrc/file.hip [filter]
Part of the kernel checklist: 0 scratch, 0 spilled VGPRs, 0 flat_* accesses."""
import os, re, subprocess, sys, tempfile
src = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
extra = sys.argv[3:]
out = os.path.join(tempfile.mkdtemp(), "k.s")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-S", "--cuda-device-only", "-std=c++17",
                       "-I" + os.path.dirname(os.path.abspath(src)), src, "-o", out] + extra, stderr=subprocess.DEVNULL)
txt = open(out).read()
filt = subprocess.run(["c++filt"], input="\n".join(re.findall(r"\.name:\s+(\S+)", txt)), capture_output=True, text=True).stdout.split("\n")
names = dict(zip(re.findall(r"\.name:\s+(\S+)", txt), filt))
for b in re.findall(r"- \.agpr_count:.*?\.wavefront_size:\s+\d+", txt, flags=re.S):
    g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, b).group(1))
    name = re.sub(r"\(.*", "", names.get(re.search(r"\.name:\s+(\S+)", b).group(1), "?").replace("(anonymous namespace)::", "")).replace("void ", "")
    if flt in name:
        print("%-90s vgpr %3d agpr %3d spill %3d scratch %4d lds %6d" % (name[:90], g("vgpr_count"), g("agpr_count"), g("vgpr_spill_count"),
                                                                          g("private_segment_fixed_size"), g("group_segment_fixed_size")))
print("flat_ accesses:", len(re.findall(r"\bflat_(load|store)", txt)))
