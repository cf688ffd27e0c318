#!/usr/bin/env python3
"""Compiler-reported resources of every kernel of libauvplan.so (hipcc -Rpass-analysis=kernel-resource-usage): VGPRs,
SGPRs, scratch bytes per lane, waves per SIMD the register allocation admits, static LDS.  Writes
profiles/<tag>_kernel_resources.md (the table DESIGN.md quotes).  Runs in the build container (no GPU needed).

usage: python tools/resource_usage.py r3"""
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r3"
src = os.path.join(REPO, "auv_sim_amd", "csrc", "auvplan.hip")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
       "-Rpass-analysis=kernel-resource-usage", "-o", "/dev/null", src]
txt = subprocess.run(cmd, capture_output=True, text=True).stderr
demangle = subprocess.run(["c++filt"], input="\n".join(re.findall(r"Function Name: (\S+)", txt)),
                          capture_output=True, text=True).stdout.split("\n")
blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
rows = []
for b, name in zip(blocks, demangle):
    def g(k):
        m = re.search(k + r": (\S+)", b)
        return m.group(1) if m else "?"
    short = re.sub(r"\(.*", "", name).replace("auvp::", "").replace("void ", "")
    rows.append((short, g("VGPRs"), g("AGPRs"), g("TotalSGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"),
                 g(r"LDS Size \[bytes/block\]")))
out = os.path.join(REPO, "profiles", "%s_kernel_resources.md" % tag)
with open(out, "w") as f:
    f.write("# Kernel resources as the compiler reports them (%s)\n\n" % tag)
    f.write("`hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Rpass-analysis=kernel-resource-usage` on `auv_sim_amd/csrc/auvplan.hip`\n"
            "(ROCm 7.2).  Occupancy = waves per SIMD the register allocation admits (dynamic LDS can lower it further: see\n"
            "DESIGN.md per kernel).  Regenerate with `python tools/resource_usage.py <tag>`.\n\n")
    f.write("| kernel | VGPRs | AGPRs | SGPRs | scratch B/lane | waves/SIMD (registers) | static LDS B |\n|---|---|---|---|---|---|---|\n")
    for r in rows:
        f.write("| `%s` | %s | %s | %s | %s | %s | %s |\n" % r)
print(out, len(rows), "kernels")
