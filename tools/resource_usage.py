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
extra = sys.argv[2:]  # extra compiler flags (experiments), e.g. -mllvm -disable-machine-licm
sys.path.insert(0, REPO)
import __graft_entry__ as ge  # the product's translation units and flags
# a unit other than auvplan.hip LAUNCHES only the kernels named here (the shared headers' other kernels are compiled into it as
# unused internal-linkage copies: not listed); auvplan.hip launches everything else
OWN = {"pf_kernels.hip": ("pf_",), "rows_kernels.hip": ("rrt_rows_kernel", "rrt_rows_stream_kernel", "rrt_stream_kernel"), "prrt_rows_kernels.hip": ("prrt_rows_kernel",)}
elsewhere = tuple(n for v in OWN.values() for n in v)
txt = ""
for unit, unit_flags in ge.UNITS:
    cmd = [ge.HIPCC] + ge.HIP_FLAGS + unit_flags + extra + ["-Rpass-analysis=kernel-resource-usage", "-c", "-o", "/dev/null",
                                                            os.path.join(ge.CSRC, unit)]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    for blk in re.split(r"(?=[^\n]*remark: [^\n]*Function Name: )", err):
        m = re.search(r"Function Name: (\S+)", blk)
        if not m:
            continue
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout
        short_name = re.sub(r"\(.*", "", name).replace("auvp::", "").replace("void ", "").strip()
        mine = short_name.startswith(OWN[unit]) if unit in OWN else not short_name.startswith(elsewhere)
        if mine:
            txt += blk
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
    f.write("`hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Rpass-analysis=kernel-resource-usage` on the translation units of\n"
            "`__graft_entry__.UNITS` with their product flags (`pf_kernels.hip`: `-mllvm -disable-machine-licm`)%s\n"
            "(ROCm 7.2).  Occupancy = waves per SIMD the register allocation admits (dynamic LDS can lower it further: see\n"
            "DESIGN.md per kernel).  Regenerate with `python tools/resource_usage.py <tag>`.\n\n" % ((", extra flags `%s`" % " ".join(extra)) if extra else ""))
    f.write("| kernel | VGPRs | AGPRs | SGPRs | scratch B/lane | waves/SIMD (registers) | static LDS B |\n|---|---|---|---|---|---|---|\n")
    for r in rows:
        f.write("| `%s` | %s | %s | %s | %s | %s | %s |\n" % r)
print(out, len(rows), "kernels")
