#!/usr/bin/env python3
"""Experiment builds with per-unit flags: python tools/variant_units.py exp/libx.so pf_kernels.hip="-mllvm -amdgpu-sched-strategy=max-ilp" ...
(a unit named on the command line gets its product flags REPLACED by the given ones; prefix the flag string with + to append)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
out = os.path.abspath(sys.argv[1])
over = dict(a.split("=", 1) for a in sys.argv[2:])
units = []
for name, flags in ge.UNITS:
    if name in over:
        v = over[name]
        flags = (flags + v[1:].split()) if v.startswith("+") else v.split()
    units.append((name, flags))
ge.UNITS = units
print(ge.compile_library(out, [], tag="variant_" + os.path.basename(out)))
