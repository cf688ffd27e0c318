"""Side measurements of bench.py, one module per family; bench.py holds the headline step, the rank plumbing and the output line.

oracle/ is test infrastructure.  Inside this package it is imported in ONE kind of place only: the `cpu_baseline` legs (functions
cpu_baseline / cpu_baseline_all_cores and the `cpu_seconds > 0` / `with_cpu` branches of the side measurements), which time the CPU
checker on the host cores AFTER the GPU measurement, on a bounded sample, at N = 1.  Nothing in a timed GPU region touches it.
"""
