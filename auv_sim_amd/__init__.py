"""auv_sim_amd -- MI355X-native path-planning hot path of hmc-lair-shark-tracking/auv-sim.

Only what the path needs: csrc/ (HIP kernels + C-ABI), the ctypes binding, and the Python mirror of
the reference planner API.  Importing the package does not touch the GPU; creating a planner does,
and fails loudly when libauvplan.so or the device is missing (there is no CPU fallback).
"""
__all__ = ["synth"]
