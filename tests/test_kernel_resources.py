"""The hot kernels' register budgets as the compiler reports them (cross-compiled for gfx950, no GPU needed): the budgets the
measured numbers of DESIGN.md rest on -- no scratch in the expansion kernels, the occupancies the LDS plans assume."""
import os
import re
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _resources(unit):
    import __graft_entry__ as ge
    flags = dict(ge.UNITS)[unit]
    cmd = [ge.HIPCC] + ge.HIP_FLAGS + flags + ["-Rpass-analysis=kernel-resource-usage", "-c", "-o", os.devnull, os.path.join(ge.CSRC, unit)]
    err = subprocess.run(cmd, capture_output=True, text=True, cwd=REPO).stderr
    out = {}
    for blk in re.split(r"(?=[^\n]*remark: [^\n]*Function Name: )", err):
        m = re.search(r"Function Name: (\S+)", blk)
        if not m:
            continue
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        g = lambda k: int(re.search(k + r": (\d+)", blk).group(1))
        out[re.sub(r"\(.*", "", name).replace("void ", "")] = dict(vgprs=g("VGPRs"), scratch=g(r"ScratchSize \[bytes/lane\]"),
                                                                  waves=g(r"Occupancy \[waves/SIMD\]"))
    return out


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_rows_kernel_fits_three_wavefronts_without_scratch():
    """rrt_rows_kernel (the headline): one 12-wavefront workgroup per CU = three per SIMD, i.e. at most 168 VGPRs, and no
    scratch (a spill in the expansion loop is memory traffic per iteration)"""
    res = _resources("rows_kernels.hip")
    r = res["auvp::rrt_rows_kernel"]
    assert r["scratch"] == 0 and r["vgprs"] <= 168 and r["waves"] >= 3, r
    # round 6: the same body reading pre-generated random numbers, and the generator that runs ahead of it (one wavefront per
    # episode, eight per SIMD so that the stores of 64 wavefronts per CU are in flight)
    r = res["auvp::rrt_rows_stream_kernel<12>"]
    assert r["scratch"] == 0 and r["vgprs"] <= 168 and r["waves"] >= 3, r
    r = res["auvp::rrt_stream_kernel"]
    assert r["scratch"] == 0 and r["vgprs"] <= 32 and r["waves"] == 8, r


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_particle_filter_and_planner_rows_kernels_without_scratch():
    r = _resources("pf_kernels.hip")["auvp::pf_step_kernel<512, 2>"]
    assert r["scratch"] == 0 and r["vgprs"] <= 128 and r["waves"] >= 4, r  # two 8-wavefront workgroups per CU
    r = _resources("prrt_rows_kernels.hip")["auvp::prrt_rows_kernel<true>"]
    assert r["scratch"] == 0 and r["vgprs"] <= 168, r


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_planner_pipeline_fits_its_workgroup_without_scratch():
    """prrt_pipe_kernel (config 4): a workgroup of three five-wavefront episodes = fifteen wavefronts on a CU, i.e. four on a SIMD:
    at most 128 VGPRs, and no scratch on any stage's chain (round 4 found register arrays under a computed index that had
    silently become scratch arrays on H's chain)"""
    res = _resources("auvplan.hip")
    for nw in (4, 5):
        r = res["auvp::prrt_pipe_kernel<4, %d>" % nw]
        assert r["scratch"] == 0 and r["vgprs"] <= 128 and r["waves"] >= 4, (nw, r)
