"""A short run of the randomised GPU-vs-checker sweep (tests/experiments/soak_gpu.py): random worlds and planner parameters
through every RRT.exploring kernel / cull variant, astar_fixLenSOG / astar_fixLen and Planner_RRT.planning, bit for bit."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [5, 6])
def test_random_cases_match_the_checker(seed):
    r = subprocess.run([sys.executable, os.path.join(REPO, "tests", "experiments", "soak_gpu.py"), "10", str(seed)],
                       cwd=REPO, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "0 mismatches" in r.stdout


def test_helper_wavefront_kernels_random_cases():
    """rrt_duo_kernel / rrt_trio_kernel (speculative stages whose redo paths depend on the timing between wavefronts) against
    rrt_explore_kernel over random worlds, parameters, batch sizes and budgets, every case repeated"""
    r = subprocess.run([sys.executable, os.path.join(REPO, "tests", "experiments", "soak_duo.py"), "60", "11"],
                       cwd=REPO, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "0 mismatches" in r.stdout


def test_pre_generated_random_streams_random_cases():
    """rrt_stream_kernel + rrt_rows_stream_kernel (round 6) against rrt_rows_kernel and the one-episode kernel: streams of every
    length around what the batch draws (finished on the stream kernel, or redone), continued generators, the host's choice"""
    r = subprocess.run([sys.executable, os.path.join(REPO, "tests", "experiments", "soak_stream.py"), "80", "7"],
                       cwd=REPO, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " 0 mismatches" in r.stdout and "80 cases" in r.stdout


def test_planner_helper_wavefronts_random_cases():
    """prrt_pipe_kernel (four wavefronts per Planner_RRT episode) against prrt_kernel: random worlds, goals near and
    far (plannings that end while the next step is already inserted), parameters and budgets, every case repeated"""
    r = subprocess.run([sys.executable, os.path.join(REPO, "tests", "experiments", "soak_planner_duo.py"), "40", "5"],
                       cwd=REPO, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "0 mismatches" in r.stdout


def test_particle_filter_random_cases():
    """pf_step_kernel against the checker: particle counts 2 .. 2048, 1 .. 4 AUVs, 1 .. 12 steps, concentrated and spread weights"""
    r = subprocess.run([sys.executable, os.path.join(REPO, "tests", "experiments", "soak_pf.py"), "80", "2"],
                       cwd=REPO, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "0 mismatching" in r.stdout


def test_shark_grid_random_cases():
    """the three window-sum kernels of SharkOccupancyGrid.convert against the checker: grid shapes around the tile edges, sparse /
    duplicated cell lists, radii 1 .. 14 cells"""
    r = subprocess.run([sys.executable, os.path.join(REPO, "tests", "experiments", "soak_sog.py"), "40", "3"],
                       cwd=REPO, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "0 mismatches" in r.stdout


# ---- the speculative pipelines under a diagnostic build (libauvplan_diag.so = -DAUVP_PIPE_DIAG, built by __graft_entry__.build):
# delay injection on one stage at a time, and bounded waits that run out
DIAG = os.path.join(REPO, "auv_sim_amd", "libauvplan_diag.so")
# (script, cases, seed, {stage name: "mode,mod,residue"}): which wavefronts of a workgroup belong to the stage
#   rrt_duo: wave & 1 (0 main, 1 helper); rrt_trio <J, 3>: wave % 3 (0 M geometry, 1 H stream, 2 T tree) -- the <J, 4> form of the
#   same sweep has wave % 4, so `1,3,r` lands on varying roles there: all the better; prrt_pipe: wave % 5 (0 M, 1 H, 2 S, 3 G, 4 D), or wave & 3 without the draw wavefront
STAGES = {"soak_duo.py": {"main_or_M": "1,3,0", "H": "1,3,1", "T": "1,3,2", "helper": "1,2,1"},
          # (round 6: five wavefronts per episode -- wave % 5: 0 M, 1 H, 2 S, 3 G, 4 D the draw wavefront; every other repetition of
          # the script runs the four-wavefront form, where `1,5,r` lands on varying roles)
          "soak_planner_duo.py": {"M": "1,5,0", "H": "1,5,1", "S": "1,5,2", "G": "1,5,3", "D": "1,5,4"}}


def _diag_run(script, cases, seed, jitter=None, spin=None):
    assert os.path.exists(DIAG), "libauvplan_diag.so is built by __graft_entry__.build()"
    env = dict(os.environ, AUVPLAN_LIBRARY=DIAG)
    env.pop("AUVP_DIAG_JITTER", None)
    env.pop("AUVP_DIAG_SPIN", None)
    if jitter:
        env["AUVP_DIAG_JITTER"] = jitter
    if spin:
        env["AUVP_DIAG_SPIN"] = str(spin)
    r = subprocess.run([sys.executable, os.path.join(REPO, "tests", "experiments", script), str(cases), str(seed)],
                       cwd=REPO, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    last = r.stdout.strip().splitlines()[-1]
    assert " 0 mismatches" in last, last
    return int(last.split("mismatches,")[1].split()[0])   # episodes redone by the pipeline fallback


@pytest.mark.parametrize("script,stage", [(s, k) for s in STAGES for k in STAGES[s]])
def test_pipelines_under_delay_injection(script, stage):
    """a pseudo-random delay of 0 .. 32 x s_sleep 1 (~0 .. 2 000 clocks) before every hand-over word of ONE stage: the rare
    orderings (a stage stalled across an epoch change, a ring wrap during a rewind) must give the one-wavefront kernel's trees"""
    redone = _diag_run(script, 12, 21, jitter="%s,32,7" % STAGES[script][stage])
    assert redone == 0   # (the default spin limit: a delayed stage is waited for, not given up on)


@pytest.mark.parametrize("script,spin", [("soak_duo.py", 6), ("soak_planner_duo.py", 6)])
def test_a_wait_that_runs_out_is_redone_on_the_one_wavefront_kernel(script, spin):
    """bounded waits of six polls (AUVP_DIAG_SPIN: a poll is ~100 clocks, a stage thousands) + a delayed stage: episodes end with AUVP_ERR_PIPELINE inside the
    launch and the host redoes them on the one-wavefront kernel in the same call -- same trees, bucket lists, generator state"""
    redone = _diag_run(script, 12, 4, jitter="%s,64,3" % list(STAGES[script].values())[1], spin=spin)
    assert redone > 0


def test_pipe_fallback_off_leaves_the_declared_status_in_the_summaries():
    """option PIPE_FALLBACK = 0 (include/auvplan.h): an episode whose bounded wait ran out keeps AUVP_ERR_PIPELINE (-9) in its
    summary record and nothing is redone; with the default the same launch returns complete trees and counts the episodes"""
    code = r'''
import numpy as np, sys
sys.path.insert(0, %r)
from auv_sim_amd import _lib, synth
ctx = _lib.Context(0)
w = synth.make_world(seed=1, n_obstacles=64)
ctx.set_world(w["obstacles"], w["habitats"], w["polygon"], w["bins"], w["cells"], w["prob"])
E = 8
init = np.zeros((E, 6)); init[:, 0], init[:, 1] = w["start"]
seeds = np.arange(E, dtype=np.uint64) + 3
ctx.set_option("ROWS", 0); ctx.set_option("TRIO", 1)
ctx.set_option("PIPE_FALLBACK", 0)
s0 = ctx.rrt_explore_batch(init, seeds, 400).copy()
f0 = ctx.pipeline_fallbacks()
ctx.set_option("PIPE_FALLBACK", None)
s1 = ctx.rrt_explore_batch(init, seeds, 400).copy()
f1 = ctx.pipeline_fallbacks()
ctx.set_option("TRIO", 0); ctx.set_option("DUO", 0)
s2 = ctx.rrt_explore_batch(init, seeds, 400).copy()
print(int((s0["status"] == -9).sum()), f0[0], int((s1["status"] < 0).sum()), f1[0], ctx.last_rrt_kernel(),
      all(np.array_equal(s1[k], s2[k]) for k in s1.dtype.names if k != "n_candidates"))
''' % REPO
    env = dict(os.environ, AUVPLAN_LIBRARY=DIAG, AUVP_DIAG_SPIN="6", AUVP_DIAG_JITTER="1,3,1,32,5")
    r = subprocess.run([sys.executable, "-c", code], cwd=REPO, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    n9, redone0, nbad1, redone1, kern, same = r.stdout.strip().splitlines()[-1].split()
    assert int(n9) > 0 and int(redone0) == 0          # the status is visible, nothing was redone
    assert int(nbad1) == 0 and int(redone1) == int(n9) > 0   # the default redoes exactly those episodes ...
    assert same == "True"                              # ... and returns the one-wavefront kernel's results
