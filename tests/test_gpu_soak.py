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


def test_planner_helper_wavefronts_random_cases():
    """prrt_duo_kernel with two and three wavefronts per Planner_RRT episode against prrt_kernel: random worlds, goals near and
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
