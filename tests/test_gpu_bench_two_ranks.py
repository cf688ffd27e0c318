"""bench.py's N > 1 path on a ONE-GPU box: `--gpus 2` under AUVP_BENCH_ONE_GPU=1 starts two ranks (fresh children of a
fresh child of this test: nothing that has touched the GPU is ever exec'ed) that share GPU 0 and talk over gloo -- RCCL
cannot put two ranks on one device -- so spawn_ranks, the episode sharding (seed = global episode id), the sharded side
measurements (configs 3, 4, 5), the result gathers and the per-rank fields of the JSON line all run where the driver
looks.  The RCCL transport itself is covered by tests/test_gpu_gather_rccl.py (world size 1 here, 2 on a 2-GPU box)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(gpus, episodes, extra=(), one_gpu=False, more_env=None):
    """-> the full record (bench_sides.json) with the parsed final stdout line under "_line" """
    import tempfile
    env = dict(os.environ)
    env.update(more_env or {})
    sides = os.path.join(tempfile.mkdtemp(prefix="auvp_bench_"), "bench_sides.json")
    env["AUVP_BENCH_SIDES"] = sides
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    if one_gpu:
        env["AUVP_BENCH_ONE_GPU"] = "1"
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", str(gpus), "--steps", "1", "--warmup", "0",
           "--episodes", str(episodes), "--iters", "300", "--no-cpu"] + list(extra)
    r = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 prints ONE JSON line, got %d" % len(lines)
    assert r.stdout.rstrip().splitlines()[-1] == lines[0], "the JSON line is the LAST line of stdout"
    assert len(lines[0]) <= 4096, "the line the driver parses stays under 4 KB (%d)" % len(lines[0])
    line = json.loads(lines[0])
    assert line["sides_file"] == "bench_sides.json"
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    rf = line["roofline"]
    assert rf["frac"] > 0 and rf["achieved"] > 0 and rf["peak"] == 8000.0 and "traffic" in rf and rf["kernel_ms"] > 0
    assert all(not isinstance(v, (dict, list)) for v in rf.values())
    full = json.load(open(sides))
    assert abs(full["value"] / line["value"] - 1.0) < 1e-6 and full["n_gpus"] == line["n_gpus"]
    full["_line"] = line
    return full


def test_two_ranks_on_one_gpu_shard_the_batch_and_report_per_rank():
    two = _bench(2, 64, one_gpu=True)
    assert two["n_gpus"] == 2 and two["scaling"] == "weak"
    assert two["config"]["parallelism"] == "episodes sharded x2"
    assert two["config"]["episodes_per_gpu"] == 64
    assert len(two["kernel_ms_per_rank"]) == 2 and all(k > 0 for k in two["kernel_ms_per_rank"])
    assert len(two["gather_ms_per_rank"]) == 2
    assert "gloo" in (two["config"]["gather"] or "") and "AUVP_BENCH_ONE_GPU" in (two["config"]["gather_note"] or "")
    assert len(two["config"]["workload"]) <= 120 and "parent sampling" in two["config"]["workload"]
    # the sharded side measurements ran on both ranks and report per rank
    for side, field in (("astar", "search_launch_ms_per_rank"), ("planner_rrt", "plan_launch_ms_per_rank")):
        assert "error" not in two[side], two[side]
        assert len(two[side][field]) == 2
    assert two["astar"]["instances_this_rank"] == 512 and two["planner_rrt"]["episodes_this_rank"] == 256
    assert "error" not in two["config5"], two["config5"]
    # the same 128 episodes (seeds = global episode ids 0..127) on one rank: same expansions, same trees
    one = _bench(1, 128, extra=["--no-extra"])
    assert one["n_gpus"] == 1 and one["config"]["parallelism"] == "episodes sharded x1"
    assert two["expansions_per_step"] == one["expansions_per_step"] == 128 * 300
    assert two["accepted_nodes_per_step"] == one["accepted_nodes_per_step"] > 0
    # roofline object: flat scalars the driver's record keeps
    rf = one["roofline"]
    for k in ("frac", "kernel_ms", "leaf_kernel_ms", "leaf_compulsory_bytes", "leaf_frac", "pass_8d_frac", "hbm_measured_GBps",
              "frac_of_measured"):
        assert isinstance(rf[k], float) and rf[k] > 0, k
    assert rf["frac"] <= 1.0 and rf["leaf_frac"] <= 1.0
    assert 1000.0 < rf["hbm_measured_GBps"] < 8000.0  # a measured HBM read rate, below the datasheet peak


def test_eight_ranks_on_one_gpu_spawn_shard_and_gather():
    """the target's world size through bench.py's own spawn path: eight fresh ranks share GPU 0 (gloo transport), eight episodes
    each; the gathered result covers the 64 global episode ids and equals one rank running all 64"""
    eight = _bench(8, 8, extra=["--no-extra"], one_gpu=True)
    assert eight["n_gpus"] == 8 and eight["config"]["parallelism"] == "episodes sharded x8" and eight["_line"]["n_gpus"] == 8
    assert len(eight["kernel_ms_per_rank"]) == 8 and len(eight["gather_ms_per_rank"]) == 8
    one = _bench(1, 64, extra=["--no-extra"])
    assert eight["expansions_per_step"] == one["expansions_per_step"] == 64 * 300
    assert eight["accepted_nodes_per_step"] == one["accepted_nodes_per_step"] > 0


@pytest.mark.parametrize("mode", ["root", "all"])
def test_two_ranks_through_the_c_abi_transport_with_a_stand_in_rccl(tmp_path, mode):
    """bench.py's RCCL-mode code path -- RcclGather over libauvplan.so's auvp_gather* entry points, device pointers, the gather
    to rank 0 enqueued on the gather stream and ended after the next step's kernels (or AUVP_BENCH_GATHER=all: the all-gathers
    inside the step) -- with two ranks on one GPU: the C-ABI binds tests/mock_rccl (AUVP_RCCL_LIBRARY) instead of RCCL, which
    cannot put two ranks on one device"""
    mock = str(tmp_path / "libmock_rccl.so")
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "-O1", "-o", mock, os.path.join(REPO, "tests", "mock_rccl", "mock_rccl.cpp"),
                        "-lpthread"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    two = _bench(2, 64, extra=["--no-extra", "--steps", "3", "--warmup", "1"], one_gpu=True,
                 more_env={"AUVP_RCCL_LIBRARY": mock, "AUVP_BENCH_GATHER": mode})
    assert two["n_gpus"] == 2 and "rccl (auvp_gather, C-ABI)" in two["config"]["gather"]
    assert two["config"]["rccl_library"] == mock and two["config"]["rccl_ranks_seen"] == 2
    assert two["config"]["rccl_comm_info_per_rank"] == [[2, 0, 2], [2, 1, 2]]
    assert len(two["gather_ms_per_rank"]) == 2 and all(g is not None and g >= 0 for g in two["gather_ms_per_rank"])
    assert all(b > 64 * 100 for b in two["gather_bytes_per_rank"])
    if mode == "root":
        assert two["config"]["gather_mode"] == "to rank 0, overlapped"
        recs, elems = two["gather_root_received"]   # what rank 0 held after the last step: both ranks' records and path elements
        assert recs == 128 and elems > 128
        assert two["gather_bytes_per_rank"][0] > two["gather_bytes_per_rank"][1]   # the root counts what it received, the other what it sent
    else:
        assert two["config"]["gather_mode"] == "all-gather in step" and two["gather_root_received"] is None
    one = _bench(1, 128, extra=["--no-extra"])
    assert two["expansions_per_step"] == one["expansions_per_step"] == 128 * 300
    assert two["accepted_nodes_per_step"] == one["accepted_nodes_per_step"] > 0


def test_eight_ranks_through_the_c_abi_transport_with_a_stand_in_rccl(tmp_path):
    """the target's world size through the RCCL-mode path: eight ranks on one GPU, the gather of every step to rank 0 on the
    gather stream (seven grouped receives on the root, one send on every other rank), results equal one rank running all 64"""
    mock = str(tmp_path / "libmock_rccl.so")
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "-O1", "-o", mock, os.path.join(REPO, "tests", "mock_rccl", "mock_rccl.cpp"),
                        "-lpthread"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    eight = _bench(8, 8, extra=["--no-extra", "--steps", "2", "--warmup", "1"], one_gpu=True, more_env={"AUVP_RCCL_LIBRARY": mock})
    assert eight["n_gpus"] == 8 and eight["config"]["rccl_ranks_seen"] == 8 and "C-ABI" in eight["config"]["gather"]
    assert eight["config"]["rccl_comm_info_per_rank"] == [[8, i, 8] for i in range(8)]
    assert eight["config"]["gather_mode"] == "to rank 0, overlapped" and eight["gather_root_received"][0] == 64
    one = _bench(1, 64, extra=["--no-extra"])
    assert eight["expansions_per_step"] == one["expansions_per_step"] == 64 * 300
    assert eight["accepted_nodes_per_step"] == one["accepted_nodes_per_step"] > 0
