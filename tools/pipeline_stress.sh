#!/bin/bash
# Stress of the speculative multi-wavefront kernels under the diagnostic build (libauvplan_diag.so, -DAUVP_PIPE_DIAG):
# a pseudo-random delay (0 .. $UNITS x s_sleep 1, ~64 clocks each) before every hand-over word of ONE stage at a time, then
# bounded waits that run out (AUVP_DIAG_SPIN) so that the pipeline fallback redoes episodes.  Runs on the GPU box:
#   gpurun -- 'bash tools/pipeline_stress.sh 300 > gpurun_out/pipeline_stress.txt 2>&1'
# One summary line per run; "0 mismatches" everywhere is the pass criterion (tests/test_gpu_soak.py runs a short form).
N=${1:-300}
UNITS=${2:-32}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export AUVPLAN_LIBRARY=$R/auv_sim_amd/libauvplan_diag.so
cd $R
run() {  # label, jitter spec, spin, command...
  local label=$1 jit=$2 spin=$3; shift 3
  if [ -n "$jit" ]; then export AUVP_DIAG_JITTER=$jit; else unset AUVP_DIAG_JITTER; fi
  if [ -n "$spin" ]; then export AUVP_DIAG_SPIN=$spin; else unset AUVP_DIAG_SPIN; fi
  local t0=$(date +%s)
  local out=$("$@" 2>&1 | tail -1)
  echo "$label | jitter=${jit:-none} spin=${spin:-default} | $out | $(( $(date +%s) - t0 )) s"
}
ONLY=${3:-all}   # "planner": only the prrt_pipe runs
if [ "$ONLY" = "all" ]; then
# RRT.exploring: rrt_duo (wave & 1: 0 main, 1 helper), rrt_trio (wave % 3: 0 M, 1 H, 2 T; the four-wavefront form of the same
# sweep has wave % 4: 0 M, 1 H, 2 T, 3 L)
for spec in "1,2,0" "1,2,1" "1,3,0" "1,3,1" "1,3,2" "1,4,0" "1,4,1" "1,4,2" "1,4,3"; do
  run "rrt duo/trio/quad (soak_duo.py $N cases)" "$spec,$UNITS,5" "" python tests/experiments/soak_duo.py $N 31
done
fi
# Planner_RRT.planning: prrt_pipe (round 6: wave % 5: 0 M, 1 H, 2 S, 3 G, 4 D the draw wavefront; the script's every other
# repetition runs the four-wavefront form, wave & 3, where 1,5,r lands on varying roles, and 1,4,r the other way round)
for spec in "1,5,0" "1,5,1" "1,5,2" "1,5,3" "1,5,4" "1,4,0" "1,4,1" "1,4,2" "1,4,3"; do
  run "prrt_pipe (soak_planner_duo.py $N cases)" "$spec,$UNITS,5" "" python tests/experiments/soak_planner_duo.py $N 32
done
if [ "$ONLY" = "all" ]; then
# astar_fixLenSOG, paired form: wavefronts 0..3 search, 4..7 are the partners (wave / 4)
for spec in "2,4,0" "2,4,1"; do
  run "astar pair (tests/test_gpu_astar.py + dropin)" "$spec,$UNITS,5" "" python -m pytest tests/test_gpu_astar.py tests/test_gpu_dropin_astar.py tests/test_gpu_full_size.py -q -m gpu -k "astar or SOG or sog or fixLen"
done
fi
# waits that run out: the fallback redoes the episodes / instances on the one-wavefront kernels
if [ "$ONLY" = "all" ]; then
run "rrt duo/trio/quad, spin 6" "1,3,1,$UNITS,5" 6 python tests/experiments/soak_duo.py $N 33
run "rrt duo/trio/quad, spin 40" "1,3,2,$UNITS,5" 40 python tests/experiments/soak_duo.py $N 34
fi
run "prrt_pipe, spin 6" "1,5,1,$UNITS,5" 6 python tests/experiments/soak_planner_duo.py $N 35
run "prrt_pipe, spin 40" "1,5,4,$UNITS,5" 40 python tests/experiments/soak_planner_duo.py $N 36
run "prrt_pipe, spin 100" "1,5,3,$UNITS,5" 100 python tests/experiments/soak_planner_duo.py $N 37
[ "$ONLY" = "all" ] && run "astar pair, spin 6" "2,4,1,$UNITS,5" 6 python -m pytest tests/test_gpu_astar.py tests/test_gpu_dropin_astar.py -q -m gpu
