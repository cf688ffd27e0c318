#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + PMC passes of bench.py.
# usage: tools/profile_bench.sh <tag> [bench args...]; writes gpurun_out/prof_<tag>/
#   trace_main   headline kernel alone (--no-extra): its average must agree with roofline.kernel_ms
#   trace        the full default command (every side measurement)
#   pmc_<C>      counters of the headline launch, one pass per counter group (counters only: no trace flags)
#   pmc_astar_<C> / pmc_planner_<C>   the same for config 3 (astar_kernel) and config 4 (prrt_kernel)
set -u
TAG=${1:-r2}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$@"
python3 $R/bench.py $ARGS > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_main -o trace_main -- python3 $R/bench.py $ARGS --no-cpu --no-extra > $OUT/trace_main_bench.json 2> $OUT/trace_main.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $R/bench.py $ARGS --no-cpu > $OUT/trace_bench.json 2> $OUT/trace.err
GROUPS_=("FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE")
for P in "${GROUPS_[@]}"; do
  N=$(echo $P | cut -d" " -f1)
  rocprofv3 --pmc $P --output-format csv -d $OUT/pmc_$N -o pmc -- python3 $R/bench.py $ARGS --no-cpu --no-extra --steps 1 --warmup 0 > $OUT/pmc_$N.json 2> $OUT/pmc_$N.err
done
for SIDE in astar planner_rrt; do
  for P in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
    N=$(echo $P | cut -d" " -f1)
    rocprofv3 --pmc $P --output-format csv -d $OUT/pmc_${SIDE}_$N -o pmc -- python3 $R/bench.py --no-cpu --no-variants --only $SIDE > $OUT/pmc_${SIDE}_$N.json 2> $OUT/pmc_${SIDE}_$N.err
  done
done
find $OUT -name "*.csv" | head -40
