#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + PMC passes of bench.py.
# usage: tools/profile_bench.sh <tag> [--sides "a b c"] [--no-headline] [bench args...]; writes gpurun_out/prof_<tag>/
#   bench.json     the plain default command (what the driver runs): its compact final line; bench_sides.json = the full record
#   trace_main     headline kernel alone (--no-extra): its average must agree with roofline.kernel_ms
#   trace          the full default command (every side measurement)
#   pmc@headline@<C>   counters of the headline launch, one pass per counter group (counters only: no trace flags); one warm-up
#                      step + one step: the first batch with a parameter block runs rrt_rows_kernel, the measured ones the stream kernels
#   trace_<side>, pmc@<side>@<C>   the same for one side measurement (bench.py --only <side>); default sides below
set -u
TAG=${1:-r3}; shift || true
SIDES="astar planner_rrt rrt_nn rrt_nn_long_horizon config5 single_episode rrt_1024_replicas particle_filter shark_grid"
HEADLINE=1
while [ $# -gt 0 ]; do
  case "$1" in
    --sides) SIDES="$2"; shift 2;;
    --no-headline) HEADLINE=0; shift;;
    *) break;;
  esac
done
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export AUVP_BENCH_PROFILE=1   # bench.py: no auxiliary launches of a measured kernel (the NN leg's half-budget run) under the profiler
ARGS="$@"
G_FETCH="FETCH_SIZE"
G_WRITE="WRITE_SIZE"
G_INSTS="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
G_ACTIVE="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
G_LANES="SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
if [ $HEADLINE = 1 ]; then
  # bench.py prints the compact (<= 4 KB) line; the full record with every side measurement goes to $AUVP_BENCH_SIDES
  export AUVP_BENCH_SIDES=$OUT/bench_sides.json
  AUVP_BENCH_PROFILE=0 python3 $R/bench.py $ARGS > $OUT/bench.json 2> $OUT/bench.err
  export AUVP_BENCH_SIDES=$OUT/trace_main_bench.sides.json
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_main -o trace_main -- python3 $R/bench.py $ARGS --no-cpu --no-extra > $OUT/trace_main_bench.json 2> $OUT/trace_main.err
  export AUVP_BENCH_SIDES=$OUT/trace_bench.sides.json
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $R/bench.py $ARGS --no-cpu > $OUT/trace_bench.json 2> $OUT/trace.err
  for P in "$G_FETCH" "$G_WRITE" "$G_INSTS" "$G_ACTIVE" "$G_LANES"; do
    N=$(echo $P | cut -d" " -f1)
    export AUVP_BENCH_SIDES=$OUT/pmc@headline@$N.sides.json
    rocprofv3 --pmc $P --output-format csv -d $OUT/pmc@headline@$N -o pmc -- python3 $R/bench.py $ARGS --no-cpu --no-extra --steps 1 --warmup 1 > $OUT/pmc@headline@$N.json 2> $OUT/pmc@headline@$N.err
  done
fi
for SIDE in $SIDES; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$SIDE -o trace_$SIDE -- python3 $R/bench.py --no-cpu --no-variants --only $SIDE > $OUT/trace_$SIDE.json 2> $OUT/trace_$SIDE.err
  for P in "$G_FETCH" "$G_WRITE" "$G_INSTS" "$G_ACTIVE"; do
    N=$(echo $P | cut -d" " -f1)
    rocprofv3 --pmc $P --output-format csv -d $OUT/pmc@$SIDE@$N -o pmc -- python3 $R/bench.py --no-cpu --no-variants --only $SIDE > $OUT/pmc@$SIDE@$N.json 2> $OUT/pmc@$SIDE@$N.err
  done
done
find $OUT -name "*.csv" | wc -l
