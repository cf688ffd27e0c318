#!/bin/bash
# usage (on the GPU box): tools/pmc_probe.sh <tag> <phase_probe args...>   -> gpurun_out/pmcp_<tag>/
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmcp_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $OUT/a -o pmc -- python3 $R/tools/phase_probe.py --clocks 0 "$@" > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/b -o pmc -- python3 $R/tools/phase_probe.py --clocks 0 "$@" > $OUT/b.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(out + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "rrt_explore" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print("%-22s %.4g (n=%d)" % (k, sum(acc[k]) / len(acc[k]), len(acc[k])))
PY
grep kernel $OUT/a.log
