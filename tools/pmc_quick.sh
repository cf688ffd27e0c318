cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/q1 -o pmc -- python3 $R/bench.py --no-cpu --no-extra --steps 1 --warmup 0 > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/q2 -o pmc -- python3 $R/bench.py --no-cpu --no-extra --steps 1 --warmup 0 > /dev/null 2>&1
python3 - <<'PY'
import csv,glob,collections,os
R=os.environ["GRAFT_REPO_ROOT"]
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(R+"/gpurun_out/q?/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        k="explore" if "rrt_explore" in r["Kernel_Name"] or "rrt_rows" in r["Kernel_Name"] else ("leaf" if "rrt_leaf" in r["Kernel_Name"] else None)
        if k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in acc:
    print(k)
    for c in sorted(acc[k]): print("  %-22s %.5g"%(c,sum(acc[k][c])/len(acc[k][c])))
PY
