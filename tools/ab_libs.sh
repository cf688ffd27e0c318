#!/bin/bash
# A/B of library builds on one box: tools/ab_libs.sh "" exp/libauvplan_x.so ...   (headline kernels + the main sides)
for lib in "$@"; do
  echo "== lib=$lib"
  AUVPLAN_LIBRARY=$lib python bench.py --steps 6 --warmup 2 --no-extra --no-cpu 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('rows', r['kernel_ms'], 'leaf', r['leaf_kernel_ms'], 'value', d['value'])"
  AUVPLAN_LIBRARY=$lib python bench.py --steps 5 --warmup 2 --no-cpu --only ${SIDES:-planner_rrt,config5,particle_filter,rrt_1024_replicas,single_episode} 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k,v in d.items(): print(' ', k, v.get('value'), v.get('ms_per_step'), {a:b for a,b in v.items() if a.endswith('_ms') and isinstance(b,(int,float))})"
done
