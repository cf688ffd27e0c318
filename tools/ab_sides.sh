#!/bin/bash
# A/B of library builds on the side measurements only: SIDES=a,b tools/ab_sides.sh "" exp/libx.so ...
for lib in "$@"; do
  echo "== lib=$lib"
  AUVPLAN_LIBRARY=$lib python bench.py --steps 5 --warmup 2 --no-cpu --only ${SIDES:-planner_rrt,config5,particle_filter} 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k,v in d.items(): print(' ', k, v.get('value'), {a:b for a,b in v.items() if a.endswith('_ms') and isinstance(b,(int,float))})"
done
