#!/bin/bash
# A/B of library builds on the headline only: tools/ab_headline.sh "" exp/libx.so ...
for lib in "$@"; do
  echo -n "== lib=$lib  "
  AUVPLAN_LIBRARY=$lib python bench.py --steps 6 --warmup 2 --no-extra --no-cpu 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('rows', r['kernel_ms'], 'leaf', r['leaf_kernel_ms'], 'value', d['value'])"
done
