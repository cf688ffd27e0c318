timeout 900 python -m pytest tests/test_gpu_planner_rrt.py tests/test_gpu_dropin_planner.py tests/test_gpu_config5.py tests/test_gpu_rrt_env.py -x -q 2>&1 | grep -n "passed\|failed\|Error\|assert\|no tests" | tail -8
timeout 300 python bench.py --only planner_rrt,config5 --no-cpu 2>/dev/null | tail -1 > gpurun_out/prrt_b.json; python3 -c "
import json;d=json.load(open('gpurun_out/prrt_b.json'));p=d['planner_rrt'];print('planner',p['value'],p['ms_per_step'],p['plan_launch_ms']);c=d['config5'];print('config5',c['value'],c['ms_per_tracking_step'],c['plan_launch_ms'])"
