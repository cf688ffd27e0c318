import sys, json
sys.path.insert(0, '.')
import bench, torch
from auv_sim_amd import _lib
class R:
    rank=0; world=1
    def sync(self): torch.cuda.synchronize()
    def max_time(self,dt): return dt
    def sum(self,v): return float(v)
ctx=_lib.Context(0)
for nf in (12, 25, 50, 100):
    r=bench.bench_config5(ctx, R(), n_filters=nf)
    print(nf*500, r["value"], r["plan_launch_ms"], r["ms_per_tracking_step"], r["episodes_done_last_step"])
