"""Wall time of the mapper loop's phases (examples/mapper_loop.py with synchronising timers around FusedMapTrainer /
densify entry points): python profiles/experiments/mapper_phases.py"""
import sys, os, time, json, collections
sys.path.insert(0, os.getcwd())
import numpy as np, torch
sys.argv = [sys.argv[0]]
import importlib.util
spec = importlib.util.spec_from_file_location("ml", "examples/mapper_loop.py"); ml = importlib.util.module_from_spec(spec); spec.loader.exec_module(ml)
from active_gs_amd import fused_map_trainer as fmt, densify, map_trainer as mt
acc = collections.defaultdict(float); cnt = collections.defaultdict(int)
def wrap(obj, name, label=None):
    fn = getattr(obj, name); label = label or name
    def w(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize(); acc[label] += time.perf_counter() - t0; cnt[label] += 1
        return r
    setattr(obj, name, w)
T = fmt.FusedMapTrainer
for n in ("add_gaussians", "_train_batched", "post_processing", "_render_counts", "prune", "_snapshot", "_make_camera"):
    wrap(T, n)
for n in ("smooth_depth", "candidates", "voxel_select", "compact_plan"):
    wrap(densify, n, "densify." + n)
wrap(densify, "add_gaussians", "densify.add_gaussians"); wrap(densify, "prune", "densify.prune")
ml.main()
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"{k:28s} {1e3 * v:8.1f} ms total  {cnt[k]:4d} calls  {1e3 * v / cnt[k]:7.3f} ms/call")
