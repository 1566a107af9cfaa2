"""Does the ORDER of the surfels in memory matter?  The synthetic rooms are sampled uniformly at random, so
neighbours in memory are anywhere in the room; a map grown by the reference (gaussian_map.py:294-468: one keyframe's
pixels in raster order, appended) is spatially coherent.  This probe times the optimisation step on the same scenes
with the rows permuted into Morton order of their means (a permutation: same surfels, same images).
usage: python profiles/experiments/morton_probe.py   (GPU box)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from active_gs_amd import env_config
env_config.apply_env(os.environ)   # the package itself reads no environment variable
import active_gs_amd.synthetic as syn

def morton_perm(p, bits=10):
    lo, hi = p.min(0).values, p.max(0).values
    q = ((p - lo) / (hi - lo).clamp_min(1e-9) * (2 ** bits - 1)).long().clamp(0, 2 ** bits - 1)
    code = torch.zeros(p.shape[0], dtype=torch.long)
    for b in range(bits):
        for a in range(3):
            code |= ((q[:, a] >> b) & 1) << (3 * b + a)
    return torch.argsort(code)

orig = syn.make_room_scene
def sorted_scene(n, room="office0", seed=0):
    raw = orig(n, room, seed)
    perm = morton_perm(raw["means"])
    return {k: v[perm].contiguous() for k, v in raw.items()}

dev = torch.device("cuda:0")
out = {}
for order in ("random", "morton"):
    syn.make_room_scene = orig if order == "random" else sorted_scene
    out[order] = {
        "c2": bench.measure_config("c2", 200_000, 680, 1200, 1, "office0", 40, dev),
        "c4_share": bench.measure_config("c4", 1_500_000, 680, 1200, 4, "room0", 20, dev),
        "c5": bench.measure_config("c5", 5_000_000, 2048, 2048, 1, "office0", 20, dev)}
    for k, v in out[order].items():
        print(order, k, v["ms_per_step"], v["stage_ms_per_view"], flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r03_morton_probe.json"), "w"), indent=1)
