import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
from _scenes import room_case
import test_gpu_parity as T
names = {0: "means3D", 1: "means2D", 2: "opacities", 4: "colors", 5: "scales", 6: "rotations"}
worst_i, worst_g = {}, {}
for (n, h, w, view, mult) in [(3000, 120, 160, 0, 3.0), (5000, 170, 300, 1, 2.0), (800, 64, 64, 2, 4.0), (2000, 100, 150, 3, 3.0)]:
    a, S = room_case(n, h, w, view=view, seed=view, scale_mult=mult)
    ins, ref, gin, out = T._run_both(a, S, seed=view)
    for nm, r, o in zip(["rgb", "normal", "depth", "opacity", "confidence"], ref[:5], out[:5]):
        worst_i[nm] = max(worst_i.get(nm, 0), (o.cpu() - r.detach()).abs().mean().item())
    for i, nm in names.items():
        r, o = ins[i].grad, gin[i].grad.cpu()
        worst_g[nm] = max(worst_g.get(nm, 0), (o - r).abs().sum().item() / max(r.abs().sum().item(), 1e-12))
print(os.environ.get("AGS_BWD_MFMA", "default"), {k: float("%.2g" % v) for k, v in worst_i.items()}, {k: float("%.2g" % v) for k, v in worst_g.items()})
