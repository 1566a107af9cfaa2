"""Where the mapper loop's device allocations happen: per keyframe, the number of hipMalloc calls the caching allocator
made (num_device_alloc), the bytes it reserved, the map size - for the first pass over the keyframes and for a second
pass in the same process (allocator warm).  usage: python profiles/experiments/mapper_mallocs.py [H W]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from active_gs_amd import env_config  # noqa: E402
env_config.apply_env(os.environ)
from active_gs_amd.gaussian_map import GaussianMap  # noqa: E402
from active_gs_amd.synthetic import make_keyframes, mapper_cfg  # noqa: E402

dev = torch.device("cuda:0")
h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (512, 512)
frames = make_keyframes(50, h, w, dev)
torch.cuda.synchronize()
for rep in range(2):
    gm = GaussianMap(mapper_cfg(10, "device"), dev)
    rows = []
    for k, f in enumerate(frames):
        s0 = torch.cuda.memory_stats(dev)
        gm.update(f)
        s1 = torch.cuda.memory_stats(dev)
        d = int(s1["num_device_alloc"] - s0["num_device_alloc"])
        if d:
            rows.append(dict(keyframe=k, surfels=gm.num_gaussians, mallocs=d,
                             reserved_mb=round((s1["reserved_bytes.all.current"] - s0["reserved_bytes.all.current"]) / 2 ** 20, 1)))
    gm.settle()
    torch.cuda.synchronize()
    print(json.dumps(dict(pass_=rep, total_mallocs=sum(r["mallocs"] for r in rows), keyframes_with_mallocs=rows,
                          reserved_gb=round(torch.cuda.memory_stats(dev)["reserved_bytes.all.current"] / 2 ** 30, 2))))
    del gm
