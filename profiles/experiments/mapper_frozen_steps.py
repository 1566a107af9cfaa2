#!/usr/bin/env python3
"""A FROZEN mapper-shaped workload for A/B runs of library builds: a map grown by the mapper loop (GaussianMap.update over
30 keyframes @512x512, ~150 k spatially coherent surfels) and the batched training iteration of configuration 3 (11 views
per iteration) with all learning rates 0, so that two builds see exactly the same scene whatever their gradients are.
  python profiles/experiments/mapper_frozen_steps.py make /tmp/frozen_map.pt      (product build: grows and saves the map)
  AGS_LIB_PATH=scratch/libags_<tag>.so python profiles/experiments/mapper_frozen_steps.py run /tmp/frozen_map.pt [iterations]
"""
import os
import sys

import numpy as np
import torch

R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, R)
from active_gs_amd import env_config
env_config.apply_env(os.environ)   # the package itself reads no environment variable
from active_gs_amd.fused_map_trainer import FusedMapTrainer  # noqa: E402
from active_gs_amd.gaussian_map import GaussianMap  # noqa: E402
from active_gs_amd.synthetic import make_keyframes, mapper_cfg  # noqa: E402

dev = torch.device("cuda:0")
mode, path = sys.argv[1], sys.argv[2]
if mode == "make":
    # AGS_FROZEN_ROOM / AGS_FROZEN_KEYFRAMES: another room stand-in / a longer mission (a larger map of which a view shows less)
    frames = make_keyframes(int(os.environ.get("AGS_FROZEN_KEYFRAMES", "30")), 512, 512, dev, gt_surfels=int(os.environ.get("AGS_FROZEN_GT", "400000")),
                            room=os.environ.get("AGS_FROZEN_ROOM", "office0"))
    np.random.seed(0)
    gm = GaussianMap(mapper_cfg(10, "device"), dev)
    for f in frames:
        gm.update(f)
    tr = gm._trainer
    keys = ("means", "scales", "rotations", "opacities", "harmonics", "view_scores", "view_supports", "view_means")
    torch.save(dict(raw={k: getattr(tr, k).detach().cpu().clone() for k in keys}, perf=tr.training_performance.cpu().clone(),
                    frames=[{k: (v.cpu() if torch.is_tensor(v) else v) for k, v in f.items() if not k.startswith("_")} for f in tr.frames],
                    cfg={k: v for k, v in tr.cfg.items()}), path)
    print("saved", path, tr.means.shape[0], "surfels")
else:
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    d = torch.load(path, weights_only=False)
    raw = {k: v.to(dev) for k, v in d["raw"].items()}
    frames = [{k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in f.items()} for f in d["frames"]]
    cfg = dict(d["cfg"], lrs=dict(mean=0.0, scale=0.0, rotation=0.0, opacity=0.0, harmonic=0.0), optimization_steps=iters,
               prune_interval=10 ** 9, sampler="device")
    tr = FusedMapTrainer(raw, frames, cfg)
    tr.training_performance = d["perf"].to(dev)
    np.random.seed(1); torch.manual_seed(1)
    assert tr._uniform_frames()
    st0 = None
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        assert tr._train_batched(iters) is True
        e1.record(); torch.cuda.synchronize()
        b = tr._batched_cache["batch"]
        vis = b.statuses()[:, 3].float().mean().item() if b is not None else float("nan")
        print("ms/iteration %.4f" % (e0.elapsed_time(e1) / iters), tr.means.shape[0], "surfels, visible per view %.0f (%.1f %%)" % (vis, 100 * vis / tr.means.shape[0]), flush=True)
