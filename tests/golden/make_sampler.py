#!/usr/bin/env python3
"""Generates tests/golden/sampler.pt.  Run ONLY in the build container (needs /root/reference).

Drives the reference's two frame samplers (/root/reference/mapping/utils.py:190-261) on seeded inputs and records
which frames they pick: UniformSampler (torch.randperm on torch's global stream; its frame container is a dict) and
WeightedSampler (np.random.choice on numpy's global stream).  Only inputs and picked ids are stored.
"""
import os
import sys
from unittest.mock import MagicMock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


class AttrDict(dict):
    __getattr__ = dict.__getitem__


def main():
    # the samplers live in mapping/utils.py; importing it runs mapping/__init__.py, which pulls in the whole mapper:
    # its absent third-party imports are stubbed (as in make_golden.py), none of them is touched by the samplers
    for name in ["jaxtyping", "cv2", "trimesh", "torchvision", "torchvision.transforms", "open3d", "torchmetrics",
                 "torchmetrics.image", "torchmetrics.image.lpip", "imgviz", "PIL", "PIL.Image",
                 "diff_gaussian_rasterization_2d"]:
        sys.modules[name] = MagicMock()
    sys.path.insert(0, REF)
    import mapping.utils as mu
    cases = []
    for case, (n_frames, batch, active, seed) in enumerate([(12, 8, 3, 5), (4, 8, 3, 6), (3, 8, 3, 7), (20, 6, 2, 8)]):
        g = torch.Generator().manual_seed(100 + case)
        frames = [dict(rgb=torch.rand(3, 4, 5, generator=g), depth=torch.rand(1, 4, 5, generator=g),
                       extrinsic=torch.rand(4, 4, generator=g), intrinsic=torch.rand(3, 3, generator=g))
                  for _ in range(n_frames)]
        cfg = AttrDict(batch_size=batch, active_size=active)
        # uniform: frames as a dict keyed by frame number (the sampler calls .keys()), three draws in a row
        torch.manual_seed(seed)
        us = mu.UniformSampler(cfg, {10 * i: f for i, f in enumerate(frames)})
        uni = []
        for _ in range(3):
            rgbs, depths, extr, intr = us.next_frames()
            uni.append(dict(rgb_sum=rgbs.sum(dim=(1, 2, 3)), n=rgbs.shape[0], extr0=extr[:, 0, 0].clone()))
        # weighted: frames as a list, weights = seeded per-frame errors
        np.random.seed(seed)
        ws = mu.WeightedSampler(cfg, frames)
        weight = torch.rand(n_frames, generator=g) + 0.1
        wei = []
        for _ in range(3):
            (rgbs, depths, extr, intr), ids = ws.next_frames(weight.clone())
            wei.append(dict(rgb_sum=rgbs.sum(dim=(1, 2, 3)), n=rgbs.shape[0], ids=torch.as_tensor(np.asarray(ids))))
        cases.append(dict(n_frames=n_frames, batch=batch, active=active, seed=seed, frames=frames, weight=weight,
                          uniform=uni, uniform_v=us.v, weighted=wei, weighted_v=ws.v))
    torch.save(cases, os.path.join(HERE, "sampler.pt"))
    print("wrote sampler.pt:", [(c["n_frames"], c["uniform_v"], c["weighted_v"]) for c in cases])


if __name__ == "__main__":
    main()
