#!/usr/bin/env python3
"""Where one keyframe of the mapper loop (config 3) spends its time, phase by phase, WITHOUT a profiler: HIP events are
recorded on the stream at the phase boundaries (GPU-timeline time between them, idle included) next to the host's clock
at the same points (how long the host took to enqueue the phase).  A phase whose GPU time is about its host time is
host-bound (the GPU waits for launches); the training iterations' GPU time is kernel time.  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    from active_gs_amd import env_config
    env_config.apply_env(os.environ)   # the package itself reads no environment variable
    from active_gs_amd import fused_map_trainer as fmt
    from active_gs_amd import raster_api as api
    from active_gs_amd.gaussian_map import GaussianMap
    from active_gs_amd.synthetic import make_keyframes, mapper_cfg
    dev = torch.device("cuda:0")
    frames = make_keyframes(50, 512, 512, dev, gt_surfels=400_000)
    marks = []

    def mark(label):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        marks.append((label, e, time.perf_counter()))

    def wrap(cls, name, before, after):
        fn = getattr(cls, name)

        def inner(*a, **k):
            mark(before)
            try:
                return fn(*a, **k)
            finally:
                mark(after)
        setattr(cls, name, inner)

    T = fmt.FusedMapTrainer
    wrap(T, "add_gaussians", "grow (densify render, candidates, append rows)", "between")
    wrap(T, "_snapshot", "snapshot", "between")
    wrap(T, "_post_processing_begin", "count render enqueue", "settle (the wait) + between")
    wrap(T, "_post_processing_end", "view stats / prune", "between")
    first = {"pending": False}
    orig_tb = T._train_batched

    def tb(self, *a, **k):
        mark("train set-up (optimiser, row set, confidences, bind)")
        first["pending"] = True
        try:
            return orig_tb(self, *a, **k)
        finally:
            mark("between")
    T._train_batched = tb
    orig_fwd = api.ViewBatch.forward

    def fwd(self, *a, **k):
        if first["pending"]:
            first["pending"] = False
            mark("iterations (from the first batched forward)")
        return orig_fwd(self, *a, **k)
    api.ViewBatch.forward = fwd

    warm = GaussianMap(mapper_cfg(10, "device"), dev)
    for f in frames[:2]:
        warm.update(f)
    del warm
    torch.cuda.synchronize()
    marks.clear()
    np.random.seed(0)
    gm = GaussianMap(mapper_cfg(10, "device"), dev)
    t0 = time.perf_counter()
    for f in frames:
        gm.update(f)
    mark("end")
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gpu, host, cnt = {}, {}, {}
    for (la, ea, ta), (_, eb, tb_) in zip(marks[:-1], marks[1:]):
        gpu[la] = gpu.get(la, 0.0) + ea.elapsed_time(eb)
        host[la] = host.get(la, 0.0) + 1e3 * (tb_ - ta)
        cnt[la] = cnt.get(la, 0) + 1
    rows = {k: dict(gpu_timeline_ms=round(gpu[k], 2), host_enqueue_ms=round(host[k], 2), times=cnt[k]) for k in gpu}
    print(json.dumps(dict(workload="mapper loop, 50 keyframes x 10 iterations @512x512 (event marks add ~25 launches per keyframe)",
                          seconds=round(dt, 4), gpu_timeline_total_ms=round(sum(gpu.values()), 2), phases=rows)))


if __name__ == "__main__":
    main()
