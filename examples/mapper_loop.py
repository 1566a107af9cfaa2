#!/usr/bin/env python3
"""Configuration 3 of BASELINE.json in one script: the full mapper loop (grow from each new RGB-D keyframe, train,
post-process / prune) for 500 optimisation iterations on one MI355X.

Drives the drop-in class the way /root/reference/mapping/mapper.py:44,98-104 drives the reference's:
``GaussianMap(cfg.gaussian_map, device)`` once, ``gaussian_map.update(dataframe)`` per keyframe, with the values of
config/mapper/incremental.yaml (10 iterations per keyframe, batch 8 with 3 active frames, prune every 5th keyframe) and
the simulator's 512x512 frames (config/simulator/habitat.yaml).  Replica is not available offline, so the keyframes are
rendered from a dense ground-truth surfel room (``active_gs_amd.synthetic.make_keyframes``).  Prints one JSON line.
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--keyframes", type=int, default=50)
    ap.add_argument("--steps", type=int, default=10, help="optimisation iterations per keyframe")
    ap.add_argument("--size", type=int, nargs=2, default=[512, 512], metavar=("H", "W"))
    ap.add_argument("--gt-surfels", type=int, default=400_000)
    ap.add_argument("--no-warmup", action="store_true",
                    help="time the loop cold: the first process on a box then also pays torch's lazy kernel-module loads "
                         "(~0.25 s: topk, sort, index kernels) inside the timed region")
    ap.add_argument("--host-sampler", action="store_true",
                    help="draw the error-weighted frames with the reference's np.random.choice on the host (one read-back "
                         "of the per-frame errors per iteration) instead of the same distribution on the GPU")
    ap.add_argument("--split", action="store_true", help="synchronise around growth and training to time them separately")
    ap.add_argument("--repeat", type=int, default=1, help="run the loop this many times (one JSON line each)")
    ap.add_argument("--device-pose", action="store_true",
                    help="the dataframes carry the pose on the device only (the map reads it back per keyframe) instead of "
                         "also on the host (INTEGRATION.md section 3: host-pose form)")
    args = ap.parse_args()

    from active_gs_amd import env_config
    env_config.apply_env(os.environ)   # the package itself reads no environment variable
    from active_gs_amd.synthetic import make_keyframes, run_mapper_loop
    dev = torch.device("cuda:0")
    h, w = args.size
    frames = make_keyframes(args.keyframes, h, w, dev, gt_surfels=args.gt_surfels, host_pose=not args.device_pose)
    for rep in range(args.repeat):
        np.random.seed(0)
        print(json.dumps(run_mapper_loop(frames, steps=args.steps, draw="host" if args.host_sampler else "device",
                                         warmup_frames=0 if (args.no_warmup or rep) else 2, split=args.split)), flush=True)


if __name__ == "__main__":
    main()
