#!/usr/bin/env python3
"""The mapper loop's phases from ONE run (synthetic.run_mapper_loop(phases=True): HIP event + host clock at every phase
boundary), with the deferred workspace check on and off.  Prints one JSON line per setting."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from active_gs_amd import env_config
env_config.apply_env(os.environ)   # the package itself reads no environment variable
from active_gs_amd.fused_map_trainer import FusedMapTrainer
from active_gs_amd.synthetic import make_keyframes, run_mapper_loop
dev = torch.device("cuda:0")
frames = make_keyframes(50, 512, 512, dev)
run_mapper_loop(frames[:6], steps=10, draw="device", warmup_frames=2)
for defer in (False, True, False, True):
    FusedMapTrainer.DEFER_SETTLE = defer
    np.random.seed(0)
    r = run_mapper_loop(frames, steps=10, draw="device", warmup_frames=0, phases=True)
    print(json.dumps(dict(defer_settle=defer, seconds=r["seconds"], gpu_bound_frac=r["gpu_bound_frac"], device_mallocs=r["device_mallocs"],
                          phases={k: (v["gpu_timeline_ms"], v["host_enqueue_ms"]) for k, v in r["phases"].items()})), flush=True)
