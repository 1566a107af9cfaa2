#!/usr/bin/env python3
"""How full the blend loops' waves are (probe build: render.hip with -DAGS_PROBE_LANES, loaded through AGS_LIB_PATH): of
the (surfel, 8x8-pixel wave) pairs that enter the loop body, how many take at least one pixel, how many of the 64 pixels
take the surfel, and how many of the wave's four 4x4 blocks / four 8x2 row pairs have a taken pixel - what a blend loop
over 16-pixel groups (four surfels at once per wave) could skip.  Configs: C2 (200 k random surfels @1200x680), the
mapper loop's training batches (config 3), C4 share, C5.  Prints one JSON line per config.
usage: AGS_LIB_PATH=scratch/libags_probe_lanes.so AGS_FREEZE=1 python profiles/experiments/lane_occupancy.py"""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)


def read(lib, reset=True):
    out = (C.c_ulonglong * 16)()
    torch.cuda.synchronize()
    assert lib.ags_probe_read(out, int(reset)) == 0
    v = np.array(list(out), dtype=np.float64).reshape(2, 8)
    res = {}
    for kid, name in ((0, "forward"), (1, "backward")):
        pairs, anyp, px, b44, b82, waves = v[kid][:6]
        if pairs == 0:
            continue
        res[name] = dict(pairs=int(pairs), waves=int(waves), pairs_with_a_pixel=round(anyp / pairs, 3),
                         pixels_per_blended_pair=round(px / max(anyp, 1), 2), lane_use_of_blended_pairs=round(px / max(anyp, 1) / 64, 3),
                         blocks4x4_per_blended_pair=round(b44 / max(anyp, 1), 3), rowpairs8x2_per_blended_pair=round(b82 / max(anyp, 1), 3))
    return res


def step_config(tag, n, h, w, views, room):
    from active_gs_amd import env_config
    env_config.apply_env(os.environ)   # the package itself reads no environment variable
    from active_gs_amd import raster_api as api
    from active_gs_amd.camera import camera_matrices
    from active_gs_amd.synthetic import make_camera, make_room_scene
    from active_gs_amd.trainer import SurfelTrainer
    dev = torch.device("cuda:0")
    raw = {k: v.to(dev) for k, v in make_room_scene(n, room=room, seed=0).items()}
    trainer = SurfelTrainer(raw, lrs=dict(mean=0.0, scale=0.0, rotation=0.0, opacity=0.0, harmonic=0.0), view_streams=1)
    cams = []
    for v in range(views):
        c2w, K = make_camera(v, h, w)
        cm = camera_matrices(c2w[None].to(dev), K[None].to(dev), 0.001, 10.0)
        tan = cm["tanfov"][0].cpu()
        cams.append(api.Camera(h, w, float(tan[0]), float(tan[1]), cm["viewmatrix"][0].contiguous(),
                               cm["projmatrix"][0].contiguous(), torch.zeros(4, device=dev)))
    gen = torch.Generator().manual_seed(4)
    d_img = [(torch.randn(c, h, w, generator=gen) / (h * w * views)).to(dev) for c in (3, 3, 1)]
    fn = lambda v, st: (d_img[0], d_img[1], d_img[2], None, None)
    cap = 1 << 22
    while True:
        trainer.step(cams, fn, cap, device_clock=True)
        need = 0
        for cam in cams:
            st = trainer.state_for(h, w, cap)
            api.forward(cam, trainer.gaussians(), st)
            need = max(need, api.read_status(st)["needed"])
        if need <= cap:
            break
        cap = int(need * 1.25)
    return trainer, cams, fn, cap


def main():
    from active_gs_amd import _lib
    lib = _lib.load()
    lib.ags_probe_read.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    lib.ags_probe_read.restype = C.c_int
    for tag, n, h, w, views, room in (("C2: 200 k surfels @1200x680", 200_000, 680, 1200, 1, "room0"),
                                      ("C4 share: 1.5 M surfels, 4 views @1200x680", 1_500_000, 680, 1200, 4, "room0"),
                                      ("C5: 5 M surfels @2048x2048", 5_000_000, 2048, 2048, 1, "office0")):
        trainer, cams, fn, cap = step_config(tag, n, h, w, views, room)
        read(lib)
        trainer.step(cams, fn, cap, device_clock=True)
        print(json.dumps(dict(config=tag, **read(lib))), flush=True)
        del trainer
        torch.cuda.empty_cache()
    from active_gs_amd.synthetic import make_keyframes, run_mapper_loop
    frames = make_keyframes(50, 512, 512, torch.device("cuda:0"))
    read(lib)
    np.random.seed(0)
    out = run_mapper_loop(frames, warmup_frames=0)
    print(json.dumps(dict(config="C3: mapper loop, 50 keyframes x 10 iterations x 11 views @512x512 (all its renders)", **read(lib))), flush=True)


if __name__ == "__main__":
    main()
