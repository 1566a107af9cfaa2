#!/usr/bin/env python3
"""One GPU's share of BASELINE.json's configurations 4 and 5 (the 8-GPU runs are the driver's):
  c4: room0 stand-in, 1.5 M surfels, 4 views per GPU @1200x680 (32 views over 8 GPUs)
  c5: 5 M surfels, 2048x2048, all channels, 1 view
Same step as bench.py (activations + forward + backward + Adam, hipGraph replay); prints one JSON
line per configuration with the step time, the instance counts and the workspace size."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(tag, n, h, w, views, room, steps, warmup):
    from active_gs_amd import env_config
    env_config.apply_env(os.environ)   # the package itself reads no environment variable
    from active_gs_amd import raster_api as api
    from active_gs_amd.camera import camera_matrices
    from active_gs_amd.synthetic import make_camera, make_room_scene
    from active_gs_amd.trainer import SurfelTrainer
    dev = torch.device("cuda:0")
    raw = {k: v.to(dev) for k, v in make_room_scene(n, room=room, seed=0).items()}
    # AGS_FREEZE=1 (kernel A/B experiments): learning rates of zero - the scene stays what it is whatever a variant's
    # gradients are, so per-kernel times of different builds can be compared
    lrs = dict(mean=0.0, scale=0.0, rotation=0.0, opacity=0.0, harmonic=0.0) if os.environ.get("AGS_FREEZE") == "1" else None
    trainer = SurfelTrainer(raw, lrs=lrs, view_streams=int(os.environ.get("AGS_VIEW_STREAMS", "4")))
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
    while True:                                     # size the workspace: grow until no view overflows
        trainer.step(cams, fn, cap, device_clock=True)
        need = 0
        for cam in cams:
            st = trainer.state_for(h, w, cap)
            api.forward(cam, trainer.gaussians(), st)
            need = max(need, api.read_status(st)["needed"])
        if need <= cap:
            break
        cap = int(need * 1.25)
    replay = trainer.capture(cams, fn, cap)
    for _ in range(warmup):
        replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    st = trainer.state_for(h, w, cap)
    info = api.read_status(st)
    print(json.dumps(dict(config=tag, surfels=n, image=[h, w], views_per_step=views, ms_per_step=round(dt * 1e3, 3),
                          gaussians_per_s=round(n * views / dt / 1e6, 1), unit="M Gaussians/s",
                          instances_last_view=info["num_instances"], visible_last_view=info["num_visible"],
                          member_rows=None if trainer.rows is None else int(trainer.rows.count.item()),
                          workspace_MB=round(st.workspace.numel() / 2**20, 1))), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", choices=["c4", "c5"])
    ap.add_argument("--steps", type=int, default=30)
    args = ap.parse_args()
    if args.only in (None, "c4"):
        run("c4 (one GPU's 4 of 32 views)", 1_500_000, 680, 1200, 4, "room0", args.steps, 5)
    if args.only in (None, "c5"):
        run("c5 (2048x2048, 1 view)", 5_000_000, 2048, 2048, 1, "office0", args.steps, 5)


if __name__ == "__main__":
    main()
