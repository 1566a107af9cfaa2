#!/usr/bin/env python3
"""The path an UNMODIFIED ActiveGS takes through this library: ``GaussianRenderer.render_view_all(require_grad=True)``
-> ``diff_gaussian_rasterization_2d.GaussianRasterizer`` (one autograd Function per view, operations.py:682-713,854) ->
``loss.backward()`` (gaussian_map.py:125), here through this repository's mirror of the facade
(``active_gs_amd.facade.SurfelRenderer``, same call sequence, same two ``.item()`` reads per view).

Timed: one training-iteration's worth of renders + backward, (a) BASELINE config C2 (200 k surfels, one 1200x680 view)
and (b) the reference's own shape (8 views @512x512, incremental.yaml:24-25, habitat.yaml:8-9), against the fused
trainer path (``SurfelTrainer`` / C ABI, no autograd, no per-call allocation) on the same views."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from active_gs_amd import env_config
env_config.apply_env(os.environ)   # the package itself reads no environment variable
from active_gs_amd import raster_api as api  # noqa: E402
from active_gs_amd.camera import camera_matrices  # noqa: E402
from active_gs_amd.facade import SurfelRenderer  # noqa: E402
from active_gs_amd.synthetic import activate, make_camera, make_room_scene  # noqa: E402
from active_gs_amd.trainer import SurfelTrainer  # noqa: E402


def run(tag, n, h, w, views, iters, focal=None):
    dev = torch.device("cuda:0")
    raw = {k: v.to(dev) for k, v in make_room_scene(n, seed=0).items()}
    c2w, K = zip(*[make_camera(v, h, w, focal_px=focal) for v in range(views)])
    extr, intr = torch.stack(c2w).to(dev), torch.stack(K).to(dev)
    params = {k: raw[k].clone().requires_grad_(True) for k in ("means", "scales", "rotations", "opacities", "harmonics")}
    conf = raw["confidences"]
    gen = torch.Generator().manual_seed(0)
    tgt = [torch.rand(views, c, h, w, generator=gen).to(dev) for c in (3, 1, 3)]

    def attr():   # get_attr(): activations under autograd, once per iteration (gaussian_map.py:573-581)
        return (params["means"], params["harmonics"], torch.sigmoid(params["opacities"]), conf,
                torch.clamp(0.01 * torch.exp(params["scales"]), 0, 0.05), torch.nn.functional.normalize(params["rotations"]))

    def iteration():
        r = SurfelRenderer(extr, intr, attr(), torch.zeros(4, device=dev), (0.001, 10.0), (h, w), dev)
        rgb, depth, normal, opacity, d2n, confidence, *_ = r.render_view_all(require_grad=True)
        loss = (rgb - tgt[0]).abs().mean() + 0.8 * (depth - tgt[1]).abs().mean() + 0.1 * (normal - tgt[2]).abs().mean()
        loss.backward()
        for p in params.values():
            p.grad = None

    for _ in range(3):
        iteration()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        iteration()
    torch.cuda.synchronize()
    t_mod = (time.perf_counter() - t0) / iters

    # the module alone: GaussianRasterizer.__call__ + backward of a fixed linear functional of its five images (no
    # facade post-processing, no loss head, no activations) - what the extension itself costs an unmodified caller
    from diff_gaussian_rasterization_2d import GaussianRasterizationSettings, GaussianRasterizer
    cm0 = camera_matrices(extr.cpu(), intr.cpu(), 0.001, 10.0)
    a = activate({k: v.detach() for k, v in raw.items()})
    leaves = [a["means"].clone().requires_grad_(True), torch.zeros(n, 3, device=dev, requires_grad=True),
              a["opacities"][:, None].clone().requires_grad_(True), a["confidences"], a["colors"].clone().requires_grad_(True),
              a["scales"].clone().requires_grad_(True), a["rotations"].clone().requires_grad_(True)]
    gimg = [torch.randn(c, h, w, generator=gen).to(dev) / (h * w) for c in (3, 3, 1, 1, 1)]
    settings = [GaussianRasterizationSettings(
        image_height=h, image_width=w, tanfovx=float(cm0["tanfov"][v, 0]), tanfovy=float(cm0["tanfov"][v, 1]),
        bg=torch.zeros(4, device=dev), scale_modifier=1.0, viewmatrix=cm0["viewmatrix"][v].to(dev),
        projmatrix=cm0["projmatrix"][v].to(dev), sh_degree=0, campos=cm0["campos"][v].to(dev), prefiltered=False,
        render_mask=torch.tensor([], device=dev), weight_thres=0.03, debug=False,
        config=torch.tensor([1.0, 1, 1, 0, 0]).to(dev)) for v in range(views)]

    def module_only():
        outs = [GaussianRasterizer(s)(leaves[0], leaves[1], leaves[2], leaves[3], None, leaves[4], leaves[5], leaves[6], None)
                for s in settings]
        torch.autograd.backward([o[k] for o in outs for k in range(5)], [gimg[k] for _ in outs for k in range(5)])
        for t in leaves:
            t.grad = None

    for _ in range(3):
        module_only()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        module_only()
    torch.cuda.synchronize()
    t_only = (time.perf_counter() - t0) / iters

    # the fused path on the same views: C ABI, fixed image gradients, no autograd
    tr = SurfelTrainer({k: v.clone() for k, v in raw.items()})
    cm = camera_matrices(extr.cpu(), intr.cpu(), 0.001, 10.0)
    cams = [api.Camera(h, w, float(cm["tanfov"][v, 0]), float(cm["tanfov"][v, 1]), cm["viewmatrix"][v].contiguous().to(dev),
                       cm["projmatrix"][v].contiguous().to(dev), torch.zeros(4, device=dev)) for v in range(views)]
    d_img = [(torch.randn(c, h, w, generator=gen) / (h * w * views)).to(dev) for c in (3, 3, 1)]
    fn = lambda v, st: (d_img[0], d_img[1], d_img[2], None, None)
    probe = api.alloc_state(n, h, w, 1 << 24, dev)
    need = 0
    for cam in cams:
        api.forward(cam, tr.gaussians(), probe)
        need = max(need, api.read_status(probe)["needed"])
    del probe
    cap = int(need * 1.3) + 4096
    for _ in range(3):
        tr.step(cams, fn, cap)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        tr.step(cams, fn, cap)
    torch.cuda.synchronize()
    t_fused = (time.perf_counter() - t0) / iters
    return dict(config=tag, surfels=n, image=[h, w], views=views, iterations=iters,
                dropin_facade_ms_per_iteration=round(t_mod * 1e3, 3), dropin_facade_ms_per_view=round(t_mod * 1e3 / views, 3),
                module_only_ms_per_iteration=round(t_only * 1e3, 3), module_only_ms_per_view=round(t_only * 1e3 / views, 3),
                fused_trainer_eager_ms_per_iteration=round(t_fused * 1e3, 3),
                note="dropin_facade: SurfelRenderer.render_view_all (the reference facade's call sequence: settings + module + "
                     "normalise + depth_to_normal per view) + a torch L1 loss + backward, no optimiser step; module_only: the "
                     "autograd Function alone; fused: SurfelTrainer.step incl. Adam, eager launches")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    out = [run("C2: 200k surfels, one 1200x680 view", 200_000, 680, 1200, 1, a.iters),
           run("reference shape: 200k surfels, 8 views @512x512 (60 deg)", 200_000, 512, 512, 8, a.iters, focal=0.5 * 512 / 0.57735)]
    for o in out:
        print(json.dumps(o))
