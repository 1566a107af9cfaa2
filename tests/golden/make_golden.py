#!/usr/bin/env python3
"""Generates tests/golden/*.pt.  Run ONLY in the build container (needs /root/reference).

Imports the reference's host-side Python (utils.operations, mapping.gaussian_map,
mapping.utils) with its absent third-party imports stubbed, plugs the CPU oracle in under
the module name ``diff_gaussian_rasterization_2d`` and records
  camera.pt     K, c2w -> fov / view / proj / campos from the reference's GaussianRenderer.__init__
  facade.pt     the 9-tuple of GaussianRenderer.render_view_all() + gradients of a fixed scalar
  train.pt      one GaussianMap.train() (3 iterations x 4 views + post_processing): inputs,
                raw parameters after Adam, training_performance, counts-derived state
  adam.pt       torch.optim.Adam (eps 1e-15, 5 groups) for 3 steps incl. zero-gradient rows
  oracle_*.pt   oracle forward/backward vectors on seeded scenes (detects oracle drift and is
                the fixed target of the GPU parity tests)
  densify.pt    GaussianMap.add_gaussians on a first keyframe (no map yet: every valid pixel is a
                candidate) and on a second one (candidates from the error mask of a render of the
                map), then GaussianMap.prune.  cv2 is absent, so cv2.bilateralFilter is served by
                oracle/densify_oracle.py's restatement; torch.randperm is the identity during
                voxel_downsample (see that file's header).
  mapper_loop.pt  GaussianMap.update() x 4 keyframes from an empty map (grow, train, prune): sizes
                and per-frame errors after every keyframe, final parameters.
Only tensors (inputs and expected outputs) are stored; no reference source text.
"""
import json
import os
import sys
import types
from unittest.mock import MagicMock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = "/root/reference"

from oracle.surfel_oracle import OracleSettings, rasterize  # noqa: E402
from _scenes import oracle_inputs, room_case  # noqa: E402


def install_reference():
    for name in ["jaxtyping", "cv2", "trimesh", "torchvision", "torchvision.transforms", "open3d", "torchmetrics",
                 "torchmetrics.image", "torchmetrics.image.lpip", "imgviz", "PIL", "PIL.Image"]:
        sys.modules[name] = MagicMock()
    mod = types.ModuleType("diff_gaussian_rasterization_2d")

    class GaussianRasterizationSettings:
        def __init__(self, **kw):
            self.__dict__.update(kw)

    class GaussianRasterizer:
        calls = []

        def __init__(self, raster_settings):
            self.s = raster_settings

        def __call__(self, means3D, means2D, opacities, confidences, shs, colors_precomp, scales, rotations,
                     cov3D_precomp):
            s = self.s
            S = OracleSettings(s.image_height, s.image_width, s.tanfovx, s.tanfovy, s.bg, s.scale_modifier,
                               s.viewmatrix, s.projmatrix, s.sh_degree, s.campos, s.prefiltered, s.render_mask,
                               s.weight_thres, s.debug, s.config)
            GaussianRasterizer.calls.append(dict(config=s.config.tolist(), mask=tuple(s.render_mask.shape)))
            return rasterize(means3D, means2D, opacities, confidences, colors_precomp, scales, rotations, S)

    mod.GaussianRasterizationSettings = GaussianRasterizationSettings
    mod.GaussianRasterizer = GaussianRasterizer
    sys.modules["diff_gaussian_rasterization_2d"] = mod
    sys.path.insert(0, REF)
    import utils.operations as ops
    import mapping.gaussian_map as gm
    torch.cuda.empty_cache = lambda: None
    return ops, gm, GaussianRasterizer


class AttrDict(dict):
    __getattr__ = dict.__getitem__


def mapper_cfg(steps):
    # values of /root/reference/config/mapper/incremental.yaml:12-32
    return AttrDict(use_view_distribution=True, bound=[0.001, 10.0], sparse_ratio=1.0, scale_factor=0.01,
                    error_thres=0.25, prune_interval=5, optimization_steps=steps, background=[0.0, 0.0, 0.0, 0.0],
                    sampler=AttrDict(sampler_type="weighted", batch_size=8, active_size=3),
                    optimizer=AttrDict(mean_lr=5e-4, scale_lr=1e-2, rotation_lr=5e-4, opacity_lr=1e-2,
                                       harmonic_lr=1e-4))


def main():
    # AGS_GOLDEN_ONLY="mapper_loop.pt,..." : run everything (the sections feed each other and share RNG streams) but
    # write only the named files - the other fixtures stay byte-identical
    only = [x for x in os.environ.get("AGS_GOLDEN_ONLY", "").split(",") if x]
    if only:
        real_save = torch.save
        torch.save = lambda obj, path, *a, **kw: real_save(obj, path, *a, **kw) if os.path.basename(path) in only else None
    ops, gm, Rast = install_reference()
    from active_gs_amd.synthetic import activate, make_camera, make_room_scene
    torch.manual_seed(0)
    np.random.seed(0)

    # ---------------------------------------------------------------- camera.pt
    cams = []
    for (h, w, f) in [(512, 512, 0.5 * 512 / np.tan(np.pi / 6)), (680, 1200, 600.0), (64, 64, 55.4)]:
        c2w, K = zip(*[make_camera(v, h, w, focal_px=float(f)) for v in range(3)])
        c2w, K = torch.stack(c2w), torch.stack(K)
        attr = tuple(torch.zeros(1, 3) for _ in range(6))
        r = ops.GaussianRenderer(c2w, K, attr, torch.zeros(4), (0.001, 10.0), (h, w), "cpu")
        cams.append(dict(h=h, w=w, c2w=c2w, K=K, near=0.001, far=10.0, fovs=r.fovs.clone(),
                         view_matrices=r.view_matrices.clone(), projection_matrices=r.projection_matrices.clone(),
                         cam_pos=r.cam_pos.clone()))
    torch.save(cams, os.path.join(HERE, "camera.pt"))

    # ---------------------------------------------------------------- facade.pt
    n, h, w = 400, 48, 64
    raw = make_room_scene(n, seed=11)
    raw["scales"][:, :2] += 1.2
    a = activate(raw)
    c2w, K = zip(*[make_camera(v, h, w) for v in range(2)])
    c2w, K = torch.stack(c2w), torch.stack(K)
    leaves = dict(means=a["means"].clone().requires_grad_(True), harmonics=raw["harmonics"].clone().requires_grad_(True),
                  opacities=a["opacities"].clone().requires_grad_(True), scales=a["scales"].clone().requires_grad_(True),
                  rotations=a["rotations"].clone().requires_grad_(True))
    attr = (leaves["means"], leaves["harmonics"], leaves["opacities"], a["confidences"], leaves["scales"],
            leaves["rotations"])
    bgc = torch.tensor([0.05, 0.1, 0.15, 0.0])
    r = ops.GaussianRenderer(c2w, K, attr, bgc, (0.001, 10.0), (h, w), "cpu")
    outs = r.render_view_all(require_grad=True)
    gen = torch.Generator().manual_seed(5)
    wts = [torch.randn(o.shape, generator=gen) for o in outs[:6]]
    scalar = sum((o * g).sum() for o, g in zip(outs[:6], wts))
    scalar.backward()
    torch.save(dict(h=h, w=w, c2w=c2w, K=K, bg=bgc, near=0.001, far=10.0,
                    attr={k: v.detach().clone() for k, v in leaves.items()}, confidences=a["confidences"],
                    outputs=[o.detach().clone() for o in outs], weights=wts, scalar=scalar.detach(),
                    grads={k: v.grad.clone() for k, v in leaves.items()}), os.path.join(HERE, "facade.pt"))

    # ---------------------------------------------------------------- train.pt
    n, h, w, nframes, steps = 1500, 64, 64, 4, 3
    raw = make_room_scene(n, seed=21)
    raw["scales"][:, :2] += 1.0
    gt_raw = {k: v.clone() for k, v in raw.items()}
    gjit = torch.Generator().manual_seed(2)
    gt_raw["means"] = gt_raw["means"] + 0.01 * torch.randn(n, 3, generator=gjit)
    gt_raw["harmonics"] = (gt_raw["harmonics"] + 0.1 * torch.randn(n, 1, 3, generator=gjit)).clamp(0, 1)
    ga = activate(gt_raw)
    frames = []
    for v in range(nframes):
        c2w, K = make_camera(v, h, w)
        with torch.no_grad():
            rr = ops.GaussianRenderer(c2w[None], K[None], (ga["means"], gt_raw["harmonics"], ga["opacities"],
                                                           ga["confidences"], ga["scales"], ga["rotations"]),
                                      torch.zeros(4), (0.001, 10.0), (h, w), "cpu").render_view_all()
        frames.append(dict(rgb=rr[0][0].clone(), depth=rr[1][0].clone(), extrinsic=c2w, intrinsic=K,
                           depth_range=torch.tensor([0.001, 10.0])))
    Rast.calls.clear()
    m = gm.GaussianMap(mapper_cfg(steps), "cpu")
    m._means, m._scales, m._rotations = raw["means"].clone(), raw["scales"].clone(), raw["rotations"].clone()
    m._opacities, m._harmonics = raw["opacities"].clone(), raw["harmonics"].clone()
    m.view_scores, m.view_supports, m.view_means = torch.zeros(n), torch.zeros(n), torch.zeros(n, 3)
    for f in frames:
        m.training_data.append(f)
        m.training_performance = torch.cat((m.training_performance, torch.tensor([10.0])), 0)
    np.random.seed(7)
    m.train()
    torch.save(dict(n=n, h=h, w=w, steps=steps, cfg=json.loads(json.dumps(mapper_cfg(steps))), raw_init=raw, frames=frames,
                    raw_final=dict(means=m._means.detach().clone(), scales=m._scales.detach().clone(),
                                   rotations=m._rotations.detach().clone(), opacities=m._opacities.detach().clone(),
                                   harmonics=m._harmonics.detach().clone()),
                    training_performance=m.training_performance.clone(), view_supports=m.view_supports.clone(),
                    view_scores=m.view_scores.clone(), view_means=m.view_means.clone(),
                    rasterizer_calls=list(Rast.calls)), os.path.join(HERE, "train.pt"))

    # ---------------------------------------------------------------- densify.pt
    from oracle import densify_oracle as dor
    ops.cv2.bilateralFilter = lambda img, d, sc, ss: dor.bilateral_filter(img, d, sc, ss)
    h, w = 64, 96
    gen = torch.Generator().manual_seed(31)
    dframes = []
    for v in range(2):
        c2w, K = make_camera(v + 5, h, w)
        with torch.no_grad():
            rr = ops.GaussianRenderer(c2w[None], K[None], (ga["means"], gt_raw["harmonics"], ga["opacities"],
                                                           ga["confidences"], ga["scales"], ga["rotations"]),
                                      torch.zeros(4), (0.001, 10.0), (h, w), "cpu").render_view_all()
        depth = rr[1][0].clone()
        depth[0, 5:12, 7:20] = 0.0       # holes: not valid, but not "invalid" for the smoothing either
        depth[0, 40:47, 60:70] = -1.0    # sensor-invalid
        depth += 0.002 * torch.randn(depth.shape, generator=gen) * (depth > 0)
        dframes.append(dict(rgb=rr[0][0].clone(), depth=depth, extrinsic=c2w, intrinsic=K,
                            depth_range=torch.tensor([0.001, 10.0])))
    m = gm.GaussianMap(mapper_cfg(2), "cpu")
    captured = {}
    orig_cal_mask = m.cal_mask

    def cal_mask_rec(rgb_gt, depth_gt, pred):
        captured["pred"] = None if pred is None else {k: v.detach().clone() for k, v in pred.items()}
        return orig_cal_mask(rgb_gt, depth_gt, pred)
    m.cal_mask = cal_mask_rec
    real_randperm = torch.randperm
    snaps = []

    def snap():
        return dict(means=m._means.detach().clone(), scales=m._scales.detach().clone(),
                    rotations=m._rotations.detach().clone(), opacities=m._opacities.detach().clone(),
                    harmonics=m._harmonics.detach().clone(), view_scores=m.view_scores.clone(),
                    view_supports=m.view_supports.clone(), view_means=m.view_means.clone())
    torch.randperm = lambda n, device=None: torch.arange(n)
    try:
        m.add_gaussians(dframes[0])
        snaps.append(dict(state=snap(), pred=captured["pred"]))
        # make the map differ from the second frame so that the error mask is non-trivial
        with torch.no_grad():
            m._opacities += 2.0 * torch.randn(m._opacities.shape, generator=gen)
            m._harmonics += 0.4 * torch.randn(m._harmonics.shape, generator=gen)
            m._scales[:, :2] += 0.5
        m.is_init = True
        before = snap()
        m.add_gaussians(dframes[1])
        snaps.append(dict(before=before, state=snap(), pred=captured["pred"]))
    finally:
        torch.randperm = real_randperm
    pm = torch.zeros(m._means.shape[0])
    pm[::7] = 1.0
    before_prune = snap()
    m.prune(pm.clone())
    torch.save(dict(h=h, w=w, frames=dframes, error_thres=0.25, first=snaps[0], second=snaps[1],
                    prune_mask=pm, before_prune=before_prune, after_prune=snap()), os.path.join(HERE, "densify.pt"))

    # ---------------------------------------------------------------- mapper_loop.pt
    # GaussianMap.update() (= add_gaussians + train + post_processing) for four keyframes from an
    # empty map, prune every 2nd keyframe: the "full mapper loop" configuration in miniature.
    loop_cfg = mapper_cfg(4)
    loop_cfg["prune_interval"] = 2
    loop_cfg["sampler"] = AttrDict(sampler_type="weighted", batch_size=4, active_size=2)
    m = gm.GaussianMap(loop_cfg, "cpu")
    torch.randperm = lambda n, device=None: torch.arange(n)
    np.random.seed(11)
    history = []
    # Where every row of the map comes from: growth appends rows and prune is a stable compaction, so an ORIGIN id
    # (keyframe << 32 | index among the rows that keyframe added) can ride along from outside - recorded by wrapping the
    # instance's add_gaussians / prune (the latter ORs the opacity rule into the mask it is handed, in place), so that a
    # test can compare the final parameters row by row even when a threshold pixel made the two maps differ in a row.
    origin = torch.zeros(0, dtype=torch.int64)
    track = dict(k=0, added=None, pruned=None)
    ref_add, ref_prune = m.add_gaussians, m.prune

    def add_rec(frame):
        nonlocal origin
        n0 = m._means.shape[0]
        out = ref_add(frame)
        n1 = m._means.shape[0]
        track["added"] = m._means.detach()[n0:n1].clone()           # the new rows as spawned (before any training)
        origin = torch.cat([origin, (track["k"] << 32) + torch.arange(n1 - n0, dtype=torch.int64)])
        return out

    def prune_rec(mask):
        nonlocal origin
        out = ref_prune(mask)
        track["pruned"] = mask.bool().clone()
        origin = origin[~mask.bool()]
        return out
    m.add_gaussians, m.prune = add_rec, prune_rec
    try:
        for k in range(4):
            n_before = m._means.shape[0]
            track.update(k=k, added=None, pruned=None)
            m.update(dict(dframes[k % 2]))
            assert origin.shape[0] == m._means.shape[0]
            history.append(dict(n_before=n_before, n_after=m._means.shape[0],
                                training_performance=m.training_performance.clone(),
                                opacity_mean=float(torch.sigmoid(m._opacities.detach()).mean()),
                                supports=m.view_supports.clone(), scores_mean=float(m.view_scores.mean()),
                                added_means=track["added"], pruned=track["pruned"]))
    finally:
        torch.randperm = real_randperm
    torch.save(dict(frames=dframes, cfg=json.loads(json.dumps(loop_cfg)), seed=11, history=history,
                    final=dict(means=m._means.detach().clone(), opacities=m._opacities.detach().clone(),
                               harmonics=m._harmonics.detach().clone(), scales=m._scales.detach().clone(),
                               rotations=m._rotations.detach().clone(), origin=origin.clone())),
               os.path.join(HERE, "mapper_loop.pt"))

    # ---------------------------------------------------------------- adam.pt
    gen = torch.Generator().manual_seed(9)
    shapes = [(300, 3), (300, 3), (300, 4), (300,), (300, 1, 3)]
    lrs = [5e-4, 1e-2, 5e-4, 1e-2, 1e-4]
    p0 = [torch.randn(*s, generator=gen) for s in shapes]
    params = [torch.nn.Parameter(p.clone()) for p in p0]
    opt = torch.optim.Adam([{"params": [p], "lr": lr} for p, lr in zip(params, lrs)], eps=1e-15)
    grads_all = []
    for _ in range(3):
        grads = [torch.randn(*s, generator=gen) for s in shapes]
        grads[0][::2] = 0
        grads_all.append(grads)
        for p, g in zip(params, grads):
            p.grad = g.clone()
        opt.step()
    torch.save(dict(lrs=lrs, eps=1e-15, p0=p0, grads=grads_all, p3=[p.detach().clone() for p in params]),
               os.path.join(HERE, "adam.pt"))

    # ---------------------------------------------------------------- oracle_*.pt
    for tag, (n, h, w, view, mult, cfg) in dict(small=(64, 64, 64, 0, 6.0, (1, 1, 1, 0, 0)),
                                                c1=(5000, 170, 300, 1, 2.0, (1, 1, 1, 1, 0))).items():
        a, S = room_case(n, h, w, view=view, seed=view + 40, scale_mult=mult, config=cfg)
        ins = oracle_inputs(a)
        outs = rasterize(*ins, S)
        gen = torch.Generator().manual_seed(3)
        gr = [torch.randn(o.shape, generator=gen) for o in outs[:5]]
        sum((o * g).sum() for o, g in zip(outs[:5], gr)).backward()
        torch.save(dict(n=n, h=h, w=w, view=view, mult=mult, config=cfg, seed=view + 40,
                        inputs=[t.detach().clone() for t in ins],
                        outputs=[o.detach().clone() for o in outs], image_grads=gr,
                        grads={i: ins[i].grad.clone() for i in (0, 1, 2, 4, 5, 6)}), os.path.join(HERE, f"oracle_{tag}.pt"))
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
