"""The reference's REAL entry points over the fused path: ``active_gs_amd.gaussian_map.GaussianMap(cfg, device)`` with the
surface an untouched ``mapping.Mapper`` / planner / voxel map / recorder uses (/root/reference/mapping/mapper.py:44,101,
mapping/gaussian_map.py:62-64,491-581, mapping/voxel_map.py:71-74, utils/common.py:249) and ``facade.SurfelRenderer`` serving
``render_view(i)`` under no_grad from ONE batched render (/root/reference/planning/confidence.py:24-46)."""
import os
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DEV = torch.device("cuda:0")


def _ns(d):
    """a nested dict as attribute-style config (what hydra hands to mapping.Mapper: cfg.gaussian_map)"""
    return NS(**{k: (_ns(v) if isinstance(v, dict) else v) for k, v in d.items()})


def _to_dev(frame):
    return {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in frame.items()}      # mapper.py:95


def test_class_api_replays_the_reference_mapper_loop_capture(agslib):
    """tests/golden/mapper_loop.pt = the reference's own ``GaussianMap.update()`` x 4 keyframes from an empty map (over the
    CPU oracle).  The same calls on the drop-in class - ``GaussianMap(cfg, device)``, ``.update(dataframe)`` - give the same
    growth after every keyframe, the same per-frame errors, supports and scores, the same prune decisions."""
    from active_gs_amd.gaussian_map import GaussianMap
    g = torch.load(os.path.join(GOLD, "mapper_loop.pt"))
    from _origin import RowOrigins
    from test_gpu_densify import CAPTURE_GATES, check_capture_final, check_capture_keyframe
    gm = GaussianMap(_ns(g["cfg"]), DEV)
    gm.frame_sampler = "host"                      # the capture drew its batches from a seeded numpy stream
    np.random.seed(g["seed"])
    assert not gm.is_init and gm.get_means.shape[0] == 0
    origins = RowOrigins(gm._fused())              # (the class's add_gaussians / prune are the trainer's)
    for k, ref in enumerate(g["history"]):
        assert abs(gm.get_means.shape[0] - ref["n_before"]) <= CAPTURE_GATES["rows"]
        gm.update(_to_dev(g["frames"][k % 2]))
        assert gm.is_init and len(gm.training_data) == k + 1
        n = gm.get_means.shape[0]
        check_capture_keyframe(k, ref, n, gm.training_performance.cpu(), float(gm.get_opacities.mean()),
                               float(gm.view_supports.mean()), float(gm.view_scores.mean()))
        # the surface the reference's other components read after every keyframe
        means, harmonics, opac, conf, scales, rot = gm.get_attr()                   # planners, eval, mesh
        assert means.shape == (n, 3) and harmonics.shape == (n, 1, 3) and opac.shape == conf.shape == (n,)
        assert scales.shape == (n, 3) and rot.shape == (n, 4)
    # the final parameters row by row (rows aligned by where they were spawned: tests/_origin.py) - always
    check_capture_final(g, origins, dict(means=gm._means, harmonics=gm._harmonics, scales=gm._scales, opacities=gm._opacities,
                                         rotations=gm._rotations))


def test_voxel_map_properties_recorder_save_and_planner_inputs(agslib, tmp_path):
    """What mapping/voxel_map.py:71-74 reads (``get_means / get_normals / get_confidences / get_opacities``), what the
    recorder calls (``save(path, index=)``, utils/common.py:249) and what the planners hand to ``GaussianRenderer``
    (``get_attr()``, ``background_color``, ``scene_near / scene_far``: planning/confidence.py:24-32), on a map grown by the
    class itself; ``load`` in a fresh map gives the same getters back (eval.py / mesh_generation.py / visualize.py)."""
    from active_gs_amd.gaussian_map import GaussianMap
    g = torch.load(os.path.join(GOLD, "mapper_loop.pt"))
    cfg = _ns(g["cfg"])
    gm = GaussianMap(cfg, DEV)
    for k in range(2):
        gm.update(_to_dev(g["frames"][k]))
    n = gm.get_means.shape[0]
    mean, normal = gm.get_means.detach(), gm.get_normals.detach()
    conf, opac = gm.get_confidences.detach(), gm.get_opacities.detach()
    assert mean.shape == normal.shape == (n, 3) and conf.shape == opac.shape == (n,)
    assert torch.allclose(normal.norm(dim=-1), torch.ones(n, device=DEV), atol=1e-5)
    # the same four through the reference's own expressions (gaussian_map.py:529-571)
    q = torch.nn.functional.normalize(gm._rotations)
    r, x, y, z = q.unbind(-1)
    assert torch.allclose(normal, torch.nn.functional.normalize(torch.stack([2 * (x * z + r * y), 2 * (y * z - r * x),
                                                                            1 - 2 * (x * x + y * y)], -1)), atol=1e-6)
    var = gm.view_means.norm(dim=-1)
    var = torch.where(torch.isnan(var), torch.ones_like(var), var)
    assert torch.allclose(conf, torch.clamp(torch.exp(1 - var) * gm.view_scores, min=0, max=1), atol=1e-6)
    assert torch.equal(opac, torch.sigmoid(gm._opacities)) and float(conf.min()) >= 0.0 and float(conf.max()) <= 1.0
    assert gm.background_color.shape == (4,) and (gm.scene_near, gm.scene_far) == (0.001, 10.0)
    gm.save(str(tmp_path), index="003")
    path = os.path.join(str(tmp_path), "map_003.th")
    st = torch.load(path)
    assert sorted(st) == sorted(["means", "scales", "harmonics", "opacities", "rotations", "view_scores", "view_supports",
                                 "view_means", "near", "far", "use_view_direction", "background_color", "scale_factor"])
    # the file holds the map's rows, not the larger buffers they are the leading rows of
    n = gm.get_means.shape[0]
    assert st["means"].untyped_storage().nbytes() == n * 3 * 4 and os.path.getsize(path) < 4 * 19 * n + (1 << 16)
    g2 = GaussianMap(None, DEV)
    g2.load(path)
    for k, (a, b) in enumerate(zip(gm.get_attr(), g2.get_attr())):
        # (confidences: one launch on a map that has a trainer, the reference's torch expression on a freshly loaded one)
        assert torch.allclose(a, b, atol=1e-6) if k == 3 else torch.equal(a, b), k
    assert torch.equal(g2.get_normals, gm.get_normals) and g2.is_init
    # a loaded map renders (mesh_generation.py:74-82) and trains on (cfg = None: the yaml's defaults)
    from active_gs_amd.facade import SurfelRenderer
    f = _to_dev(g["frames"][0])
    h, w = f["rgb"].shape[-2:]
    with torch.no_grad():
        rgb, depth, *_ = SurfelRenderer(f["extrinsic"][None], f["intrinsic"][None], g2.get_attr(), g2.background_color,
                                        (g2.scene_near, g2.scene_far), (h, w), DEV).render_view_all()
    valid = f["depth"] > 0
    assert float((rgb[0] - f["rgb"]).abs().mean()) < 0.1 and float((depth[0] - f["depth"])[valid].abs().mean()) < 0.1


def test_prune_add_and_train_as_separate_calls(agslib):
    """The reference's methods one by one (gaussian_map.py:66,141,234,294): add_gaussians registers the frame and grows the
    map, train(steps) runs that many iterations + post_processing, prune(mask) deletes the masked surfels and those whose
    opacity fell under 0.1, attribute changes between calls are honoured."""
    from active_gs_amd.gaussian_map import GaussianMap
    g = torch.load(os.path.join(GOLD, "mapper_loop.pt"))
    gm = GaussianMap(_ns(g["cfg"]), DEV)
    added = gm.add_gaussians(_to_dev(g["frames"][0]))
    n0 = gm.get_means.shape[0]
    assert added == n0 > 1000 and len(gm.training_data) == 1 and gm.training_performance.tolist() == [10.0] and not gm.is_init
    assert float(gm.view_supports.sum()) == 0.0
    gm.optimization_steps = 3
    gm.train()
    assert gm.is_init and len(gm._trainer.last_losses) == 3 and float(gm.training_performance[0]) < 10.0
    assert float(gm.view_supports.sum()) > 0.5 * n0          # post_processing counted the newest view
    gm.train(steps=2)
    assert len(gm._trainer.last_losses) == 2
    n1 = gm.get_means.shape[0]
    mask = torch.zeros(n1, device=DEV)
    mask[::3] = 1.0
    low = int((gm.get_opacities < 0.1).sum())
    expect = int((((mask > 0) | (gm.get_opacities < 0.1))).sum())
    gm.prune(mask)
    assert gm.get_means.shape[0] == n1 - expect and expect >= (n1 + 2) // 3 and low >= 0
    for t, wdt in ((gm._scales, 3), (gm._rotations, 4), (gm._harmonics, 3), (gm.view_means, 3)):
        assert t.numel() == gm.get_means.shape[0] * wdt
    gm.update(_to_dev(g["frames"][1]))                       # and the loop goes on
    assert len(gm.training_data) == 2 and gm.get_means.shape[0] > n1 - expect


def test_batched_render_view_serves_the_reference_renderer_capture(agslib):
    """tests/golden/facade.pt = the reference's own ``GaussianRenderer.render_view_all`` (over the CPU oracle).  On the
    GPU ``SurfelRenderer.render_view(i)`` under no_grad renders ALL the renderer's views as one batch at the first
    request and serves view i from it: same 9-tuples as the capture and - bitwise - as the view-by-view path."""
    from active_gs_amd import raster_api as api
    from active_gs_amd.facade import SurfelRenderer
    d = torch.load(os.path.join(GOLD, "facade.pt"))
    t = lambda x: x.to(DEV)
    attr = (t(d["attr"]["means"]), t(d["attr"]["harmonics"]), t(d["attr"]["opacities"]), t(d["confidences"]),
            t(d["attr"]["scales"]), t(d["attr"]["rotations"]))
    calls = dict(n=0)
    orig = api.ViewBatch.forward

    def counting(self, *a, **k):
        calls["n"] += 1
        return orig(self, *a, **k)
    api.ViewBatch.forward = counting
    try:
        r = SurfelRenderer(t(d["c2w"]), t(d["K"]), attr, t(d["bg"]), (d["near"], d["far"]), (d["h"], d["w"]), DEV)
        views = [r.render_view(i) for i in range(d["c2w"].shape[0])]
        assert calls["n"] == 1                                   # one batched render served both requests
        allv = r.render_view_all()
        assert calls["n"] == 1
    finally:
        api.ViewBatch.forward = orig
    for i, v in enumerate(views):
        assert len(v) == 9
        for k in range(8):
            ref = d["outputs"][k][i]
            assert v[k].shape == ref.shape and v[k].dtype == ref.dtype, (i, k)
            if ref.dtype.is_floating_point:
                assert torch.allclose(v[k].cpu(), ref, rtol=2e-3, atol=2e-4), (i, k, float((v[k].cpu() - ref).abs().max()))
            else:
                assert torch.equal(v[k].cpu(), ref), (i, k)
            assert torch.equal(allv[k][i], v[k])
        assert v[8].dtype == torch.bool
    assert torch.equal(allv[8].cpu(), d["outputs"][8])
    # the view-by-view path (what render_view(i, require_grad=True) takes) renders the same bits
    one = [SurfelRenderer(t(d["c2w"])[i:i + 1], t(d["K"])[i:i + 1], attr, t(d["bg"]), (d["near"], d["far"]), (d["h"], d["w"]),
                          DEV).render_view(0, require_grad=True) for i in range(len(views))]
    for i, v in enumerate(views):
        for k in range(9):
            assert torch.equal(one[i][k].detach(), v[k]), (i, k)
    # update_attr forgets the batch
    moved = (attr[0] + 0.01,) + attr[1:]
    r.update_attr(moved)
    assert not torch.equal(r.render_view(0)[0], views[0][0])


def test_planner_shaped_batch_of_candidate_views(agslib):
    """planning/confidence.py:24-46: one renderer for ~100 candidate poses at 128x128, ``render_view(i)`` per candidate under
    no_grad, ``confidence[0]`` and ``depth[0]`` consumed.  Importance / front_only / render-mask requests (post-processing,
    gaussian_map.py:183-192) batch the same way; every view equals the one-view-at-a-time render bitwise."""
    from active_gs_amd.facade import SurfelRenderer
    from active_gs_amd.synthetic import activate, make_camera, make_room_scene
    n, V, h, w = 40000, 24, 128, 128
    a = {k: v.to(DEV) for k, v in activate(make_room_scene(n, seed=2)).items()}
    attr = (a["means"], a["colors"][:, None, :].contiguous(), a["opacities"], a["confidences"], a["scales"] * 2.5, a["rotations"])
    cams = [make_camera(100 + v, h, w, focal_px=0.5 * w / np.tan(np.pi / 6)) for v in range(V)]
    c2w, K = torch.stack([c[0] for c in cams]).to(DEV), torch.stack([c[1] for c in cams]).to(DEV)
    bg = torch.zeros(4, device=DEV)
    gen = torch.Generator().manual_seed(0)
    masks = (torch.rand(V, 1, h, w, generator=gen) > 0.2).float().to(DEV)
    for kw, m in ((dict(), None), (dict(require_importance=True, front_only=True), masks)):
        r = SurfelRenderer(c2w, K, attr, bg, (0.001, 10.0), (h, w), DEV, render_masks=m)
        with torch.no_grad():
            got = [r.render_view(i, **kw) for i in range(V)]
        for i in (0, 7, V - 1):
            one = SurfelRenderer(c2w[i:i + 1], K[i:i + 1], attr, bg, (0.001, 10.0), (h, w), DEV,
                                 render_masks=None if m is None else m[i:i + 1]).render_view(0, **kw)
            for k in range(9):
                if k == 6:       # importance: float atomics over the waves that blended a surfel
                    assert float((one[k] - got[i][k]).abs().sum()) <= 1e-5 * float(one[k].abs().sum()) + 1e-12
                else:
                    assert torch.equal(one[k], got[i][k]), (i, k)
            conf, depth = got[i][5][0], got[i][1][0]
            assert float(conf.min()) >= 0.0 and float(conf.max()) <= 1.0 and depth.shape == (h, w)
        if kw:
            assert int(got[0][7].sum()) > 0 and got[0][7].dtype == torch.int32


def test_class_api_on_the_less_travelled_paths(agslib):
    """The same class where the fused batch does not apply or the configuration differs from the yaml's defaults:
    keyframes with DIFFERENT intrinsics (per-view launches, the torch-side post-processing), ``use_view_distribution = False``
    (confidence from the support count, gaussian_map.py:560-563), ``sampler_type: uniform`` (gaussian_map.py:253-254), the
    host-side frame draw, ``train(steps=0)`` (post-processing only)."""
    import torch.nn.functional as F
    from active_gs_amd.gaussian_map import GaussianMap
    g = torch.load(os.path.join(GOLD, "mapper_loop.pt"))
    base = dict(g["cfg"])
    # (a) keyframes with DIFFERENT intrinsics (one size): the batch path needs one field of view, so the views go through
    # per-view launches and the torch-side post-processing; a keyframe of another SIZE is refused like the reference's
    # torch.stack of the sampled frames refuses it (mapping/utils.py:220-221)
    gm = GaussianMap(_ns(base), DEV)
    gm.frame_sampler = "host"
    np.random.seed(1)
    for k in range(2):
        gm.update(_to_dev(g["frames"][k]))
    f = _to_dev(g["frames"][0])
    n0 = gm.get_means.shape[0]
    wide = dict(f, intrinsic=f["intrinsic"] * torch.tensor([[0.9, 1.0, 1.0], [1.0, 0.9, 1.0], [1.0, 1.0, 1.0]], device=DEV))
    gm.update(wide)
    assert len(gm.training_data) == 3 and gm.training_performance.numel() == 3 and not gm._trainer._uniform_frames()
    assert bool(torch.isfinite(gm._means).all()) and float(gm.training_performance[2]) < 10.0 and gm.get_means.shape[0] > 0.5 * n0
    assert len(gm._trainer.last_losses) == base["optimization_steps"] and all(np.isfinite(gm._trainer.last_losses))
    small = dict(f, rgb=F.interpolate(f["rgb"][None], size=(48, 80), mode="bilinear")[0].contiguous(),
                 depth=F.interpolate(f["depth"][None], size=(48, 80), mode="nearest")[0].contiguous())
    n1 = gm.get_means.shape[0]
    with pytest.raises(ValueError, match="different image sizes"):
        gm.update(small)
    assert len(gm.training_data) == 3 and gm.get_means.shape[0] == n1          # refused before anything was touched
    # (b) confidence from the support count
    cfg_b = dict(base, use_view_distribution=False)
    gb = GaussianMap(_ns(cfg_b), DEV)
    gb.update(_to_dev(g["frames"][0]))
    conf = gb.get_confidences
    assert torch.allclose(conf, torch.clamp(1 - torch.exp(-gb.view_supports), 0, 1), atol=1e-6)
    assert float(gb.view_scores.abs().sum()) == 0.0 and float(gb.view_supports.sum()) > 0
    # (c) the uniform sampler and a post-processing-only call
    cfg_c = dict(base, sampler=dict(base["sampler"], sampler_type="uniform"))
    gc = GaussianMap(_ns(cfg_c), DEV)
    torch.manual_seed(0)
    for k in range(4):
        gc.update(_to_dev(g["frames"][k % 2]))
    assert gc._trainer_cfg()["sampler_type"] == "uniform" and len(gc.training_data) == 4
    assert float(gc.training_performance.max()) < 10.0 and all(np.isfinite(gc._trainer.last_losses))
    before = [p.clone() for p in gc.get_params()]
    supports = gc.view_supports.clone()
    gc.train(steps=0)
    assert all(torch.equal(a, b) for a, b in zip(before, gc.get_params()))          # no iteration ran ...
    assert float((gc.view_supports - supports).sum()) > 0                             # ... post_processing did


def test_host_pose_form_of_update_equals_the_device_pose_form(agslib):
    """``update(dataframe)`` where the caller still has the pose on the host (CPU tensors under the usual keys, or
    ``*_host`` keys next to device copies - what /root/reference/mapping/mapper.py:94 holds before line 95 moves the dict
    to the device): the map reads nothing back for the camera, and the camera, the far bound and the map after two
    keyframes are what the device-pose form gives (same matrices bit for bit: the algebra runs on the host either way)."""
    from active_gs_amd.gaussian_map import GaussianMap
    g = torch.load(os.path.join(GOLD, "mapper_loop.pt"))
    maps = {}
    for form in ("device", "host_keys", "cpu_pose"):
        torch.manual_seed(0)
        torch.cuda.manual_seed(0)
        gm = GaussianMap(_ns(dict(g["cfg"])), DEV)
        for k in range(2):
            f = g["frames"][k]
            d = _to_dev(f)
            if form == "host_keys":
                d.update({key + "_host": f[key].clone() for key in ("extrinsic", "intrinsic", "depth_range")})
            elif form == "cpu_pose":
                d.update({key: f[key].clone() for key in ("extrinsic", "intrinsic", "depth_range")})
            gm.update(d)
        tr = gm._trainer
        assert all(("_pose_host" in fr) == (form != "device") for fr in tr.frames)
        assert all(not any(k.endswith("_host") and k != "_pose_host" and k != "_far_host" for k in fr) for fr in tr.frames)
        assert all(fr["extrinsic"].is_cuda and fr["intrinsic"].is_cuda for fr in tr.frames)
        cams = [tr._camera(i) for i in range(2)]
        maps[form] = dict(view=[c[0].viewmatrix.clone() for c in cams], proj=[c[0].projmatrix.clone() for c in cams],
                          fov=[(c[1], c[2]) for c in cams], far=[fr.get("_far_host") for fr in tr.frames],
                          n=gm.get_means.shape[0], perf=gm.training_performance.clone())
    for form in ("host_keys", "cpu_pose"):
        a, b = maps["device"], maps[form]
        assert a["fov"] == b["fov"] and a["far"] == b["far"] and a["n"] == b["n"]
        for x, y in zip(a["view"] + a["proj"], b["view"] + b["proj"]):
            assert torch.equal(x, y)
        assert torch.allclose(a["perf"], b["perf"], rtol=5e-3, atol=1e-5)
