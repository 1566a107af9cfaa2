"""HIP path against the oracle AT the BASELINE sizes (not only through size-independent properties):
C2 in full (200 k surfels @1200x680, every tile), the per-GPU share of C4 (1.5 M surfels, one 1200x680 view)
and C5 (5 M surfels @2048x2048: two quadrants per wave, separate tile-count scan, over-full tile lists) on a
spread subset of tiles, and the 1024x1024 forward of the mesh-extraction render
(/root/reference/mesh_generation.py:19,74-82).  Tolerances are BASELINE.json's: 1e-4 mean L1 on the images
(depth, in metres: 1e-3), 1e-3 relative L1 on every gradient - incl. means2D (operations.py:703-713 outputs,
gaussian_map.py:125 gradients).  All calls go through the drop-in module, i.e. the C ABI."""
import pytest
import torch

import _parity
from _scenes import oracle_inputs, oracle_on_tiles, product_settings, room_case

pytestmark = pytest.mark.gpu
NAMES = ("rgb", "normal", "depth", "opacity", "confidence")


def _compare(n, h, w, room_seed, view, max_tiles, fullest, grads=True, scale_mult=1.0, config=(1, 1, 1, 0, 0), focal=None,
             last_contributor=False):
    a, S = room_case(n, h, w, view=view, seed=room_seed, scale_mult=scale_mult, config=config, focal_px=focal)
    return _compare_case(a, S, f"{n} surfels {w}x{h} view {view} x{scale_mult}", max_tiles, fullest, grads, last_contributor)


def _compare_case(a, S, label, max_tiles, fullest, grads=True, last_contributor=False):
    """``a``: activated surfels (means, scales, rotations, opacities, colors, confidences) on the CPU; ``S``: the view."""
    from diff_gaussian_rasterization_2d import GaussianRasterizer, check_overflow
    dev = torch.device("cuda:0")
    torch.set_num_threads(min(16, torch.get_num_threads()))
    n, h, w = a["means"].shape[0], S.image_height, S.image_width
    ins = oracle_inputs(a, requires_grad=grads)
    gen = torch.Generator().manual_seed(11)
    d_img = [torch.randn(c, h, w, generator=gen) / (h * w) for c in (3, 3, 1, 1, 1)]
    if grads:
        ref, covered, aux = oracle_on_tiles(ins, S, d_img, max_tiles=max_tiles, fullest=fullest)
    else:
        with torch.no_grad():
            ref, covered, aux = oracle_on_tiles(ins, S, [None] * 5, max_tiles=max_tiles, fullest=fullest)
    gin = [t.detach().clone().to(dev) for t in ins]
    if grads:
        for i in (0, 1, 2, 4, 5, 6):
            gin[i].requires_grad_(True)
    out = GaussianRasterizer(product_settings(S, dev))(gin[0], gin[1], gin[2], gin[3], None, gin[4], gin[5], gin[6], None)
    check_overflow()
    m = covered.to(dev)
    what = f"{label}, {len(aux['tiles'])} of {aux['nonempty']} tiles"
    # contract mean-L1, largest pixel error, worst-tile mean L1, regression gate - over the compared tiles
    err = _parity.check_images(ref, {k: o.detach().cpu() for k, o in zip(NAMES, out[:5])}, covered, what=what)
    # integer output: exact but for rows on a rounding boundary, every one of them counted and explained
    err["radii"] = _parity.radii_report(out[7], aux["G"], ins, S, what)
    if grads:
        sum((o * (g.to(dev) * m)).sum() for o, g in zip(out[:5], d_img)).backward()
        gn = {0: "means3D", 1: "means2D", 2: "opacities", 4: "colors", 5: "scales", 6: "rotations"}
        err["grads"] = _parity.check_grads({v: ins[i].grad for i, v in gn.items()}, {v: gin[i].grad for i, v in gn.items()}, what=what)
    if last_contributor:
        # which surfel every pixel blended LAST (the forward's n_contrib, turned into ids): the binning order, the
        # alpha cut and the transmittance stop all have to agree for this to match - through the C ABI, both binning
        # modes whose lists differ in layout
        from active_gs_amd import raster_api as api
        ref_last = _parity.oracle_last_contributor(aux, aux["n_contrib"], h, w)
        cam = api.Camera(h, w, S.tanfovx, S.tanfovy, S.viewmatrix.to(dev), S.projmatrix.to(dev), S.bg.to(dev))
        g = api.Gaussians(gin[0].detach(), gin[5].detach(), gin[6].detach(), gin[2].detach().reshape(-1).contiguous(),
                          gin[4].detach(), gin[3].detach())
        for mode in (api.BIN_DIRECT, api.BIN_TILE_SORT):
            st = api.alloc_state(n, h, w, 1 << 22, dev, mode)
            api.forward(cam, g, st)
            info = api.read_status(st)
            assert not info["overflow"], info
            err[f"last_{mode}"] = _parity.last_contributor_report(api.last_contributor(st, n, h, w), ref_last, covered, f"{what} mode {mode}")
            # final_T is what the opacity image is made of
            fT = api.workspace_region(st, n, h, w, api.REGION_FINAL_T, torch.float32).view(h, w)
            assert torch.equal(1.0 - fT, st.opacity[0])
    return err, aux


def test_c2_full_size_matches_oracle(agslib):
    """BASELINE config C2 in full: every non-empty tile of the 200 k-surfel 1200x680 view, 5 images, 6 gradients."""
    err, aux = _compare(200_000, 680, 1200, room_seed=0, view=0, max_tiles=None, fullest=0, last_contributor=True)
    assert len(aux["tiles"]) == aux["nonempty"] > 2000 and aux["instances"] > 50_000


def test_c4_share_view_matches_oracle_on_tile_subset(agslib):
    """The per-GPU share of C4: 1.5 M surfels (scales x1.5: tile lists of up to ~1 700 surfels), one 1200x680 view;
    oracle on ~160 spread tiles plus the 12 fullest."""
    err, aux = _compare(1_500_000, 680, 1200, room_seed=0, view=1, max_tiles=160, fullest=12, scale_mult=1.5)
    assert aux["max_list"] > 1000 and aux["instances"] > 2_000_000


def test_c5_2048_two_quadrants_per_wave_matches_oracle_on_tile_subset(agslib):
    """C5: 5 M surfels @2048x2048 = 16 384 tiles - the forward runs two quadrants per wave (SLOTS = 2) and the
    separate tile-count scan (ags_k_scan_tiles); oracle on ~100 spread tiles plus the 10 fullest."""
    err, aux = _compare(5_000_000, 2048, 2048, room_seed=0, view=0, max_tiles=100, fullest=10)
    assert aux["instances"] > 2_000_000


def test_overfull_tiles_at_2048_match_oracle(agslib):
    """2048x2048 with surfels large enough that tile lists exceed the 2048-key LDS sort (chunked sort + global
    merge passes) while SLOTS = 2 / scan_tiles are active."""
    err, aux = _compare(100_000, 2048, 2048, room_seed=2, view=2, max_tiles=40, fullest=16, scale_mult=10.0)
    assert aux["max_list"] > 3000, aux["max_list"]     # rect lists; the HIP lists (exact reach test) stay above 2048


def test_mesh_render_1024_forward_matches_oracle(agslib):
    """mesh_generation.py:19,74-82 renders every keyframe at 1024x1024, forward only (rgb + depth feed the TSDF)."""
    err, aux = _compare(200_000, 1024, 1024, room_seed=0, view=3, max_tiles=300, fullest=10, grads=False, focal=512.0)
    assert aux["nonempty"] > 1000


def test_reference_checkpoint_renders_at_mesh_resolution(agslib):
    """f4 end to end: the map the REFERENCE's GaussianMap.save wrote (tests/golden/map_ref.th) is loaded with map_io,
    activated like GaussianMap.get_attr and rendered forward-only at 1024x1024 through the facade mirror
    (mesh_generation.py:19,74-82: GaussianRenderer(...).render_view(i) per keyframe camera) - against the oracle."""
    import os
    from active_gs_amd import map_io
    from active_gs_amd.facade import SurfelRenderer
    from active_gs_amd.synthetic import activate, make_camera
    from oracle.surfel_oracle import OracleSettings, rasterize
    from active_gs_amd.camera import camera_matrices
    dev = torch.device("cuda:0")
    raw, cfg = map_io.load_map(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "map_ref.th"))
    raw["scales"] = raw["scales"].clone()
    raw["scales"][:, :2] += 2.0                                   # visible at 1024x1024 from inside the room
    raw["confidences"] = torch.ones(raw["means"].shape[0])
    a = activate(raw)
    h = w = 1024
    c2w, K = make_camera(3, h, w, focal_px=512.0)
    bg = torch.tensor(cfg["background"])
    attr = tuple(t.to(dev) for t in (a["means"], raw["harmonics"], a["opacities"], a["confidences"], a["scales"], a["rotations"]))
    r = SurfelRenderer(c2w[None].to(dev), K[None].to(dev), attr, bg.to(dev), cfg["bound"], (h, w), dev)
    rgb, depth, normal, opacity, d2n, confidence, importance, count, in_view = r.render_view(0)
    cm = camera_matrices(c2w[None], K[None], *cfg["bound"])
    S = OracleSettings(h, w, cm["tanfov"][0, 0].item(), cm["tanfov"][0, 1].item(), bg, 1.0, cm["viewmatrix"][0], cm["projmatrix"][0])
    with torch.no_grad():
        ref = rasterize(a["means"], torch.zeros_like(a["means"]), a["opacities"][:, None], a["confidences"], a["colors"],
                        a["scales"], a["rotations"], S)
    assert float(ref[3].max()) > 0.3                               # the view shows something
    assert float((rgb.cpu() - ref[0]).abs().mean()) < 1e-4 and float((opacity.cpu() - ref[3]).abs().mean()) < 1e-4
    assert float((depth.cpu() - ref[2]).abs().mean()) < 1e-3
    assert torch.equal(in_view.cpu(), ref[7] > 0)


def test_c3_size_mapper_grown_map_matches_oracle_on_tile_subset(agslib):
    """Configuration 3's own workload against the ORACLE: a map GROWN by the mapper loop (``GaussianMap.update`` over 40
    keyframes @512x512: ~200 k surfels, spatially coherent rows, the scales / opacities / rotations training left, the
    confidences post_processing left - not the seeded room of the other cases), one of its keyframe views at the
    reference's 512x512 (config/simulator/habitat.yaml:8-9): 5 images, 6 gradients, radii, on ~160 spread tiles plus the
    12 fullest - the rasterizer calls one fused iteration makes for a view, held to the same gates as C2 / C4 / C5."""
    import numpy as np
    from active_gs_amd.camera import camera_matrices
    from active_gs_amd.gaussian_map import GaussianMap
    from active_gs_amd.synthetic import make_keyframes, mapper_cfg
    from oracle.surfel_oracle import OracleSettings
    dev = torch.device("cuda:0")
    frames = make_keyframes(40, 512, 512, dev, gt_surfels=400_000)
    np.random.seed(0)
    gm = GaussianMap(mapper_cfg(10, "device"), dev)
    for f in frames:
        gm.update(f)
    n = gm.get_means.shape[0]
    assert n > 150_000
    means, harmonics, opac, conf, scales, rot = (t.detach().float().cpu().contiguous() for t in gm.get_attr())
    a = dict(means=means, scales=scales, rotations=rot, opacities=opac, colors=harmonics[:, 0, :].contiguous(), confidences=conf)
    f = frames[17]
    cm = camera_matrices(f["extrinsic"][None].cpu(), f["intrinsic"][None].cpu(), 0.001, 10.0)
    S = OracleSettings(512, 512, cm["tanfov"][0, 0].item(), cm["tanfov"][0, 1].item(), torch.zeros(4), 1.0,
                       cm["viewmatrix"][0].contiguous(), cm["projmatrix"][0].contiguous(), campos=cm["campos"][0],
                       render_mask=None, config=torch.tensor([1.0, 1.0, 1.0, 0.0, 0.0]))
    del gm
    torch.cuda.empty_cache()
    err, aux = _compare_case(a, S, f"mapper-grown map, {n} surfels 512x512 keyframe 17", max_tiles=160, fullest=12)
    assert aux["nonempty"] > 900 and aux["instances"] > 100_000


def test_c3_size_fused_iterations_match_the_autograd_mirror(agslib):
    """Configuration 3's shape (512x512 keyframes, batch 8 + 3, a mapper-grown map of ~200 k surfels): three iterations of
    the fused batched loop (HIP loss head, one per-Gaussian backward + Adam over the rows the views showed, iterations
    chained on the device) against the same three iterations through the reference-shaped mirror (facade + drop-in module
    under autograd + the torch loss head, dense Adam) from the same state and the same frame draws: losses, per-frame
    errors and where the parameters land."""
    import numpy as np
    from active_gs_amd.fused_map_trainer import FusedMapTrainer
    from active_gs_amd.gaussian_map import GaussianMap
    from active_gs_amd.map_trainer import GaussianMapTrainer
    from active_gs_amd.synthetic import make_keyframes, mapper_cfg
    dev = torch.device("cuda:0")
    frames = make_keyframes(12, 512, 512, dev, gt_surfels=400_000)
    np.random.seed(0)
    gm = GaussianMap(mapper_cfg(10, "device"), dev)
    for f in frames:
        gm.update(f)
    tr0 = gm._trainer
    n = tr0.means.shape[0]
    assert n > 100_000
    keys = ("means", "scales", "rotations", "opacities", "harmonics", "view_scores", "view_supports", "view_means")
    raw = {k: getattr(tr0, k).detach().clone() for k in keys}
    cfg = dict(tr0.cfg, optimization_steps=3, sampler="host", prune_interval=10 ** 9)
    perf0 = tr0.training_performance.clone()
    out = []
    for cls in (FusedMapTrainer, GaussianMapTrainer):
        t = cls({k: v.clone() for k, v in raw.items()}, list(tr0.frames), dict(cfg))
        t.training_performance = perf0.clone()
        np.random.seed(3)
        if cls is FusedMapTrainer:
            assert t._uniform_frames()
            assert t._train_batched(3) is True          # (the loop alone: no post-processing on either side)
        else:
            t.train(3)                                   # (its post-processing touches the view statistics only)
        torch.cuda.synchronize()
        out.append(({k: getattr(t, k).detach().clone() for k in keys[:5]}, t.training_performance.clone(), [float(x) for x in t.last_losses]))
    (pa, ea, la), (pb, eb, lb) = out
    assert len(la) == len(lb) == 3 and np.allclose(la, lb, rtol=2e-4), (la, lb)
    assert torch.allclose(ea, eb, rtol=1e-3, atol=1e-5)
    for k in keys[:5]:
        travel = (pb[k] - raw[k]).abs().mean()
        diff = (pa[k] - pb[k]).abs()
        assert float(travel) > 0 and float(diff.mean()) < 5e-3 * float(travel), (k, float(diff.mean()), float(travel))
