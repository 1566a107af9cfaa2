"""Fused loss head (csrc/loss.hip) vs torch autograd of the facade mirror, and the fully fused
train loop vs the reference's GaussianMap.train() capture."""
import os

import numpy as np
import pytest
import torch

from _scenes import room_case

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_fused_loss_matches_torch_autograd(agslib):
    from active_gs_amd import raster_api as api
    from active_gs_amd.facade import _depth_to_normal_torch as depth_to_normal, training_losses   # (the torch statement)
    from active_gs_amd.fused_loss import FusedLoss
    dev = torch.device("cuda:0")
    n, h, w, B = 6000, 96, 128, 3
    a, _ = room_case(n, h, w, view=0, seed=12, scale_mult=3.0)
    a["opacities"] = a["opacities"] * 0.6   # leave holes: opacity crosses the 1e-2 / 1e-3 masks
    g = api.Gaussians(*(a[k].to(dev).contiguous() for k in ("means", "scales", "rotations", "opacities", "colors",
                                                              "confidences")))
    states, fovs = [], None
    for v in range(B):
        _, S = room_case(16, h, w, view=v, seed=0)
        cam = api.Camera(h, w, S.tanfovx, S.tanfovy, S.viewmatrix.to(dev), S.projmatrix.to(dev), S.bg.to(dev))
        st = api.alloc_state(n, h, w, 1 << 21, dev)
        api.forward(cam, g, st)
        states.append(st)
        fovs = (2 * np.arctan(S.tanfovx), 2 * np.arctan(S.tanfovy))
    gen = torch.Generator().manual_seed(3)
    gts = []
    for st in states:
        gt_rgb = (st.rgb.cpu() + 0.1 * torch.randn(3, h, w, generator=gen)).clamp(0, 1).to(dev)
        gt_depth = (st.depth.cpu() * (1 + 0.05 * torch.randn(1, h, w, generator=gen))).to(dev)
        gt_depth[:, :, : w // 5] = 0.0           # invalid depth region
        gt_depth[:, h // 2, :] = -1.0
        gts.append((gt_rgb, gt_depth))
    # ---- torch reference on the same rasterizer outputs
    leaves = [[getattr(st, k).clone().requires_grad_(True) for k in ("rgb", "normal", "depth")] for st in states]
    fov_t = torch.tensor(fovs)
    posts = []
    for st, (rgb, nrm, dep) in zip(states, leaves):
        mask = st.opacity > 1e-2
        normal = torch.nn.functional.normalize(nrm, dim=0) * mask
        posts.append((rgb, dep, normal, st.opacity, depth_to_normal(dep, mask, fov_t)))
    stack = lambda k: torch.stack([p[k] for p in posts])
    total, per_frame = training_losses(stack(0), stack(1), stack(2), stack(3), stack(4),
                                       torch.stack([x[0] for x in gts]), torch.stack([x[1] for x in gts]))
    total.backward()
    # ---- fused kernels
    fl = FusedLoss(h, w, fovs[0], fovs[1], B, 8, dev)
    bufs = [fl.alloc_view() for _ in range(B)]
    fl.begin_step()
    for v, st in enumerate(states):
        fl.stage1(st, gts[v][0], gts[v][1], bufs[v], v, v == 0)
    for v, st in enumerate(states):
        fl.stage2(st, gts[v][1], bufs[v])
    torch.cuda.synchronize()
    total_ref = float(total.detach())
    assert abs(float(fl.total_loss()) - total_ref) < 1e-5 * max(1.0, abs(total_ref))
    assert torch.allclose(fl.per_frame_errors(B), per_frame, rtol=1e-4, atol=1e-6)
    for v in range(B):
        for name, ref, got in (("rgb", leaves[v][0].grad, bufs[v].d_rgb), ("normal", leaves[v][1].grad, bufs[v].d_normal),
                               ("depth", leaves[v][2].grad, bufs[v].d_depth)):
            denom = ref.abs().sum().item()
            rel = (got - ref).abs().sum().item() / max(denom, 1e-12)
            assert rel < 1e-3, (v, name, rel, denom)


def test_fused_train_loop_matches_reference_train_capture(agslib):
    from active_gs_amd.fused_map_trainer import FusedMapTrainer
    dev = torch.device("cuda:0")
    d = torch.load(os.path.join(GOLD, "train.pt"))
    cfg = d["cfg"]
    mine = dict(bound=tuple(cfg["bound"]), scale_factor=cfg["scale_factor"], optimization_steps=cfg["optimization_steps"],
                prune_interval=cfg["prune_interval"], background=tuple(cfg["background"]),
                batch_size=cfg["sampler"]["batch_size"], active_size=cfg["sampler"]["active_size"],
                use_view_distribution=cfg["use_view_distribution"],
                lrs=dict(mean=cfg["optimizer"]["mean_lr"], scale=cfg["optimizer"]["scale_lr"],
                         rotation=cfg["optimizer"]["rotation_lr"], opacity=cfg["optimizer"]["opacity_lr"],
                         harmonic=cfg["optimizer"]["harmonic_lr"]))
    raw = {k: v.to(dev) for k, v in d["raw_init"].items()}
    frames = [{k: v.to(dev) for k, v in f.items()} for f in d["frames"]]
    t = FusedMapTrainer(raw, frames, mine)
    np.random.seed(7)
    t.train()
    torch.cuda.synchronize()
    for k, ref in d["raw_final"].items():
        got, init = getattr(t, k).cpu(), d["raw_init"][k]
        diff, travel = (got - ref).abs(), (ref - init).abs().mean()
        assert diff.mean() < 2e-3 * travel, (k, float(diff.mean()), float(travel))
        assert (diff > 1e-4).float().mean() < 5e-3, k
    assert torch.allclose(t.training_performance.cpu(), d["training_performance"], rtol=1e-3, atol=1e-5)
    assert (t.view_supports.cpu() != d["view_supports"]).float().mean() < 2e-3
    assert len(t.last_losses) == d["steps"] and all(np.isfinite(t.last_losses))


def test_graph_replayed_fused_iteration_equals_eager(agslib):
    """train_graph() (first iteration eager, the rest replayed from one hipGraph, frames staged into
    static buffers) lands on the same parameters as train()."""
    from active_gs_amd.fused_map_trainer import FusedMapTrainer
    dev = torch.device("cuda:0")
    d = torch.load(os.path.join(GOLD, "train.pt"))
    cfg = d["cfg"]
    mine = dict(bound=tuple(cfg["bound"]), scale_factor=cfg["scale_factor"], optimization_steps=5,
                prune_interval=1000, background=tuple(cfg["background"]), batch_size=3, active_size=2,
                use_view_distribution=cfg["use_view_distribution"])
    res = []
    for graph in (False, True):
        raw = {k: v.to(dev) for k, v in d["raw_init"].items()}
        frames = [{k: v.to(dev) for k, v in f.items()} for f in d["frames"]]
        t = FusedMapTrainer(raw, frames, mine)
        np.random.seed(11)           # 4 frames, batch 3 = 2 newest + 1 error-weighted draw per iteration
        (t.train_graph if graph else t.train)()
        torch.cuda.synchronize()
        if graph:
            assert t._graph is not None
        res.append(([getattr(t, k).clone() for k in ("means", "scales", "rotations", "opacities", "harmonics")],
                    t.training_performance.clone(), list(t.last_losses)))
    for a, b, init in zip(res[0][0], res[1][0], [d["raw_init"][k] for k in ("means", "scales", "rotations", "opacities", "harmonics")]):
        travel = (a.cpu() - init).abs().mean()
        assert (a - b).abs().mean() < 2e-3 * travel + 1e-9
    assert torch.allclose(res[0][1], res[1][1], rtol=1e-3, atol=1e-5)
    assert np.allclose(res[0][2], res[1][2], rtol=1e-4)


def test_batched_iteration_equals_per_view_and_replays_from_a_graph(agslib):
    """train_batched() (all views of an iteration in one set of launches; optionally the iteration
    replayed from a hipGraph) lands where the per-view train() lands."""
    from active_gs_amd.fused_map_trainer import FusedMapTrainer
    dev = torch.device("cuda:0")
    d = torch.load(os.path.join(GOLD, "train.pt"))
    cfg = d["cfg"]
    mine = dict(bound=tuple(cfg["bound"]), scale_factor=cfg["scale_factor"], optimization_steps=5,
                prune_interval=1000, background=tuple(cfg["background"]), batch_size=3, active_size=2,
                use_view_distribution=cfg["use_view_distribution"])
    res = []
    for variant in ("per_view", "batched", "batched_graph"):
        raw = {k: v.to(dev) for k, v in d["raw_init"].items()}
        frames = [{k: v.to(dev) for k, v in f.items()} for f in d["frames"]]
        t = FusedMapTrainer(raw, frames, mine, batched=variant != "per_view", num_streams=1)
        t.graph_min_steps = 0 if variant == "batched_graph" else 10 ** 9
        np.random.seed(11)
        t.train()
        torch.cuda.synchronize()
        res.append(([getattr(t, k).clone() for k in ("means", "scales", "rotations", "opacities", "harmonics")],
                    t.training_performance.clone(), list(t.last_losses)))
    init = [d["raw_init"][k] for k in ("means", "scales", "rotations", "opacities", "harmonics")]
    for other in res[1:]:
        for a, b, i0 in zip(res[0][0], other[0], init):
            travel = (a.cpu() - i0).abs().mean()
            assert (a - b).abs().mean() < 2e-3 * travel + 1e-9
        assert torch.allclose(res[0][1], other[1], rtol=1e-3, atol=1e-5)
        assert np.allclose(res[0][2], other[2], rtol=1e-4)


def test_stage_frames_and_loss_finish_match_torch(agslib):
    """ags_stage_frames == four index_select calls (+ zeroed visibility count); ags_loss_finish ==
    FusedLoss.per_frame_errors / total_loss computed with torch, and it leaves the accumulators clean."""
    from active_gs_amd.fused_loss import FusedLoss
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(21)
    K, B, h, w = 7, 4, 36, 52
    all_view = torch.randn(K, 4, 4, generator=gen).to(dev)
    all_proj = torch.randn(K, 4, 4, generator=gen).to(dev)
    all_rgb = torch.rand(K, 3, h, w, generator=gen).to(dev)
    all_depth = torch.rand(K, 1, h, w, generator=gen).to(dev)
    loss = FusedLoss(h, w, 1.0, 0.9, B, 8, dev)
    loss.msum.fill_(5)
    idx = torch.tensor([6, 0, 3, 3], device=dev)
    dv, dp = torch.zeros(8, 4, 4, device=dev), torch.zeros(8, 4, 4, device=dev)
    dr, dd = torch.zeros(8, 3, h, w, device=dev), torch.zeros(8, 1, h, w, device=dev)
    loss.stage_frames(B, idx, all_view, all_proj, all_rgb, all_depth, dv, dp, dr, dd)
    assert torch.equal(dv[:B], all_view[idx]) and torch.equal(dp[:B], all_proj[idx])
    assert torch.equal(dr[:B], all_rgb[idx]) and torch.equal(dd[:B], all_depth[idx])
    assert float(dv[B:].abs().sum()) == 0 and float(dr[B:].abs().sum()) == 0
    assert int(loss.msum.abs().sum()) == 0
    # accumulators as the loss stages would leave them
    loss.accum.copy_(torch.rand(loss.accum.shape, generator=gen).to(dev) * 100)
    want_err = loss.per_frame_errors(B).clone()
    want_total = loss.total_loss().clone()
    perf = torch.full((K,), 10.0, device=dev)
    total = torch.zeros(1, device=dev)
    idx2 = torch.tensor([6, 0, 3, 5], device=dev)
    loss.finish(B, idx2, perf, total)
    assert torch.allclose(perf[idx2], want_err, rtol=1e-5) and float(perf[1]) == 10.0
    assert torch.allclose(total[0], want_total, rtol=1e-5)
    assert float(loss.accum.abs().sum()) == 0.0


def test_zero_many_clears_exactly_the_given_regions(agslib):
    """ags_zero_many: regions of any whole-word size (16-byte body + word tail), up to sixteen per launch (more: several
    launches), nothing touched outside them; misaligned / odd-sized regions are refused."""
    from active_gs_amd import _lib
    dev = torch.device("cuda:0")
    sizes = [1, 3, 4, 5, 16, 17, 1023, 4096, 100_003, 7, 64, 2, 33, 255, 256, 257, 9, 1_000_001]      # floats (18 regions)
    bufs = [torch.full((s + 8,), 3.0, device=dev) for s in sizes]
    views = [b[4:4 + s] for b, s in zip(bufs, sizes)]                                # 16-byte aligned starts inside guards
    _lib.zero_many(views + [None, torch.empty(0, device=dev)])
    torch.cuda.synchronize()
    for b, s in zip(bufs, sizes):
        assert float(b[4:4 + s].abs().sum()) == 0.0 and bool((b[:4] == 3.0).all()) and bool((b[4 + s:] == 3.0).all()), s
    with pytest.raises(ValueError):
        _lib.zero_many([bufs[0][1:3]])                                               # not 16-byte aligned
    h = torch.zeros(8, device=dev, dtype=torch.float16)
    with pytest.raises(ValueError):
        _lib.zero_many([h[:3]])                                                      # 6 bytes


@pytest.mark.parametrize("k_random", [3, 0])
def test_finish_next_equals_finish_then_draw_then_stage(agslib, k_random):
    """ags_loss_finish_next == ags_loss_finish, then ags_weighted_topk over the errors just written, then ags_stage_frames
    (matrices only, visibility count cleared): the same errors, loss, indices and staged matrices, bit for bit."""
    from active_gs_amd import _lib
    from active_gs_amd.fused_loss import FusedLoss
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(5)
    K, h, w, n_active = 40, 36, 52, 2
    B = n_active + k_random
    n_old = K - n_active
    all_view = torch.randn(K, 4, 4, generator=gen).to(dev)
    all_proj = torch.randn(K, 4, 4, generator=gen).to(dev)
    u = torch.rand(n_old, generator=gen).to(dev)
    acc0 = (torch.rand(64, 4 + 2 * 8, generator=gen) * 100).to(dev)
    idx0 = torch.tensor([K - 1, K - 2, 7, 0, 21][:B], device=dev)
    perf0 = (torch.rand(K, generator=gen) * 0.2 + 0.01).to(dev)

    def separate():
        loss = FusedLoss(h, w, 1.0, 0.9, B, 8, dev)
        loss.accum.copy_(acc0); loss.msum.fill_(3)
        idx, perf, total = idx0.clone(), perf0.clone(), torch.zeros(1, device=dev)
        dv, dp = torch.zeros(8, 4, 4, device=dev), torch.zeros(8, 4, 4, device=dev)
        loss.finish(B, idx, perf, total)
        if k_random:
            _lib.check(_lib.load().ags_weighted_topk(_lib.ptr(u), _lib.ptr(perf), n_old, k_random, _lib.ptr(idx[n_active:]),
                                                     _lib.current_stream()), "ags_weighted_topk")
        loss.stage_frames(B, idx, all_view, all_proj, None, None, dv, dp, None, None)
        return idx, perf, total, dv, dp, loss

    def fused():
        loss = FusedLoss(h, w, 1.0, 0.9, B, 8, dev)
        loss.accum.copy_(acc0); loss.msum.fill_(3)
        idx, perf, total = idx0.clone(), perf0.clone(), torch.zeros(1, device=dev)
        dv, dp = torch.zeros(8, 4, 4, device=dev), torch.zeros(8, 4, 4, device=dev)
        loss.finish_next(B, idx, perf, total, u if k_random else None, n_old, k_random, n_active, all_view, all_proj, dv, dp)
        return idx, perf, total, dv, dp, loss

    a, b = separate(), fused()
    for x, y in zip(a[:5], b[:5]):
        assert torch.equal(x, y)
    assert int(b[5].msum.abs().sum()) == 0 and float(b[5].accum.abs().sum()) == 0.0
    if k_random:
        drawn = b[0][n_active:].tolist()
        assert len(set(drawn)) == k_random and all(0 <= i < n_old for i in drawn)
        assert not torch.equal(b[0], idx0)                    # (the draw happened)
    else:
        assert torch.equal(b[0], idx0)
    assert torch.equal(b[3][:B], all_view[b[0]]) and torch.equal(b[4][:B], all_proj[b[0]])


@pytest.mark.parametrize("h,w", [(64, 96), (97, 51), (16, 16)])
def test_facade_post_kernel_matches_the_torch_statements(agslib, h, w):
    """render_cuda_core's two post-processing statements (operations.py:714-718: normalize(normal) * (opacity > 1e-2),
    depth2normal with replicate padding and the fov/H pairing) as ONE launch and ONE backward launch
    (ags_facade_post[_backward]) against their torch statement, values and gradients wrt the raw normal and the depth;
    holes in the opacity (masked centres and masked neighbours), non-square and one-tile images."""
    import math
    import torch.nn.functional as F
    from active_gs_amd.facade import _FacadePost, _depth_to_normal_torch, depth_to_normal
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(h * 1000 + w)
    ys, xs = torch.meshgrid(torch.linspace(0, 1, h), torch.linspace(0, 1, w), indexing="ij")
    depth = (1.5 + 0.8 * xs + 0.3 * torch.sin(6 * ys) + 0.003 * torch.randn(h, w, generator=gen))[None]
    normal = torch.randn(3, h, w, generator=gen)
    opacity = torch.rand(1, h, w, generator=gen)
    opacity[:, h // 3: h // 3 + 3, :] = 0.0                      # a masked band, masked borders
    opacity[:, :, 0] = 0.005
    opacity[0, -1, -1] = 0.0
    fov = torch.tensor([math.radians(70.0), math.radians(50.0)])
    tan = [math.tan(float(f) / 2) for f in fov]
    g_n, g_d = torch.randn(3, h, w, generator=gen), torch.randn(3, h, w, generator=gen)
    res = []
    for native in (False, True):
        # the torch statement in float64: in float32 its own (padded point - centre) cancellations cost it 2.5e-4 of the
        # depth gradient (relative L1 against float64, 22 of 228 at the worst pixel); the kernel's rounded products
        # keep 8e-7
        dt = torch.float32 if native else torch.float64
        N = normal.clone().to(dev, dt).requires_grad_(True)
        D = depth.clone().to(dev, dt).requires_grad_(True)
        O = opacity.to(dev)
        if native:
            n_out, d2n = _FacadePost.apply(N, D, O, tan[0], tan[1])
            alone = depth_to_normal(D.detach(), O > 1e-2, fov)         # the public function takes the kernel too
            assert torch.equal(alone, d2n.detach())
        else:
            mask = O > 1e-2
            n_out = F.normalize(N, dim=0) * mask
            d2n = _depth_to_normal_torch(D, mask, fov.to(dev, dt))
        ((n_out * g_n.to(dev, dt)).sum() + (d2n * g_d.to(dev, dt)).sum()).backward()
        res.append([t.detach().cpu().double() for t in (n_out, d2n, N.grad, D.grad)])
    for k, name in enumerate(("normal", "d2n", "d_normal_raw", "d_depth")):
        a, b = res[1][k], res[0][k]
        scale = float(b.abs().max()) + 1e-12
        diff = (a - b).abs()
        assert float(diff.max()) <= 5e-5 * scale + 1e-6, (name, float(diff.max()), scale)
        assert float(diff.sum()) <= 1e-5 * float(b.abs().sum()) + 1e-6, (name, float(diff.sum()), float(b.abs().sum()))


@pytest.mark.parametrize("h,w,views,use_index", [(64, 96, 3, True), (40, 52, 2, False), (136, 240, 4, True)])
def test_loss_stage1_as_forward_epilogue_equals_the_separate_launch(agslib, h, w, views, use_index):
    """``ags_forward_batch_loss``: stage 1 of the loss head as the epilogue of the forward blend kernel against
    ``ags_forward_batch`` followed by ``ags_loss_stage1`` - the rendered images, the post-processed normal image, d_rgb,
    d_depth and the visibility count bit for bit (same operations, same roundings), the L1 sums equal up to the order of
    their additions (per wave of 64 pixels instead of per block of 256); partial tiles at the image border (40x52), the
    ground truth read in place from a keyframe store through an index, and taken in batch order."""
    from active_gs_amd import raster_api as api
    from active_gs_amd.fused_loss import FusedLoss
    from active_gs_amd.synthetic import make_room_scene, activate
    dev = torch.device("cuda:0")
    n = 6000
    a = activate({k: v.to(dev) for k, v in make_room_scene(n, seed=4).items()})
    a["scales"] = a["scales"] * 2.5
    g = api.Gaussians(a["means"], a["scales"].contiguous(), a["rotations"], a["opacities"], a["colors"].contiguous(), a["confidences"])
    _, S0 = room_case(16, h, w, view=0, seed=0)
    gen = torch.Generator().manual_seed(5)
    store = 6
    gt_rgb = torch.rand(store, 3, h, w, generator=gen).to(dev)
    gt_depth = (torch.rand(store, 1, h, w, generator=gen) * 3 - 0.3).to(dev)        # some pixels without ground truth (<= 0)
    idx = torch.tensor([4, 0, 5, 2][:views], device=dev, dtype=torch.long) if use_index else None
    if not use_index:
        gt_rgb, gt_depth = gt_rgb[:views].contiguous(), gt_depth[:views].contiguous()
    out = {}
    for form in ("separate", "epilogue"):
        batch = api.ViewBatch(g, views, h, w, S0.tanfovx, S0.tanfovy, S0.bg.to(dev), 1 << 20)
        for v in range(views):
            _, S = room_case(16, h, w, view=v, seed=0)
            batch.viewmats[v].copy_(S.viewmatrix.to(dev)); batch.projmats[v].copy_(S.projmatrix.to(dev))
        loss = FusedLoss(h, w, 2 * np.arctan(S0.tanfovx), 2 * np.arctan(S0.tanfovy), views, views, dev)
        bufs = loss.alloc_batch(views)
        for t in (bufs.n_img, bufs.d_rgb, bufs.d_depth):
            t.fill_(7.0)
        if form == "separate":
            batch.forward(views)
            loss.stage1_batch(batch._structs()[0], gt_rgb, gt_depth, bufs, views, gt_index=idx)
        else:
            batch.forward(views, loss=loss.epilogue(gt_rgb, gt_depth, bufs, gt_index=idx))
        torch.cuda.synchronize()
        assert int(batch.statuses(views)[:, 2].max()) == 0
        out[form] = dict(rgb=batch.rgb.clone(), normal=batch.normal.clone(), depth=batch.depth.clone(), opacity=batch.opacity.clone(),
                         n_img=bufs.n_img.clone(), d_rgb=bufs.d_rgb.clone(), d_depth=bufs.d_depth.clone(), msum=loss.msum.clone(),
                         accum=loss.accum.sum(0).clone())
    s, e = out["separate"], out["epilogue"]
    for k in ("rgb", "normal", "depth", "opacity", "n_img", "d_rgb", "d_depth", "msum"):
        assert torch.equal(s[k], e[k]), (k, float((s[k].float() - e[k].float()).abs().max()), int((s[k] != e[k]).sum()))
    assert float(s["opacity"].max()) > 0.5 and int(s["msum"].max()) >= 1 and float(s["d_depth"].abs().max()) > 0
    assert float(s["accum"][:2].min()) > 0 and torch.allclose(s["accum"], e["accum"], rtol=2e-5, atol=1e-6)
    # argument checks: statistics cannot ride along, the accumulator rows must hold the views
    batch.cam.want_stats = True
    with pytest.raises(RuntimeError):
        batch.forward(views, loss=loss.epilogue(gt_rgb, gt_depth, bufs, gt_index=idx))
    batch.cam.want_stats = False
