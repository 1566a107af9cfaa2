"""The drop-in module's call path itself (active-gs_amd/rasterizer.py + csrc/torch_binding.cpp): what an unmodified caller of
``diff_gaussian_rasterization_2d`` gets per call (/root/reference/utils/operations.py:682-713) - the configuration
read on the device, every call checked and repaired before it returns BY DEFAULT (an unmodified caller never sees
truncated tile lists or an exception for a legitimately larger view), workspaces checked one call late as an opt-in
(AGS_DROPIN_STATUS=deferred / rasterizer.deferred_status, which facade.SurfelRenderer settles per batch of views), the
fallback for skewed tile lists."""
import pytest
import torch

from _scenes import oracle_inputs, product_settings, room_case

pytestmark = pytest.mark.gpu


def _call(S, gin, settings=None, dev=None):
    from diff_gaussian_rasterization_2d import GaussianRasterizer
    return GaussianRasterizer(settings or product_settings(S, dev))(gin[0], gin[1], gin[2], gin[3], None, gin[4], gin[5], gin[6], None)


@pytest.fixture
def fresh_module():
    """every test starts from a module that has seen nothing"""
    import active_gs_amd.rasterizer as R
    names = ("always_check", "skew_factor", "direct_budget_bytes", "binning_mode")
    saved = {k: R.get_option(k) for k in names}
    R.reset_state()                     # (settles pending checks first)
    yield R
    R.reset_state()
    for k, v in saved.items():
        R.set_option(k, v)


def test_config_on_the_device_equals_config_on_the_host(agslib, fresh_module):
    """The reference hands `config` over as a device tensor (operations.py:697-699); the kernels read the four flags
    there.  A host tensor is decoded on the host (the template / int-flag path).  Same bits out, for every combination
    the reference uses and the two depth flags."""
    dev = torch.device("cuda:0")
    h, w = 96, 128
    gen = torch.Generator().manual_seed(5)
    mask = (torch.rand(1, h, w, generator=gen) > 0.3).float()
    for config in [(1, 1, 1, 0, 0), (1, 1, 1, 1, 1), (1, 0, 0, 0, 0), (1, 1, 0, 1, 0), (1, 0, 1, 0, 1)]:
        a, S = room_case(2500, h, w, view=4, seed=4, scale_mult=3.0, config=config, mask=mask)
        ins = oracle_inputs(a)
        outs, grads = [], []
        for where in ("device", "host"):
            gin = [t.detach().clone().to(dev).requires_grad_(t.requires_grad) for t in ins]
            s = product_settings(S, dev)
            if where == "host":
                s = s._replace(config=S.config.clone())         # a CPU tensor
            out = _call(S, gin, s)
            (out[0].sum() + out[1].sum() + 2 * out[2].sum()).backward()
            outs.append([o.detach().cpu() for o in out])
            grads.append([gin[i].grad.cpu() for i in (0, 2, 4, 5, 6)])
        for k in range(8):
            if k == 5:      # importance: float atomics over the waves that blended a surfel - order-dependent last bits
                assert float((outs[0][k] - outs[1][k]).abs().sum()) <= 1e-5 * float(outs[1][k].abs().sum()) + 1e-12
            else:
                assert torch.equal(outs[0][k], outs[1][k]), (config, k)
        for ga, gb in zip(*grads):                                # (the backward's atomics: order-dependent last bits)
            assert float((ga - gb).abs().sum()) <= 1e-5 * float(gb.abs().sum()) + 1e-12
        if not config[3]:
            assert int(outs[0][6].abs().sum()) == 0 and float(outs[0][5].abs().sum()) == 0.0
    fresh_module.check_overflow()


def test_no_host_synchronisation_once_a_view_size_is_known(agslib, fresh_module):
    """Deferred checks (opt-in): a call that finds a pooled workspace does not read anything back: the status block is
    copied to page-locked memory without waiting and looked at by a later call."""
    R = fresh_module
    R.set_option("always_check", 0)
    dev = torch.device("cuda:0")
    a, S = room_case(3000, 120, 160, view=0, seed=0, scale_mult=3.0)
    ins = oracle_inputs(a)
    gin = [t.detach().clone().to(dev).requires_grad_(t.requires_grad) for t in ins]
    c0 = R.counters()
    ref = None
    for it in range(6):
        out = _call(S, gin, dev=dev)
        out[0].sum().backward()
        if ref is None:
            ref = out[0].detach().clone()
        else:
            assert torch.equal(out[0].detach(), ref)
        for t in gin:
            t.grad = None
        del out
    c = R.counters()
    assert c["forward_calls"] - c0["forward_calls"] == 6
    assert c["status_syncs"] - c0["status_syncs"] == 1          # the first call made the workspace
    R.check_overflow()
    c = R.counters()
    assert c["deferred_checks"] - c0["deferred_checks"] == 5 and c["overflows"] == c0["overflows"] and c["pending"] == 0
    # the default (every call checked before it returns): one wait per call - with one-pass binning for the event behind
    # the per-Gaussian kernel (AgsWorkspace.early_status_*), not for the stream
    R.set_option("always_check", 1)
    c1 = R.counters()
    with torch.no_grad():
        for _ in range(3):
            _call(S, gin, dev=dev)
    c2 = R.counters()
    assert (c2["early_waits"] - c1["early_waits"]) + (c2["status_syncs"] - c1["status_syncs"]) == 3


def _grad_inputs(ts, dev):
    return [t.detach().clone().to(dev).requires_grad_(i in (0, 2, 4, 5, 6)) for i, t in enumerate(ts)]


def test_overflow_of_a_pooled_workspace_is_reported_one_call_late_and_repaired(agslib, fresh_module):
    """Deferred checks (opt-in, passes under grad): a view that outgrows the pooled workspace (here: the same surfels 40x
    larger) cannot be repaired inside the call that has already returned - the NEXT call / check_overflow() raises, the
    size is raised, and repeating the call gives the right image."""
    R = fresh_module
    R.set_option("always_check", 0)
    from active_gs_amd import raster_api as api
    dev = torch.device("cuda:0")
    n, h, w = 20000, 120, 160
    a, S = room_case(n, h, w, view=1, seed=1, scale_mult=1.0)
    ins = oracle_inputs(a, requires_grad=False)
    gin = _grad_inputs(ins, dev)
    _call(S, gin, dev=dev)                                            # sized for small surfels
    big = _grad_inputs(ins, dev)
    big[5] = (big[5].detach() * 40.0).requires_grad_(True)            # every visible surfel now covers most tiles
    bad = _call(S, big, dev=dev)                                      # pooled workspace: not checked yet
    with pytest.raises(RuntimeError, match="truncated"):
        R.check_overflow()
    assert R.counters()["overflows"] >= 1
    good = _call(S, big, dev=dev)                                     # a workspace is made for the size just learnt
    R.check_overflow()
    cam = api.Camera(h, w, S.tanfovx, S.tanfovy, S.viewmatrix.to(dev), S.projmatrix.to(dev), S.bg.to(dev))
    g = api.Gaussians(big[0].detach(), big[5].detach().contiguous(), big[6].detach(), big[2].detach().reshape(-1).contiguous(),
                      big[4].detach(), big[3])
    st = api.alloc_state(n, h, w, 1 << 22, dev, api.BIN_TILE_SORT)
    api.forward(cam, g, st)
    assert not api.read_status(st)["overflow"]
    assert torch.equal(good[0].detach(), st.rgb) and not torch.equal(bad[0].detach(), st.rgb)
    # ... and the error is raised by the next CALL too, before it enqueues anything
    R.reset_state()
    _call(S, gin, dev=dev)
    _call(S, big, dev=dev)
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="repeat the iteration"):
        _call(S, gin, dev=dev)
    # a pass WITHOUT grad is checked before it returns even with deferred checks switched on (its caller reads the
    # result on the host anyway): nothing truncated, nothing to report
    R.reset_state()
    with torch.no_grad():
        _call(S, [t.detach() for t in gin], dev=dev)
        fine = _call(S, [t.detach() for t in big], dev=dev)
    assert torch.equal(fine[0], st.rgb)
    R.check_overflow()


def _view_zoo(dev):
    """views of wildly different footprint of one room: near a wall, from a far corner, zoomed out, 128x128 and 1200x680"""
    from active_gs_amd.camera import camera_matrices
    from active_gs_amd.synthetic import make_camera
    zoo = []
    gen = torch.Generator().manual_seed(11)
    for k in range(50):
        h, w = ((128, 128), (96, 160), (340, 600), (680, 1200))[k % 4] if k % 7 else (128, 128)
        c2w, K = make_camera(k, h, w)
        kind = k % 5
        if kind == 0:        # nose against a wall: few surfels, each over hundreds of tiles
            c2w = c2w.clone(); c2w[:3, 3] = torch.tensor([2.35, 0.0, 0.0]) * (1 if k % 2 else -1) + 0.05 * torch.randn(3, generator=gen)
        elif kind == 1:      # a far corner looking across the room: most of the map in a few tiles
            c2w = c2w.clone(); c2w[:3, 3] = torch.tensor([-2.3, -1.8, 1.2]) * (1 if k % 2 else -1)
        cm = camera_matrices(c2w[None], K[None], 0.001, 10.0)
        zoo.append((h, w, cm))
    return zoo


def test_default_is_safe_for_an_unmodified_caller(agslib, fresh_module):
    """The reference has no retry anywhere above the rasterizer (mapping/mapper.py:98-104, utils/operations.py:682-713):
    with the module's DEFAULT settings 50 views of wildly different footprint and size - under grad and without, no
    check_overflow() anywhere - raise nothing and every image equals the one a generously sized, per-call-checked
    workspace renders."""
    R = fresh_module
    assert R.get_option("always_check") == 1.0
    c0 = R.counters()
    from active_gs_amd import raster_api as api
    from diff_gaussian_rasterization_2d import GaussianRasterizationSettings, GaussianRasterizer
    dev = torch.device("cuda:0")
    n = 30000
    a, _ = room_case(n, 64, 64, view=0, seed=9, scale_mult=2.5)
    ins = oracle_inputs(a, requires_grad=False)
    base = [t.to(dev) for t in ins]
    g = api.Gaussians(base[0], base[5].contiguous(), base[6], base[2].reshape(-1).contiguous(), base[4], base[3])
    bg = torch.tensor([0.1, 0.2, 0.3, 0.0], device=dev)
    for k, (h, w, cm) in enumerate(_view_zoo(dev)):
        s = GaussianRasterizationSettings(image_height=h, image_width=w, tanfovx=float(cm["tanfov"][0, 0]),
                                          tanfovy=float(cm["tanfov"][0, 1]), bg=bg, scale_modifier=1.0,
                                          viewmatrix=cm["viewmatrix"][0].to(dev), projmatrix=cm["projmatrix"][0].to(dev),
                                          sh_degree=0, campos=cm["campos"][0].to(dev), prefiltered=False,
                                          render_mask=torch.tensor([], device=dev), weight_thres=0.03, debug=False,
                                          config=torch.tensor([1.0, 1, 1, 0, 0]).to(dev))
        grad = k % 2 == 0
        gin = _grad_inputs(ins, dev) if grad else base
        with torch.set_grad_enabled(grad):
            out = GaussianRasterizer(s)(gin[0], gin[1], gin[2], gin[3], None, gin[4], gin[5], gin[6], None)
        if grad:
            (out[0].sum() + out[2].sum()).backward()
            assert bool(torch.isfinite(gin[0].grad).all())
        cam = api.Camera(h, w, s.tanfovx, s.tanfovy, s.viewmatrix, s.projmatrix, bg)
        st = api.alloc_state(n, h, w, 1 << 23, dev, api.BIN_TILE_SORT)
        api.forward(cam, g, st)
        assert not api.read_status(st)["overflow"]
        assert torch.equal(out[0].detach(), st.rgb) and torch.equal(out[2].detach(), st.depth) and torch.equal(out[7], st.radii), k
    R.check_overflow()                                                     # (the sizing copies of the last calls: nothing to report)
    c = R.counters()
    assert c["overflows"] == c0["overflows"] and c["pending"] == 0        # no truncated pass was ever left to a later call
    assert (c["status_syncs"] - c0["status_syncs"]) + (c["early_waits"] - c0["early_waits"]) >= 50      # every call was checked
    assert c["repaired"] > c0["repaired"]        # ... and some of these views did outgrow a pooled workspace: repaired in the call


def test_surfel_renderer_settles_its_views_once_per_batch_and_repairs(agslib, fresh_module):
    """facade.SurfelRenderer (the GaussianRenderer mirror) runs its view loops with the module's checks deferred and waits
    ONCE per batch; a view that outgrew its pooled workspace makes the loop run again - same images as the checked
    path, no exception, and under grad the gradients are those of the repaired pass."""
    R = fresh_module
    from active_gs_amd.facade import SurfelRenderer
    from active_gs_amd.synthetic import make_camera
    dev = torch.device("cuda:0")
    n, h, w = 20000, 120, 160
    a, _ = room_case(n, h, w, view=1, seed=1, scale_mult=1.0)
    cams = [make_camera(v, h, w) for v in range(4)]
    c2w = torch.stack([c[0] for c in cams]).to(dev); K = torch.stack([c[1] for c in cams]).to(dev)
    bg = torch.zeros(4, device=dev)

    def attr(scale, grad):
        t = lambda x: x.to(dev).clone().requires_grad_(grad)
        return (t(a["means"]), t(a["colors"][:, None, :]), t(a["opacities"]), a["confidences"].to(dev), t(a["scales"] * scale),
                t(a["rotations"]))

    small = attr(1.0, True)
    SurfelRenderer(c2w, K, small, bg, (0.001, 10.0), (h, w), dev).render_view_all(require_grad=True)     # sizes learnt: small surfels
    c0 = R.counters()
    big = attr(40.0, True)
    outs = SurfelRenderer(c2w, K, big, bg, (0.001, 10.0), (h, w), dev).render_view_all(require_grad=True)
    c1 = R.counters()
    assert c1["overflows"] > c0["overflows"]                      # the first pass over the views was truncated ... and repeated
    (outs[0].sum() + outs[1].sum()).backward()
    R.set_option("always_check", 1)
    R.reset_state()
    big2 = attr(40.0, True)
    ref = [SurfelRenderer(c2w[i:i + 1], K[i:i + 1], big2, bg, (0.001, 10.0), (h, w), dev).render_view(0, require_grad=True)
           for i in range(4)]
    for i in range(4):
        for k in (0, 1, 2, 3, 4, 5):
            assert torch.equal(outs[k][i].detach(), ref[i][k].detach()), (i, k)
    sum(r[0].sum() + r[1].sum() for r in ref).backward()
    for x, y in zip(big, big2):
        if x.requires_grad:
            assert float((x.grad - y.grad).abs().sum()) <= 1e-4 * float(y.grad.abs().sum()) + 1e-12


def test_skewed_tile_lists_fall_back_to_scan_based_binning(agslib, fresh_module):
    """One-pass binning needs tiles x the LONGEST list of key slots.  When that is far above the instance total (a
    distant camera: most surfels in a few tiles) the view size moves to the scan-based binning - same image."""
    R = fresh_module
    from active_gs_amd import raster_api as api
    dev = torch.device("cuda:0")
    n, h, w = 3000, 128, 160
    a, S = room_case(n, h, w, view=2, seed=2, scale_mult=2.0)
    a["means"] = a["means"] * 0.05 + torch.tensor([0.0, 0.0, 0.0])       # everything in a small clump
    ins = oracle_inputs(a, requires_grad=False)
    gin = [t.to(dev) for t in ins]
    R.set_option("skew_factor", 1e-9); R.set_option("direct_budget_bytes", 0)      # (thresholds a test scene can reach)
    with torch.no_grad():
        first = _call(S, gin, dev=dev)
        assert R.state()["mode_for"].get((0, h, w)) == api.BIN_TILE_SORT and R.counters()["mode_switches"] >= 1
        second = _call(S, gin, dev=dev)
        R.check_overflow()
    assert any(k[-1] == api.BIN_TILE_SORT for k in R.state()["pooled"])
    for k in range(8):
        assert torch.equal(first[k], second[k]), k


def test_a_growing_map_does_not_leave_a_set_of_workspaces_per_size_behind(agslib, fresh_module):
    """The mapper's map changes size at every keyframe (gaussian_map.py:294-468, :234-246).  Pooled workspaces are keyed
    by image size and binning mode only: a slab laid out for another map size is re-initialised when it is large enough,
    dropped when it is not - the pool holds a bounded number of slabs however many sizes it has seen."""
    R = fresh_module
    dev = torch.device("cuda:0")
    h, w = 96, 128
    ref = {}
    for rnd in range(2):
        for n in (3000, 1800, 2600, 4100, 2200, 5000, 900):          # grows and shrinks
            a, S = room_case(n, h, w, view=3, seed=n, scale_mult=3.0)
            ins = oracle_inputs(a)
            gin = [t.detach().clone().to(dev).requires_grad_(t.requires_grad) for t in ins]
            outs = [_call(S, gin, dev=dev) for _ in range(3)]           # three views alive at once, like a training batch
            sum(o[0].sum() for o in outs).backward()
            img = outs[0][0].detach().clone()
            if n in ref:
                assert torch.equal(img, ref[n])                          # a re-initialised slab renders the same bits
            ref[n] = img
            del outs
    R.check_overflow()
    pooled = R.state()["pooled"]
    assert len(pooled) == 1 and next(iter(pooled.values()))["count"] <= 3, pooled
