"""HIP path (through the drop-in module -> ctypes -> C ABI) vs the CPU oracle.

Tolerances are BASELINE.json's: <= 1e-4 mean-L1 on rendered RGB, <= 1e-3 relative L1 on
gradients (the backward's atomics make summation order nondeterministic)."""
import pytest
import torch

import _parity
from _scenes import oracle_inputs, product_settings, room_case

pytestmark = pytest.mark.gpu

RGB_TOL = 1e-4
GRAD_TOL = 1e-3


@pytest.fixture(params=["direct", "tile_sort", "radix"])
def binning(request):
    """Run the parity cases with all three binning algorithms of the C ABI."""
    import active_gs_amd.rasterizer as R
    from active_gs_amd import raster_api as api
    old = R.get_option("binning_mode")
    R.set_option("binning_mode", {"direct": api.BIN_DIRECT, "tile_sort": api.BIN_TILE_SORT, "radix": api.BIN_RADIX}[request.param])
    yield request.param
    R.set_option("binning_mode", old)


def _run_both(a, S, seed=0, grad_channels=(1, 1, 1, 1, 1)):
    from diff_gaussian_rasterization_2d import GaussianRasterizer
    from oracle.surfel_oracle import rasterize
    dev = torch.device("cuda:0")
    ins = oracle_inputs(a)
    ref = rasterize(*ins, S)
    gen = torch.Generator().manual_seed(seed)
    gr = [torch.randn(o.shape, generator=gen) * c for o, c in zip(ref[:5], grad_channels)]
    sum((o * g).sum() for o, g in zip(ref[:5], gr)).backward()
    gin = [t.detach().clone().to(dev).requires_grad_(t.requires_grad) for t in ins]
    out = GaussianRasterizer(product_settings(S, dev))(
        means3D=gin[0], means2D=gin[1], opacities=gin[2], confidences=gin[3], shs=None, colors_precomp=gin[4],
        scales=gin[5], rotations=gin[6], cov3D_precomp=None)
    sum((o * g.to(dev)).sum() for o, g in zip(out[:5], gr)).backward()
    torch.cuda.synchronize()
    from diff_gaussian_rasterization_2d import check_overflow
    check_overflow()         # the module checks its workspaces one call late (rasterizer.STATUS_CHECK): settle it here
    return ins, ref, gin, out


def _check_images(ref, out, what=""):
    """contract mean-L1 + largest pixel error + worst-tile mean L1 + regression gate (tests/_parity.py)"""
    return _parity.check_images([r.detach() for r in ref[:5]], [o.cpu() for o in out[:5]], what=what)


def _check_grads(ins, gin, what=""):
    names = {0: "means3D", 1: "means2D", 2: "opacities", 4: "colors", 5: "scales", 6: "rotations"}
    return _parity.check_grads({n: ins[i].grad for i, n in names.items()}, {n: gin[i].grad for i, n in names.items()}, what=what)


@pytest.mark.parametrize("n,h,w,view,mult", [(3000, 120, 160, 0, 3.0), (5000, 170, 300, 1, 2.0), (800, 64, 64, 2, 4.0),
                                             (2000, 100, 150, 3, 3.0)])
def test_forward_backward_matches_oracle(agslib, binning, n, h, w, view, mult):
    a, S = room_case(n, h, w, view=view, seed=view, scale_mult=mult)
    ins, ref, gin, out = _run_both(a, S, seed=view)
    what = f"{n} surfels {w}x{h} view {view} x{mult} {binning}"
    _check_images(ref, out, what)
    assert out[7].dtype == torch.int32 and out[6].dtype == torch.int32
    from oracle.surfel_oracle import preprocess
    with torch.no_grad():
        G = preprocess(*[t.detach() for t in ins], S)
    _parity.radii_report(out[7], G, ins, S, what)    # exact but for rows on a rounding boundary (counted, explained)
    _check_grads(ins, gin, what)


def test_importance_count_front_only_mask(agslib, binning):
    h, w = 96, 128
    gen = torch.Generator().manual_seed(5)
    mask = (torch.rand(1, h, w, generator=gen) > 0.3).float()
    a, S = room_case(2500, h, w, view=4, seed=4, scale_mult=3.0, config=(1, 1, 1, 1, 1), mask=mask)
    ins, ref, gin, out = _run_both(a, S, seed=9)
    _check_images(ref, out, f"stats {binning}")
    imp_err = (out[5].cpu() - ref[5]).abs().sum().item() / max(ref[5].abs().sum().item(), 1e-9)
    assert imp_err < 1e-4, imp_err
    _parity.count_report(out[6], ref[6], f"stats {binning}")
    from oracle.surfel_oracle import preprocess
    with torch.no_grad():
        G = preprocess(*[t.detach() for t in ins], S)
    _parity.radii_report(out[7], G, ins, S, f"stats {binning}")


def test_last_contributor_and_final_T_match_oracle(agslib, binning):
    """The per-pixel state the forward leaves for the backward, as integers: WHICH surfel every pixel blended last
    (n_contrib is a position in the pixel's tile list; the HIP lists are shorter than the oracle's - tiles no pixel can
    reach are not emitted - so positions are turned into surfel ids on both sides) and the transmittance behind it."""
    from active_gs_amd import raster_api as api
    from oracle.surfel_oracle import rasterize
    dev = torch.device("cuda:0")
    mode = {"direct": api.BIN_DIRECT, "tile_sort": api.BIN_TILE_SORT, "radix": api.BIN_RADIX}[binning]
    for (n, h, w, view, mult) in [(5000, 170, 300, 1, 2.0), (3000, 120, 160, 0, 3.0)]:
        a, S = room_case(n, h, w, view=view, seed=view, scale_mult=mult)
        ins = oracle_inputs(a, requires_grad=False)
        with torch.no_grad():
            ref, aux = rasterize(*ins, S, return_aux=True)
        cam = api.Camera(h, w, S.tanfovx, S.tanfovy, S.viewmatrix.to(dev), S.projmatrix.to(dev), S.bg.to(dev))
        g = api.Gaussians(*(a[k].to(dev).contiguous() for k in ("means", "scales", "rotations", "opacities", "colors", "confidences")))
        st = api.alloc_state(n, h, w, 1 << 20, dev, mode)
        api.forward(cam, g, st)
        assert not api.read_status(st)["overflow"]
        ref_last = _parity.oracle_last_contributor(aux, aux["n_contrib"], h, w)
        rep = _parity.last_contributor_report(api.last_contributor(st, n, h, w), ref_last, None, f"{n} {w}x{h} {binning}")
        assert rep["pixels"] == h * w
        fT = api.workspace_region(st, n, h, w, api.REGION_FINAL_T, torch.float32).view(h, w)
        assert torch.equal(1.0 - fT, st.opacity[0])
        assert float((fT.cpu() - aux["final_T"]).abs().max()) < 1e-4


def test_config_flags_center_depth_unnormalized(agslib):
    a, S = room_case(1500, 80, 112, view=5, seed=5, scale_mult=3.0, config=(1, 0, 0, 0, 0))
    ins, ref, gin, out = _run_both(a, S, seed=2)
    _check_images(ref, out)
    _check_grads(ins, gin)


def test_empty_and_culled_inputs(agslib, binning):
    from diff_gaussian_rasterization_2d import GaussianRasterizer
    dev = torch.device("cuda:0")
    a, S = room_case(64, 48, 64, view=0, seed=0)
    # everything behind the camera -> nothing visible, background only
    ins = oracle_inputs(a, requires_grad=False)
    ins[0] = ins[0] * 0 + torch.tensor([100.0, 100.0, 100.0])
    gin = [t.to(dev) for t in ins]
    gin[0].requires_grad_(True)
    out = GaussianRasterizer(product_settings(S, dev))(gin[0], gin[1], gin[2], gin[3], None, gin[4], gin[5], gin[6], None)
    torch.cuda.synchronize()
    assert torch.allclose(out[0].cpu(), S.bg[:3, None, None].expand(3, 48, 64))
    assert out[3].abs().max().item() == 0 and (out[7] == 0).all()
    out[0].sum().backward()
    assert gin[0].grad.abs().max().item() == 0
    # N = 0
    z = lambda *s: torch.zeros(*s, device=dev)
    out = GaussianRasterizer(product_settings(S, dev))(z(0, 3), z(0, 3), z(0, 1), z(0), None, z(0, 3), z(0, 3), z(0, 4), None)
    torch.cuda.synchronize()
    assert out[0].shape == (3, 48, 64) and out[5].shape == (0,)


def test_argument_validation(agslib):
    from diff_gaussian_rasterization_2d import GaussianRasterizer
    dev = torch.device("cuda:0")
    a, S = room_case(16, 32, 32)
    r = GaussianRasterizer(product_settings(S, dev))
    z = lambda *s: torch.zeros(*s, device=dev)
    with pytest.raises(Exception):
        r(z(4, 3), z(4, 3), z(4, 1), z(4), None, None, z(4, 3), z(4, 4), None)
    with pytest.raises(Exception):
        r(z(4, 3), z(4, 3), z(4, 1), z(4), z(4, 1, 3), z(4, 3), z(4, 3), z(4, 4), None)
    with pytest.raises(Exception):
        r(z(4, 3), z(4, 3), z(4, 1), z(4), None, z(4, 3), None, None, None)


def test_adam_matches_torch(agslib):
    from active_gs_amd.optimizer import FusedAdam
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    n = 1000
    shapes = [(n, 3), (n, 3), (n, 4), (n,), (n, 1, 3)]
    lrs = [5e-4, 1e-2, 5e-4, 1e-2, 1e-4]
    ref_p = [torch.randn(*s) for s in shapes]
    dev_p = [p.clone().to(dev) for p in ref_p]
    topt = torch.optim.Adam([{"params": [torch.nn.Parameter(p)], "lr": lr} for p, lr in zip(ref_p, lrs)], eps=1e-15)
    tparams = [g["params"][0] for g in topt.param_groups]
    fopt = FusedAdam(dev_p, lrs, eps=1e-15)
    for step in range(3):
        grads = [torch.randn(*s) for s in shapes]
        grads[0][::2] = 0  # zero-gradient rows still decay m, v and move
        for p, g in zip(tparams, grads):
            p.grad = g.clone()
        topt.step()
        fopt.step([g.to(dev) for g in grads])
    torch.cuda.synchronize()
    for p, q in zip(tparams, dev_p):
        assert torch.allclose(p.detach(), q.cpu(), rtol=1e-5, atol=1e-7)


def test_binning_modes_agree_bitwise_on_order(agslib):
    """Both binning algorithms must produce the same per-tile (depth, id) order: the
    forward images are then bit-identical (same blend order, same arithmetic)."""
    from active_gs_amd import raster_api as api
    dev = torch.device("cuda:0")
    a, S = room_case(20000, 340, 600, view=6, seed=6, scale_mult=2.0)
    cam = api.Camera(S.image_height, S.image_width, S.tanfovx, S.tanfovy, S.viewmatrix.to(dev), S.projmatrix.to(dev),
                     S.bg.to(dev))
    g = api.Gaussians(*(a[k].to(dev).contiguous() for k in ("means", "scales", "rotations", "opacities", "colors",
                                                              "confidences")))
    outs = []
    for mode in (api.BIN_TILE_SORT, api.BIN_RADIX, api.BIN_DIRECT):
        st = api.alloc_state(g.n, cam.image_height, cam.image_width, 1 << 21, dev, mode)
        api.forward(cam, g, st)
        info = api.read_status(st)
        assert not info["overflow"]
        outs.append((st, info))
    # tile-sort / direct mode also drop (surfel, tile) pairs no pixel can reach; radix keeps the D3 rect
    assert 0 < outs[0][1]["num_instances"] <= outs[1][1]["num_instances"]
    assert outs[2][1]["num_instances"] == outs[0][1]["num_instances"] and outs[2][1]["num_visible"] == outs[0][1]["num_visible"]
    tiles = ((cam.image_height + 15) // 16) * ((cam.image_width + 15) // 16)
    assert outs[2][1]["needed"] == tiles * outs[2][1]["max_tile_instances"] and outs[0][1]["needed"] == outs[0][1]["num_instances"]
    # the direct mode's per-tile ranges hold the same id lists as the scan-based layout
    for other in (1, 2):
        for name in ("rgb", "normal", "depth", "opacity", "confidence", "radii"):
            assert torch.equal(getattr(outs[0][0], name), getattr(outs[other][0], name)), (other, name)
    # a second pass on the same workspaces: the direct mode's spread counters were left clean
    for st, info in outs:
        api.forward(cam, g, st)
        again = api.read_status(st)
        assert again["num_instances"] == info["num_instances"] and again["peak_instances"] == info["needed"]
        assert again["overflow_passes"] == 0


def test_workspace_overflow_is_flagged_not_fatal(agslib):
    from active_gs_amd import raster_api as api
    dev = torch.device("cuda:0")
    a, S = room_case(5000, 170, 300, view=1, seed=1, scale_mult=2.0)
    cam = api.Camera(S.image_height, S.image_width, S.tanfovx, S.tanfovy, S.viewmatrix.to(dev), S.projmatrix.to(dev),
                     S.bg.to(dev))
    g = api.Gaussians(*(a[k].to(dev).contiguous() for k in ("means", "scales", "rotations", "opacities", "colors",
                                                              "confidences")))
    for mode in (api.BIN_TILE_SORT, api.BIN_RADIX):
        st = api.alloc_state(g.n, cam.image_height, cam.image_width, 64, dev, mode)  # far too small
        api.forward(cam, g, st)
        info = api.read_status(st)
        assert info["overflow"] and info["num_instances"] > 64 and info["num_sorted"] == 64
        assert info["needed"] == info["num_instances"] == info["peak_instances"] and info["overflow_passes"] == 1
        assert torch.isfinite(st.rgb).all()
    # direct mode: every tile owns max_instances // tiles key slots; ONE over-full tile overflows the pass, the status
    # says what capacity would do, and a workspace of exactly that size renders the reference image
    tiles = ((cam.image_height + 15) // 16) * ((cam.image_width + 15) // 16)
    ref = api.alloc_state(g.n, cam.image_height, cam.image_width, 1 << 20, dev, api.BIN_TILE_SORT)
    api.forward(cam, g, ref)
    st = api.alloc_state(g.n, cam.image_height, cam.image_width, tiles * 4, dev, api.BIN_DIRECT)   # 4 slots per tile
    api.forward(cam, g, st)
    info = api.read_status(st)
    assert info["overflow"] and info["max_tile_instances"] > 4 and info["needed"] == tiles * info["max_tile_instances"]
    assert info["overflow_passes"] == 1 and torch.isfinite(st.rgb).all()
    assert info["num_instances"] == api.read_status(ref)["num_instances"]
    st2 = api.alloc_state(g.n, cam.image_height, cam.image_width, info["needed"], dev, api.BIN_DIRECT)
    api.forward(cam, g, st2)
    info2 = api.read_status(st2)
    assert not info2["overflow"] and info2["overflow_passes"] == 0 and torch.equal(st2.rgb, ref.rgb)
    assert lib_rc_too_small(api, cam, g, dev, tiles)


def lib_rc_too_small(api, cam, g, dev, tiles):
    """fewer key slots than tiles: the direct mode cannot run at all -> AGS_E_WORKSPACE, not a crash"""
    st = api.alloc_state(g.n, cam.image_height, cam.image_width, max(1, tiles - 1), dev, api.BIN_DIRECT)
    try:
        api.forward(cam, g, st)
    except RuntimeError as e:
        return "workspace" in str(e)
    return False


def test_graph_replay_matches_eager_steps(agslib):
    """A captured optimisation step replayed k times == k eager steps (device-side Adam clock)."""
    from active_gs_amd import raster_api as api
    from active_gs_amd.synthetic import make_room_scene
    from active_gs_amd.trainer import SurfelTrainer
    dev = torch.device("cuda:0")
    _, S = room_case(4000, 136, 240, view=2, seed=2)
    cam = api.Camera(S.image_height, S.image_width, S.tanfovx, S.tanfovy, S.viewmatrix.to(dev), S.projmatrix.to(dev),
                     S.bg.to(dev))
    gen = torch.Generator().manual_seed(3)
    d = [(torch.randn(c, 136, 240, generator=gen) / (136 * 240)).to(dev) for c in (3, 3, 1)]
    fn = lambda v, st: (d[0], d[1], d[2], None, None)
    results = []
    for use_graph in (False, True):
        raw = {k: v.to(dev) for k, v in make_room_scene(4000, seed=2).items()}
        raw["scales"][:, :2] += 1.0
        tr = SurfelTrainer(raw)
        tr.step([cam], fn, 1 << 20, device_clock=True)           # step 1 (eager in both arms)
        if use_graph:
            replay = tr.capture([cam], fn, 1 << 20)              # capture performs no work
            for _ in range(3):
                replay()
        else:
            for _ in range(3):
                tr.step([cam], fn, 1 << 20, device_clock=True)
        torch.cuda.synchronize()
        assert int(tr.optim.device_clock[0].item()) == 4
        results.append([p.clone() for p in tr.params])
    for a, b in zip(*results):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-6)


def test_sparse_row_set_equals_dense_training(agslib):
    """AgsRowSet: the per-Gaussian backward and Adam run over the sticky list of surfels the views
    have shown.  Must equal the dense update (untouched rows have g = m = v = 0 -> zero update),
    and the set must be exactly the union of the visible surfels, each listed once."""
    from active_gs_amd import raster_api as api
    from active_gs_amd.synthetic import make_room_scene
    from active_gs_amd.trainer import SurfelTrainer
    dev = torch.device("cuda:0")
    n, h, w = 9000, 136, 240
    cams = []
    for view in range(4):
        _, S = room_case(n, h, w, view=view, seed=5)
        cams.append(api.Camera(S.image_height, S.image_width, S.tanfovx, S.tanfovy, S.viewmatrix.to(dev),
                               S.projmatrix.to(dev), S.bg.to(dev)))
    gen = torch.Generator().manual_seed(6)
    d = [(torch.randn(c, h, w, generator=gen) / (h * w)).to(dev) for c in (3, 3, 1)]
    fn = lambda v, st: (d[0], d[1], d[2], None, None)
    schedule = [[0], [1, 0], [2], [3, 1], [0, 2]]          # the set grows over the first four steps
    out = {}
    for sparse in (False, True):
        raw = {k: v.to(dev) for k, v in make_room_scene(n, seed=5).items()}
        init = [raw[k].clone() for k in ("means", "scales", "rotations", "opacities", "harmonics")]
        tr = SurfelTrainer(raw, sparse_rows=sparse)
        assert (tr.rows is not None) == sparse
        seen = torch.zeros(n, dtype=torch.bool, device=dev)
        for views in schedule:
            tr.step([cams[v] for v in views], fn, 1 << 20, device_clock=True)
            for v in views:   # radii of the last-rendered view only survive; re-render to collect the union
                st = tr.state_for(h, w, 1 << 20)
                api.forward(cams[v], tr.gaussians(), st)
                seen |= st.radii > 0
        torch.cuda.synchronize()
        out[sparse] = dict(params=[p.clone() for p in tr.params], grads=[g.clone() for g in tr.slab.as_list()],
                           m=[x.clone() for x in tr.optim.exp_avg], seen=seen, init=init, tr=tr)
    # the blend backward sums with float atomics, so two runs differ in the last bits whatever the
    # row-set setting; near-zero gradients then flip the sign-like eps=1e-15 Adam update of a few rows
    # (the fused single-rank step keeps no gradient slab - the gradient never leaves the registers, whether the step has
    # one view or several (ags_backward_rows) - so the first moments stand for the gradients here)
    for key in ("m",):
        for a, b in zip(out[False][key], out[True][key]):
            assert float((a - b).abs().sum()) <= 1e-3 * float(a.abs().sum()) + 1e-12, key
    for a, b in zip(out[False]["params"], out[True]["params"]):
        diff = (a - b).abs()
        assert float(diff.mean()) < 2e-6 and float((diff > 1e-4).float().mean()) < 0.01
    tr, seen = out[True]["tr"], out[True]["seen"]
    count = int(tr.rows.count.item())
    listed = tr.rows.rows[:count].long()
    # parameters move between steps, so a surfel can enter/leave the frustum: the set holds at least
    # everything visible at the final parameters that was visible when rendered, and nothing twice
    assert listed.unique().numel() == count
    member = tr.rows.member.bool()
    assert torch.equal(torch.sort(listed).values, torch.nonzero(member).flatten())
    assert 0 < count < n
    untouched = ~member
    for p, p0 in zip(out[True]["params"], out[True]["init"]):
        assert torch.equal(p.reshape(n, -1)[untouched], p0.reshape(n, -1)[untouched])
    for mom in tr.optim.exp_avg + tr.optim.exp_avg_sq:
        assert float(mom.reshape(n, -1)[untouched].abs().max()) == 0.0
    # every surfel seen at the final parameters of a replayed view belongs to the set
    assert int((seen & ~member).sum()) <= int(0.002 * n)


def test_reset_optimizer_restarts_adam_and_row_set(agslib):
    """SurfelTrainer.reset_optimizer() == a fresh trainer on the current parameters (the reference
    re-creates Adam per train() call); a captured graph keeps working across the reset."""
    from active_gs_amd import raster_api as api
    from active_gs_amd.synthetic import make_room_scene
    from active_gs_amd.trainer import SurfelTrainer
    dev = torch.device("cuda:0")
    n, h, w = 5000, 136, 240
    _, S = room_case(n, h, w, view=1, seed=7)
    cam = api.Camera(S.image_height, S.image_width, S.tanfovx, S.tanfovy, S.viewmatrix.to(dev), S.projmatrix.to(dev),
                     S.bg.to(dev))
    gen = torch.Generator().manual_seed(8)
    d = [(torch.randn(c, h, w, generator=gen) / (h * w)).to(dev) for c in (3, 3, 1)]
    fn = lambda v, st: (d[0], d[1], d[2], None, None)
    raw = {k: v.to(dev) for k, v in make_room_scene(n, seed=7).items()}
    a = SurfelTrainer(raw)
    for _ in range(3):
        a.step([cam], fn, 1 << 20, device_clock=True)
    replay = a.capture([cam], fn, 1 << 20)
    a.reset_optimizer()
    assert int(a.rows.count.item()) == 0 and int(a.optim.device_clock[0].item()) == 0
    b = SurfelTrainer({k: v.clone() for k, v in a.raw.items()})       # fresh optimiser, same parameters
    for _ in range(2):
        replay()
        b.step([cam], fn, 1 << 20, device_clock=True)
    torch.cuda.synchronize()
    assert int(a.optim.device_clock[0].item()) == 2 and int(a.rows.count.item()) == int(b.rows.count.item()) > 0
    for x, y in zip(a.params, b.params):
        diff = (x - y).abs()
        assert float(diff.mean()) < 2e-6 and float((diff > 1e-4).float().mean()) < 0.01


def test_fused_activations_match_separate_kernels(agslib):
    """raw_params mode (activations + chain rule inside the per-Gaussian kernels) == ags_activate
    -> forward/backward on activated values -> ags_activate_backward."""
    from active_gs_amd import raster_api as api
    from active_gs_amd.synthetic import make_room_scene
    from active_gs_amd.trainer import SurfelTrainer
    dev = torch.device("cuda:0")
    _, S = room_case(6000, 136, 240, view=3, seed=3)
    cam = api.Camera(S.image_height, S.image_width, S.tanfovx, S.tanfovy, S.viewmatrix.to(dev), S.projmatrix.to(dev),
                     S.bg.to(dev))
    gen = torch.Generator().manual_seed(4)
    d = [(torch.randn(c, 136, 240, generator=gen) / (136 * 240)).to(dev) for c in (3, 3, 1)]
    fn = lambda v, st: (d[0], d[1], d[2], None, None)
    grads, params = [], []
    for fused in (False, True):
        raw = {k: v.to(dev) for k, v in make_room_scene(6000, seed=3).items()}
        raw["scales"][:, :2] += 1.5   # some scales hit the 0.05 clamp
        raw["rotations"] *= 1.7       # un-normalised raw quaternions
        tr = SurfelTrainer(raw, fused_activations=fused)
        tr.step([cam, cam], fn, 1 << 20)
        torch.cuda.synchronize()
        grads.append([g.clone() for g in tr.optim.exp_avg])      # (first step: exp_avg = 0.1 x the gradient; the fused step keeps no slab)
        params.append([p.clone() for p in tr.params])
    for a, b in zip(grads[0], grads[1]):
        assert (a - b).abs().sum() <= 1e-4 * a.abs().sum() + 1e-12
    for a, b in zip(params[0], params[1]):
        assert (a - b).abs().mean() < 1e-6


def _dense_patch_case(n=7000, h=128, w=128):
    """Thousands of large, faint surfels stacked in front of the camera: every central tile
    holds far more than 2048 keys (in-place global bitonic path of the tile sort), footprints
    span > 32 tiles (wave-cooperative emission in the radix path), depths collide often."""
    gen = torch.Generator().manual_seed(77)
    means = torch.stack([(torch.rand(n, generator=gen) - 0.5) * 0.3, (torch.rand(n, generator=gen) - 0.5) * 0.3,
                         0.5 + torch.rand(n, generator=gen) * 0.5], -1)
    means[::50, 2] = 0.75                                    # exact depth ties
    scales = torch.cat([0.02 + 0.03 * torch.rand(n, 2, generator=gen), torch.zeros(n, 1)], 1)
    rots = torch.nn.functional.normalize(torch.tensor([1.0, 0, 0, 0]) + 0.2 * torch.randn(n, 4, generator=gen), dim=-1)
    a = dict(means=means, scales=scales, rotations=rots, opacities=0.01 + 0.03 * torch.rand(n, generator=gen),
             colors=torch.rand(n, 3, generator=gen), confidences=torch.rand(n, generator=gen))
    from oracle.surfel_oracle import OracleSettings
    near, far, t = 0.001, 10.0, 1.0
    P = torch.zeros(4, 4)
    P[0, 0] = 1 / t; P[1, 1] = 1 / t; P[3, 2] = 1; P[2, 2] = far / (far - near); P[2, 3] = -far * near / (far - near)
    S = OracleSettings(h, w, t, t, torch.tensor([0.3, 0.2, 0.1, 0.0]), 1.0, torch.eye(4), (torch.eye(4) @ P.t()).contiguous(),
                       campos=torch.zeros(3), config=torch.tensor([1.0, 1, 1, 0, 0]))
    return a, S


def test_overfull_tiles_and_huge_footprints(agslib):
    from active_gs_amd import raster_api as api
    from oracle.surfel_oracle import rasterize
    dev = torch.device("cuda:0")
    a, S = _dense_patch_case()
    cam = api.Camera(S.image_height, S.image_width, S.tanfovx, S.tanfovy, S.viewmatrix.to(dev), S.projmatrix.to(dev),
                     S.bg.to(dev))
    g = api.Gaussians(*(a[k].to(dev).contiguous() for k in ("means", "scales", "rotations", "opacities", "colors",
                                                              "confidences")))
    gen = torch.Generator().manual_seed(5)
    d = [torch.randn(c, S.image_height, S.image_width, generator=gen).to(dev) for c in (3, 3, 1, 1, 1)]
    res = []
    for mode in (api.BIN_TILE_SORT, api.BIN_RADIX, api.BIN_DIRECT):
        st = api.alloc_state(g.n, S.image_height, S.image_width, 1 << 22, dev, mode)
        api.forward(cam, g, st)
        info = api.read_status(st)
        assert not info["overflow"]
        T = 64
        rg = st.workspace[256 + 8192:256 + 8192 + T * 8].view(torch.int32).view(T, 2)
        # scan-based modes: ranges[tile] = [begin, end); direct mode: ranges[slot] = {tile, count}
        kmax = int(rg[:, 1].max()) if mode == api.BIN_DIRECT else int((rg[:, 1] - rg[:, 0]).max())
        assert kmax > 2048, kmax                         # the test really exercises the big-tile path
        grads = api.backward(cam, g, st, *d)
        torch.cuda.synchronize()
        res.append((st, grads))
    for other in (1, 2):
        for name in ("rgb", "normal", "depth", "opacity", "confidence"):
            assert torch.equal(getattr(res[0][0], name), getattr(res[other][0], name)), (other, name)
        for name in ("means3D", "scales", "rotations", "opacities", "colors"):
            x, y = getattr(res[0][1], name), getattr(res[other][1], name)
            assert (x - y).abs().sum() <= 1e-3 * y.abs().sum()
    ins = [a["means"].clone().requires_grad_(True), torch.zeros(g.n, 3), a["opacities"][:, None].clone().requires_grad_(True),
           a["confidences"], a["colors"].clone().requires_grad_(True), a["scales"].clone().requires_grad_(True),
           a["rotations"].clone().requires_grad_(True)]
    ref = rasterize(*ins, S)
    assert (res[0][0].rgb.cpu() - ref[0].detach()).abs().mean() < RGB_TOL
    assert (res[0][0].opacity.cpu() - ref[3].detach()).abs().mean() < RGB_TOL
    sum((o * x.cpu()).sum() for o, x in zip(ref[:5], d)).backward()
    for name, i in (("means3D", 0), ("opacities", 2), ("colors", 4), ("scales", 5), ("rotations", 6)):
        r = ins[i].grad.reshape(getattr(res[0][1], name).shape)
        rel = (getattr(res[0][1], name).cpu() - r).abs().sum() / r.abs().sum()
        assert rel < 2e-3, (name, float(rel))


def test_module_argument_variants(agslib):
    """shs of degree 0, scale_modifier != 1, non-contiguous / strided inputs, opacities of exactly 0,
    and an image smaller than one tile."""
    from diff_gaussian_rasterization_2d import GaussianRasterizer
    from oracle.surfel_oracle import OracleSettings, rasterize
    dev = torch.device("cuda:0")
    a, S = room_case(900, 10, 13, view=1, seed=9, scale_mult=6.0)          # 13x10 image: a single partial tile
    S = OracleSettings(S.image_height, S.image_width, S.tanfovx, S.tanfovy, S.bg, 1.7, S.viewmatrix, S.projmatrix,
                       campos=S.campos, config=S.config)
    a["opacities"][::7] = 0.0
    ins = oracle_inputs(a)
    ref = rasterize(*ins, S)
    (ref[0].sum() + 2 * ref[2].sum()).backward()
    # strided views of larger buffers
    big = lambda t: torch.cat([t, t], -1).to(dev)[..., : t.shape[-1]]
    gin = [big(ins[0].detach()).requires_grad_(True), torch.zeros(900, 3, device=dev), ins[2].detach().to(dev).requires_grad_(True),
           ins[3].to(dev), big(ins[4].detach()), big(ins[5].detach()).requires_grad_(True), ins[6].detach().to(dev).requires_grad_(True)]
    assert not gin[0].is_contiguous()
    out = GaussianRasterizer(product_settings(S, dev))(gin[0], gin[1], gin[2], gin[3], None, gin[4], gin[5], gin[6], None)
    (out[0].sum() + 2 * out[2].sum()).backward()
    assert (out[0].cpu() - ref[0].detach()).abs().mean() < RGB_TOL
    assert (out[2].cpu() - ref[2].detach()).abs().mean() < 1e-3
    for i in (0, 2, 5, 6):
        r = ins[i].grad
        assert (gin[i].grad.cpu() - r).abs().sum() <= GRAD_TOL * r.abs().sum() + 1e-9, i
    assert torch.all(gin[2].grad.cpu()[::7] == 0)                           # o == 0 never contributes, no NaN
    # SH degree 0: colour = max(C0*sh + 0.5, 0)
    sh = ((ins[4].detach() - 0.5) / 0.28209479177387814)[:, None, :].to(dev)
    out_sh = GaussianRasterizer(product_settings(S, dev))(gin[0].detach(), gin[1], gin[2].detach(), gin[3], sh, None,
                                                          gin[5].detach(), gin[6].detach(), None)
    assert (out_sh[0] - out[0].detach()).abs().max() < 1e-5


@pytest.mark.parametrize("degree", [1, 2, 3])
def test_module_spherical_harmonics(agslib, degree):
    """``shs=`` with sh_degree 1..3: view-dependent colours (direction = mean - campos) through the HIP rasterizer, gradients
    to the coefficients AND to the means (through the direction) - against the oracle rendering the same colours under
    autograd.  The basis itself is pinned to the reference viewer's shader by the CPU suite."""
    from diff_gaussian_rasterization_2d import GaussianRasterizer
    from active_gs_amd.rasterizer import eval_sh
    from oracle.surfel_oracle import rasterize
    import dataclasses
    dev = torch.device("cuda:0")
    n = 3000
    a, S = room_case(n, 96, 128, view=2, seed=11, scale_mult=3.0)
    S = dataclasses.replace(S, sh_degree=degree)
    gen = torch.Generator().manual_seed(degree)
    sh0 = torch.randn(n, 16, 3, generator=gen) * 0.3
    sh0[:, 0] += 0.8

    def colours(means, sh):
        d = means - S.campos.reshape(1, 3)
        d = d / d.norm(dim=1, keepdim=True).clamp_min(1e-20)
        return torch.clamp_min(eval_sh(degree, sh, d) + 0.5, 0.0)

    ins = oracle_inputs(a)
    sh_ref = sh0.clone().requires_grad_(True)
    ins[4] = colours(ins[0], sh_ref)
    ref = rasterize(*ins, S)
    gr = [torch.randn(o.shape, generator=gen) for o in ref[:3]]
    sum((o * g).sum() for o, g in zip(ref[:3], gr)).backward()
    gin = [t.detach().clone().to(dev).requires_grad_(t.requires_grad and t.is_leaf) for t in ins]
    gin[0].requires_grad_(True)
    sh_dev = sh0.clone().to(dev).requires_grad_(True)
    out = GaussianRasterizer(product_settings(S, dev))(means3D=gin[0], means2D=gin[1], opacities=gin[2], confidences=gin[3],
                                                       shs=sh_dev, colors_precomp=None, scales=gin[5], rotations=gin[6],
                                                       cov3D_precomp=None)
    sum((o * g.to(dev)).sum() for o, g in zip(out[:3], gr)).backward()
    torch.cuda.synchronize()
    assert (out[0].cpu() - ref[0].detach()).abs().mean() < RGB_TOL
    for name, got, want in (("shs", sh_dev.grad.cpu(), sh_ref.grad), ("means3D", gin[0].grad.cpu(), ins[0].grad)):
        rel = (got - want).abs().sum().item() / max(want.abs().sum().item(), 1e-12)
        assert rel < GRAD_TOL, f"d_{name}: relative L1 {rel}"
    assert float(sh_ref.grad[:, (degree + 1) ** 2:].abs().sum()) == 0.0 and float(sh_dev.grad[:, (degree + 1) ** 2:].abs().sum()) == 0.0
    assert float(sh_dev.grad[:, 1:(degree + 1) ** 2].abs().sum()) > 0.0      # the higher bands do receive gradient


def test_forward_many_streams_equal_sequential(agslib):
    """Planner-style batch: many small forward-only views on a stream pool == one by one."""
    from active_gs_amd import raster_api as api
    from active_gs_amd.camera import camera_matrices
    from active_gs_amd.synthetic import activate, make_camera, make_room_scene
    dev = torch.device("cuda:0")
    n, h, w, V = 20000, 128, 128, 24
    a = activate(make_room_scene(n, seed=5))
    a["scales"] = a["scales"] * 2
    g = api.Gaussians(*(a[k].to(dev).contiguous() for k in ("means", "scales", "rotations", "opacities", "colors",
                                                              "confidences")))
    cams = []
    for v in range(V):
        c2w, K = make_camera(v, h, w, focal_px=0.5 * 128 / 0.5773503)
        cm = camera_matrices(c2w[None], K[None], 0.001, 10.0)
        cams.append(api.Camera(h, w, cm["tanfov"][0, 0].item(), cm["tanfov"][0, 1].item(), cm["viewmatrix"][0].to(dev),
                               cm["projmatrix"][0].to(dev), torch.zeros(4, device=dev), want_stats=(v % 2 == 0),
                               front_only=(v % 3 == 0)))
    seq = [api.alloc_state(n, h, w, 1 << 19, dev) for _ in range(V)]
    par = [api.alloc_state(n, h, w, 1 << 19, dev) for _ in range(V)]
    api.forward_many(cams, g, seq, None)
    api.forward_many(cams, g, par, api.StreamPool(4))
    torch.cuda.synchronize()
    for s1, s2 in zip(seq, par):
        assert not api.read_status(s2)["overflow"]
        for name in ("rgb", "depth", "normal", "opacity", "confidence", "radii", "count"):
            assert torch.equal(getattr(s1, name), getattr(s2, name)), name
        assert torch.allclose(s1.importance, s2.importance, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("mode", ["batched", "streams"])
def test_view_batch_equals_sequential(agslib, mode):
    """ViewBatch: many small forward-only views in ONE set of launches (blockIdx.y = view), or
    replayed from a hipGraph over a stream pool == the same views rendered one after another;
    poses can change between calls."""
    from active_gs_amd import raster_api as api
    from active_gs_amd.synthetic import activate, make_room_scene
    dev = torch.device("cuda:0")
    n, h, w, V = 20000, 128, 128, 12
    a = activate(make_room_scene(n, seed=9))
    g = api.Gaussians(*(a[k].to(dev).contiguous() for k in ("means", "scales", "rotations", "opacities", "colors",
                                                              "confidences")))
    S = [room_case(16, h, w, view=v, seed=9)[1] for v in range(2 * V)]
    bg = S[0].bg.to(dev)
    batch = api.ViewBatch(g, V, h, w, S[0].tanfovx, S[0].tanfovy, bg, 1 << 18, num_streams=4, want_stats=True,
                          mode=mode)
    for rnd in range(2):                                  # second round: new poses through the same graph
        sel = S[rnd * V:(rnd + 1) * V]
        vm = torch.stack([s.viewmatrix for s in sel]).to(dev)
        pm = torch.stack([s.projmatrix for s in sel]).to(dev)
        states = batch.render(vm, pm)
        torch.cuda.synchronize()
        assert not batch.overflowed()
        got = [(st.rgb.clone(), st.depth.clone(), st.count.clone(), st.radii.clone()) for st in states]
        for v, s in enumerate(sel):
            cam = api.Camera(h, w, s.tanfovx, s.tanfovy, s.viewmatrix.to(dev), s.projmatrix.to(dev), bg, want_stats=True)
            ref = api.alloc_state(n, h, w, 1 << 18, dev)
            api.forward(cam, g, ref)
            torch.cuda.synchronize()
            assert torch.equal(ref.rgb, got[v][0]) and torch.equal(ref.depth, got[v][1])
            assert torch.equal(ref.count, got[v][2]) and torch.equal(ref.radii, got[v][3])
        assert float(got[0][0].abs().sum()) > 0


def test_seen_flags_equal_the_full_count_at_least_one(agslib):
    """want_stats = STATS_SEEN (what the mapper's post-processing renders with): count[i] == 1 exactly where the full
    statistics count >= 1 - single views (with and without a render mask, front_only) and a batch - and the images are
    the same; importance is not needed."""
    from active_gs_amd import raster_api as api
    from active_gs_amd.synthetic import activate, make_room_scene
    dev = torch.device("cuda:0")
    n, h, w, V = 30000, 160, 208, 5
    a = activate(make_room_scene(n, seed=3))
    g = api.Gaussians(*(a[k].to(dev).contiguous() for k in ("means", "scales", "rotations", "opacities", "colors",
                                                              "confidences")))
    S = [room_case(16, h, w, view=v, seed=3)[1] for v in range(V)]
    bg = S[0].bg.to(dev)
    gen = torch.Generator().manual_seed(0)
    masks = (torch.rand(V, h, w, generator=gen) > 0.3).float().to(dev)
    full_counts = []
    for v, s in enumerate(S):
        for front in (False, True):
            kw = dict(front_only=front, render_mask=masks[v].contiguous() if v % 2 else None)
            cam_f = api.Camera(h, w, s.tanfovx, s.tanfovy, s.viewmatrix.to(dev), s.projmatrix.to(dev), bg, want_stats=True, **kw)
            cam_s = api.Camera(h, w, s.tanfovx, s.tanfovy, s.viewmatrix.to(dev), s.projmatrix.to(dev), bg,
                               want_stats=api.STATS_SEEN, **kw)
            sf, ss = api.alloc_state(n, h, w, 1 << 19, dev), api.alloc_state(n, h, w, 1 << 19, dev)
            ss.importance.fill_(7.0)
            api.forward(cam_f, g, sf); api.forward(cam_s, g, ss)
            torch.cuda.synchronize()
            assert not api.read_status(sf)["overflow"]
            assert torch.equal(sf.rgb, ss.rgb) and torch.equal(sf.depth, ss.depth) and torch.equal(sf.radii, ss.radii)
            assert torch.equal(sf.count >= 1, ss.count == 1) and int(ss.count.max()) == 1 and int(ss.count.min()) == 0
            assert float((ss.importance - 7.0).abs().max()) == 0.0            # untouched
            if front:
                full_counts.append(sf.count.clone())
    # the batched launch, every view masked, front_only (the prune pass's shape)
    batch = api.ViewBatch(g, V, h, w, S[0].tanfovx, S[0].tanfovy, bg, 1 << 19, want_stats=api.STATS_SEEN, front_only=True,
                          render_masks=masks)
    ref = api.ViewBatch(g, V, h, w, S[0].tanfovx, S[0].tanfovy, bg, 1 << 19, want_stats=True, front_only=True,
                        render_masks=masks)
    vm = torch.stack([s.viewmatrix for s in S]).to(dev)
    pm = torch.stack([s.projmatrix for s in S]).to(dev)
    batch.render(vm, pm); ref.render(vm, pm)
    torch.cuda.synchronize()
    assert torch.equal(ref.count >= 1, batch.count == 1) and int(batch.count.sum()) > 0
    assert torch.equal(ref.rgb, batch.rgb)


def test_alpha_clamp_near_plane_and_grazing_surfels(agslib):
    """Branches a random room scene rarely reaches: o*G > 0.99 (clamped alpha, zero gradient through
    the clamp), surfels on both sides of the z = 0.2 near cull, edge-on surfels (grazing clamp of the
    depth slope, D5), a coloured background."""
    from diff_gaussian_rasterization_2d import GaussianRasterizer
    from oracle.surfel_oracle import OracleSettings, rasterize
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(31)
    n, h, w = 1200, 96, 96
    means = torch.stack([(torch.rand(n, generator=gen) - 0.5) * 1.6, (torch.rand(n, generator=gen) - 0.5) * 1.6,
                         0.6 + torch.rand(n, generator=gen) * 1.5], -1)
    means[:150, 2] = 0.15 + 0.1 * torch.rand(150, generator=gen)           # around the near cull
    means[:150, :2] *= 0.1
    scales = torch.cat([0.01 + 0.04 * torch.rand(n, 2, generator=gen), torch.zeros(n, 1)], 1)
    q = torch.nn.functional.normalize(torch.tensor([1.0, 0, 0, 0]) + 0.3 * torch.randn(n, 4, generator=gen), dim=-1)
    # edge-on: rotate 90 deg about x (normal ~ +-y), plus a tiny perturbation
    s2 = 0.70710678
    q[200:400] = torch.nn.functional.normalize(torch.tensor([s2, s2, 0, 0]) + 0.01 * torch.randn(200, 4, generator=gen), dim=-1)
    opac = torch.ones(n)
    opac[600:] = 0.3 + 0.7 * torch.rand(n - 600, generator=gen)
    a = dict(means=means, scales=scales, rotations=q, opacities=opac, colors=torch.rand(n, 3, generator=gen),
             confidences=torch.rand(n, generator=gen))
    near, far, t = 0.001, 10.0, 0.8
    P = torch.zeros(4, 4)
    P[0, 0] = 1 / t; P[1, 1] = 1 / t; P[3, 2] = 1; P[2, 2] = far / (far - near); P[2, 3] = -far * near / (far - near)
    S = OracleSettings(h, w, t, t, torch.tensor([0.9, 0.5, 0.1, 0.0]), 1.0, torch.eye(4), (torch.eye(4) @ P.t()).contiguous(),
                       campos=torch.zeros(3), config=torch.tensor([1.0, 1, 1, 0, 0]))
    ins = oracle_inputs(a)
    ref = rasterize(*ins, S)
    vis = ref[7] > 0
    assert 0 < int(vis[:150].sum()) < 150                                   # some culled by z <= 0.2, some not
    gen2 = torch.Generator().manual_seed(2)
    gr = [torch.randn(o.shape, generator=gen2) for o in ref[:5]]
    sum((o * g).sum() for o, g in zip(ref[:5], gr)).backward()
    gin = [t_.detach().clone().to(dev).requires_grad_(t_.requires_grad) for t_ in ins]
    out = GaussianRasterizer(product_settings(S, dev))(gin[0], gin[1], gin[2], gin[3], None, gin[4], gin[5], gin[6], None)
    sum((o * g.to(dev)).sum() for o, g in zip(out[:5], gr)).backward()
    torch.cuda.synchronize()
    _check_images(ref, out)
    assert (out[7].cpu() != ref[7]).float().mean().item() < 2e-3
    _check_grads(ins, gin)
    # the clamp really was active: many pixels carry alpha == 0.99 from an o = 1 surfel
    assert float(out[3].detach().max()) > 0.98


@pytest.mark.parametrize("use_rows", [False, True])
def test_batched_backward_equals_sum_of_per_view_backwards(agslib, use_rows):
    """ags_backward_batch (views summed atomically into one slab; dense or row-set form of the
    per-Gaussian stage) == per-view ags_backward accumulated one after another."""
    from active_gs_amd import raster_api as api
    from active_gs_amd.synthetic import activate, make_room_scene
    dev = torch.device("cuda:0")
    n, h, w, V = 12000, 136, 240, 5
    a = activate(make_room_scene(n, seed=12))
    g = api.Gaussians(*(a[k].to(dev).contiguous() for k in ("means", "scales", "rotations", "opacities", "colors",
                                                              "confidences")))
    S = [room_case(16, h, w, view=v, seed=12)[1] for v in range(V)]
    bg = S[0].bg.to(dev)
    gen = torch.Generator().manual_seed(13)
    d_rgb = (torch.randn(V, 3, h, w, generator=gen) / (h * w)).to(dev)
    d_nrm = (torch.randn(V, 3, h, w, generator=gen) / (h * w)).to(dev)
    d_dep = (torch.randn(V, 1, h, w, generator=gen) / (h * w)).to(dev)
    # reference: one view after another
    ref = api.alloc_grads(n, dev, zero=True)
    for v, s in enumerate(S):
        cam = api.Camera(h, w, s.tanfovx, s.tanfovy, s.viewmatrix.to(dev), s.projmatrix.to(dev), bg)
        st = api.alloc_state(n, h, w, 1 << 19, dev)
        api.forward(cam, g, st)
        api.backward(cam, g, st, d_rgb[v], d_nrm[v], d_dep[v], None, None, grads=ref, accumulate=True)
    batch = api.ViewBatch(g, V, h, w, S[0].tanfovx, S[0].tanfovy, bg, 1 << 19)
    batch.viewmats.copy_(torch.stack([s.viewmatrix for s in S]).to(dev))
    batch.projmats.copy_(torch.stack([s.projmatrix for s in S]).to(dev))
    rows = api.RowSet(n, dev) if use_rows else None
    batch.forward(V, touched=rows)
    out = api.alloc_grads(n, dev, zero=True)
    batch.backward(V, d_rgb, d_nrm, d_dep, out, touched=rows)
    torch.cuda.synchronize()
    assert not batch.overflowed()
    for name in ("means3D", "scales", "rotations", "opacities", "colors"):
        x, y = getattr(out, name), getattr(ref, name)
        assert float(y.abs().sum()) > 0
        assert float((x - y).abs().sum()) <= 1e-4 * float(y.abs().sum()), name
    if use_rows:
        assert int(rows.count.item()) == int((batch.radii > 0).any(0).sum())


@pytest.mark.parametrize("fused_activations", [True, False])
def test_one_backward_rows_launch_equals_per_view_backwards(agslib, fused_activations):
    """Several views per optimisation step: ags_backward_rows (every view's blend backward leaves its gradient records
    in its own workspace, ONE launch over the member rows sums the views' chain rules in registers and applies the
    optimiser step) == one per-Gaussian backward per view accumulating into the gradient slab.  Three different views,
    three steps (the row set grows), parameters, moments and - without the fused step - the gradient slab compared."""
    from active_gs_amd import raster_api as api
    from active_gs_amd.synthetic import make_room_scene
    from active_gs_amd.trainer import SurfelTrainer
    dev = torch.device("cuda:0")
    h, w, n = 136, 240, 6000
    cams = []
    for v in (0, 3, 5):
        _, S = room_case(n, h, w, view=v, seed=3)
        cams.append(api.Camera(h, w, S.tanfovx, S.tanfovy, S.viewmatrix.to(dev), S.projmatrix.to(dev), S.bg.to(dev)))
    gen = torch.Generator().manual_seed(4)
    d = [[(torch.randn(c, h, w, generator=gen) / (h * w * 3)).to(dev) for c in (3, 3, 1)] for _ in cams]
    fn = lambda v, st: (d[v][0], d[v][1], d[v][2], None, None)
    res = []
    for multi in (True, False):
        raw = {k: v.to(dev) for k, v in make_room_scene(n, seed=3).items()}
        raw["scales"][:, :2] += 1.0
        tr = SurfelTrainer(raw, fused_activations=fused_activations)
        tr.MULTI_VIEW_ROWS = multi
        for _ in range(3):
            tr.step(cams, fn, 1 << 20)
        tr.check_overflow()
        torch.cuda.synchronize()
        res.append(([p.clone() for p in tr.params], [m.clone() for m in tr.optim.exp_avg], [g.clone() for g in tr.slab.as_list()],
                    int(tr.rows.count.item())))
    (pa, ma, ga, ca), (pb, mb, gb, cb) = res
    assert ca == cb > 500
    # (the blend backward sums with float atomics: two runs differ in the last bits, and a near-zero gradient then flips
    # the sign-like eps = 1e-15 Adam update of a few entries - bulk and outlier fraction, as in the sparse-vs-dense test)
    for a, b in zip(pa, pb):
        diff = (a - b).abs()
        assert float(diff.mean()) < 2e-6 and float((diff > 1e-4).float().mean()) < 0.01, (float(diff.mean()), float(diff.max()))
    for a, b in zip(ma, mb):
        assert float((a - b).abs().sum()) <= 1e-3 * float(b.abs().sum()) + 1e-12
    if not fused_activations:          # (the fused single-rank step keeps no gradient slab)
        for a, b in zip(ga, gb):
            assert float((a - b).abs().sum()) <= 1e-3 * float(b.abs().sum()) + 1e-12



def test_views_of_a_step_on_several_streams_equal_one_stream(agslib):
    """SurfelTrainer.VIEW_STREAMS: the views of a multi-view step are enqueued on several streams (nothing a view's
    four launches read is produced by another view of the step; the row set's members are claimed with atomicExch) and
    joined in front of the one per-Gaussian backward - same parameters and moments as on one stream, stepped eagerly
    and replayed from a captured graph (whose capture forks and joins the streams)."""
    from active_gs_amd import raster_api as api
    from active_gs_amd.synthetic import make_room_scene
    from active_gs_amd.trainer import SurfelTrainer
    dev = torch.device("cuda:0")
    h, w, n = 136, 240, 6000
    cams = []
    for v in (0, 3, 5, 6, 2):
        _, S = room_case(n, h, w, view=v, seed=3)
        cams.append(api.Camera(h, w, S.tanfovx, S.tanfovy, S.viewmatrix.to(dev), S.projmatrix.to(dev), S.bg.to(dev)))
    gen = torch.Generator().manual_seed(4)
    d = [[(torch.randn(c, h, w, generator=gen) / (h * w * len(cams))).to(dev) for c in (3, 3, 1)] for _ in cams]
    fn = lambda v, st: (d[v][0], d[v][1], d[v][2], None, None)
    res = []
    for streams, graph in ((1, False), (4, False), (3, True)):
        raw = {k: v.to(dev) for k, v in make_room_scene(n, seed=3).items()}
        raw["scales"][:, :2] += 1.0
        tr = SurfelTrainer(raw)
        tr.VIEW_STREAMS = streams
        tr.step(cams, fn, 1 << 20)
        if graph:
            replay = tr.capture(cams, fn, 1 << 20)
            for _ in range(3):
                replay()
        else:
            for _ in range(3):
                tr.step(cams, fn, 1 << 20)
        tr.check_overflow()
        torch.cuda.synchronize()
        assert (len(tr._lanes) == 0) == (streams == 1)
        res.append(([p.clone() for p in tr.params], [m.clone() for m in tr.optim.exp_avg], int(tr.rows.count.item())))
    for (pa, ma, ca) in res[1:]:
        assert ca == res[0][2] > 500
        for a, b in zip(pa, res[0][0]):
            diff = (a - b).abs()
            assert float(diff.mean()) < 2e-6 and float((diff > 1e-4).float().mean()) < 0.01, (float(diff.mean()), float(diff.max()))
        for a, b in zip(ma, res[0][1]):
            assert float((a - b).abs().sum()) <= 1e-3 * float(b.abs().sum()) + 1e-12


_SWEEP = [  # (n, h, w, view, mult, config, masked): odd image sizes (ragged last tiles), one-tile images, sparse and dense
    (60, 16, 16, 0, 8.0, (1, 1, 1, 0, 0), False), (25, 17, 33, 1, 6.0, (1, 1, 1, 1, 0), False),
    (40, 31, 250, 2, 5.0, (1, 0, 1, 0, 0), False), (300, 49, 47, 3, 4.0, (1, 1, 0, 1, 1), True),
    (900, 130, 70, 4, 3.0, (1, 1, 1, 1, 1), True), (1500, 97, 161, 5, 2.5, (1, 0, 0, 1, 0), False),
    (2500, 75, 203, 6, 2.0, (1, 1, 1, 0, 1), False), (4000, 144, 176, 7, 1.5, (1, 1, 1, 1, 0), True),
    (6000, 33, 400, 8, 3.5, (1, 0, 1, 1, 1), False), (350, 200, 24, 9, 6.0, (1, 1, 0, 0, 0), False),
]


@pytest.mark.parametrize("case", range(len(_SWEEP)))
def test_sweep_of_sizes_flags_and_masks_matches_oracle(agslib, case):
    """A seeded sweep over what the other cases fix: image sizes that are not tile multiples (down to one tile), one to
    a few thousand surfels, every combination class of the reference's config flags (operations.py:697-699), with and
    without a render mask - images, statistics, radii and all six gradients against the oracle with the same gates."""
    n, h, w, view, mult, config, masked = _SWEEP[case]
    mask = None
    if masked:
        gen = torch.Generator().manual_seed(100 + case)
        mask = (torch.rand(1, h, w, generator=gen) > 0.4).float()
    a, S = room_case(n, h, w, view=view, seed=50 + case, scale_mult=mult, config=config, mask=mask,
                     bg=(0.05 * case, 0.3, 1.0 - 0.08 * case, 0.0))
    ins, ref, gin, out = _run_both(a, S, seed=case)
    what = f"sweep {case}: {n} surfels {w}x{h} config {config} mask {masked}"
    _check_images(ref, out, what)
    from oracle.surfel_oracle import preprocess
    with torch.no_grad():
        G = preprocess(*[t.detach() for t in ins], S)
    _parity.radii_report(out[7], G, ins, S, what)
    if config[3]:
        ref_imp = float(ref[5].abs().sum())
        assert float((out[5].cpu() - ref[5]).abs().sum()) <= 1e-4 * ref_imp + 1e-7, what
        _parity.count_report(out[6], ref[6], what)
    else:
        assert int(out[6].abs().sum()) == 0 and float(out[5].abs().sum()) == 0.0
    if any(float(t.grad.abs().sum()) > 0 for t in ins if t.grad is not None):
        _check_grads(ins, gin, what)


@pytest.mark.parametrize("raw_params", [False, True])
def test_batched_forward_shares_row_loads_across_views_bit_for_bit(agslib, raw_params):
    """``ags_k_preprocess_views`` (a batch's per-Gaussian stage with the rows loaded and activated once per GROUP of views,
    ``AgsTuning.view_group``) against one view per workgroup (``view_group = 1``): the same images, radii, counts, statistics
    and status words whatever the group size (2, 5, all views in one group; 0 = the default, which is off), for activated and
    for raw parameters, a map size that is no multiple of the workgroup, and the same row set (as a set)."""
    from active_gs_amd import _lib, raster_api as api
    from active_gs_amd.synthetic import activate, make_room_scene
    dev = torch.device("cuda:0")
    n, h, w, V = 70_001, 136, 240, 7
    raw = {k: v.to(dev) for k, v in make_room_scene(n, seed=12).items()}
    raw["scales"][:, :2] += 0.8
    if raw_params:
        a = activate(raw)
        g = api.Gaussians(raw["means"], raw["scales"], raw["rotations"], raw["opacities"], raw["harmonics"].view(n, 3).contiguous(),
                          a["confidences"].contiguous(), raw_params=True, scale_factor=0.01, max_scale=0.05)
    else:
        a = activate(raw)
        g = api.Gaussians(*(a[k].contiguous() for k in ("means", "scales", "rotations", "opacities", "colors", "confidences")))
    S = [room_case(16, h, w, view=v, seed=12)[1] for v in range(V)]
    bg = S[0].bg.to(dev)
    vm = torch.stack([s.viewmatrix for s in S]).to(dev)
    pm = torch.stack([s.projmatrix for s in S]).to(dev)
    out = {}
    for group in (1, 2, 5, 64, 0):
        batch = api.ViewBatch(g, V, h, w, S[0].tanfovx, S[0].tanfovy, bg, 1 << 21, want_stats=True,
                              tuning=_lib.make_tuning(view_group=group))
        batch.viewmats.copy_(vm); batch.projmats.copy_(pm)
        rows = api.RowSet(n, dev)
        batch.forward(V, touched=rows)
        torch.cuda.synchronize()
        st = batch.statuses(V)
        assert int(st[:, 2].max()) == 0
        k = int(rows.count.item())
        out[group] = dict(rgb=batch.rgb.clone(), normal=batch.normal.clone(), depth=batch.depth.clone(), opacity=batch.opacity.clone(),
                          confidence=batch.confidence.clone(), radii=batch.radii.clone(), count=batch.count.clone(),
                          importance=batch.importance.clone(), status=st[:, [0, 3, 6]].clone(),
                          rows=torch.sort(rows.rows[:k]).values.clone(), member=rows.member.clone())
    ref = out[1]
    assert float(ref["opacity"].max()) > 0.5 and int(ref["status"][:, 1].min()) > 100 and ref["rows"].numel() > 1000
    for group, o in out.items():
        for key in ("rgb", "normal", "depth", "opacity", "confidence", "radii", "count", "status", "rows", "member"):
            assert torch.equal(ref[key], o[key]), (group, key)
        assert torch.allclose(ref["importance"], o["importance"], rtol=1e-5, atol=1e-6), group      # (float atomics: order)
