"""Software-pipelined optimisation step (``ags_backward_fused_next`` + ``ags_forward_resume``): the per-Gaussian launch
of step k (chain rule + Adam over the row set) also runs the per-Gaussian stage of step k + 1.  Must give what
``ags_backward`` + ``ags_forward`` give - the reference's loop is optimizer.step() then the next render
(/root/reference/mapping/gaussian_map.py:77-127)."""
import pytest
import torch

from _scenes import room_case

pytestmark = pytest.mark.gpu


def _setup(n=9000, h=136, w=240, views=4, seed=5):
    from active_gs_amd import raster_api as api
    dev = torch.device("cuda:0")
    cams = []
    for view in range(views):
        _, S = room_case(n, h, w, view=view, seed=seed)
        cams.append(api.Camera(S.image_height, S.image_width, S.tanfovx, S.tanfovy, S.viewmatrix.to(dev),
                               S.projmatrix.to(dev), S.bg.to(dev)))
    gen = torch.Generator().manual_seed(6)
    d = [(torch.randn(c, h, w, generator=gen) / (h * w)).to(dev) for c in (3, 3, 1)]
    return dev, cams, (lambda v, st: (d[0], d[1], d[2], None, None))


def _trainer(n, dev, seed=5, scale_boost=0.0):
    from active_gs_amd.synthetic import make_room_scene
    from active_gs_amd.trainer import SurfelTrainer
    raw = {k: v.to(dev) for k, v in make_room_scene(n, seed=seed).items()}
    raw["scales"][:, :2] += scale_boost
    return SurfelTrainer(raw)


@pytest.mark.parametrize("n,h,w,boost,hint", [(9000, 136, 240, 0.0, 0), (3000, 96, 128, 1.2, 0), (40000, 340, 600, 0.0, 0),
                                              (9000, 136, 240, 0.0, 1), (40000, 340, 600, 0.0, 40)])
def test_prepared_pass_is_bitwise_the_plain_forward(agslib, n, h, w, boost, hint):
    """After a pipelined step the prepared view, resumed, must be bit for bit the plain forward of the same parameters
    (same records, same keys up to their order inside a tile, which the sort removes), with the same radii, status
    and row set."""
    from active_gs_amd import raster_api as api
    dev, cams, fn = _setup(n, h, w)
    tr = _trainer(n, dev, scale_boost=boost)
    cap = 1 << 21
    tr.step([cams[0]], fn, cap)                              # plain step: fills the row set with view 0
    tr._rows_hint = hint       # 0: sized for the whole map; tiny: the member workgroups stride over the list in many passes
    tr.step([cams[1]], fn, cap, next_cam=cams[2])            # pipelined: view 2 is prepared
    assert tr._prepared is not None
    st = tr.state_for(h, w, cap)
    g = tr.gaussians()
    count_before = int(tr.rows.count.item())
    api.forward(cams[2], g, st, touched=tr.rows, resume=True)
    torch.cuda.synchronize()
    got = dict(rgb=st.rgb.clone(), normal=st.normal.clone(), depth=st.depth.clone(), opacity=st.opacity.clone(),
               confidence=st.confidence.clone(), radii=st.radii.clone())
    info_a = api.read_status(st)
    api.init_workspace(st, n, h, w)
    api.forward(cams[2], g, st, touched=tr.rows)
    torch.cuda.synchronize()
    info_b = api.read_status(st)
    for k in ("rgb", "normal", "depth", "opacity", "confidence", "radii"):
        assert torch.equal(got[k], getattr(st, k)), k
    for k in ("num_instances", "num_visible", "max_tile_instances", "overflow"):
        assert info_a[k] == info_b[k], (k, info_a, info_b)
    assert info_a["num_visible"] > 0 and not info_a["overflow"]
    # the plain forward found nothing new to insert; the list holds every member once
    count = int(tr.rows.count.item())
    assert count == count_before
    listed = tr.rows.rows[:count].long()
    assert listed.unique().numel() == count
    assert torch.equal(torch.sort(listed).values, torch.nonzero(tr.rows.member.bool()).flatten())
    assert bool((tr.rows.member.bool() | (st.radii <= 0)).all())      # every visible surfel is a member


def test_pipelined_training_matches_plain_training(agslib):
    from active_gs_amd import raster_api as api
    n, h, w = 9000, 136, 240
    dev, cams, fn = _setup(n, h, w)
    schedule = [0, 1, 2, 3, 0, 2, 1, 1, 3]
    out = {}
    for mode in ("plain", "pipelined"):
        tr = _trainer(n, dev)
        for k, v in enumerate(schedule):
            nxt = cams[schedule[k + 1]] if (mode == "pipelined" and k + 1 < len(schedule)) else None
            tr.step([cams[v]], fn, 1 << 20, next_cam=nxt)
        torch.cuda.synchronize()
        assert tr._prepared is None
        count = int(tr.rows.count.item())
        out[mode] = dict(params=[p.clone() for p in tr.params], members=torch.sort(tr.rows.rows[:count]).values.clone(),
                         m=[x.clone() for x in tr.optim.exp_avg], steps=int(tr.optim.device_clock[0].item()))
    assert out["plain"]["steps"] == out["pipelined"]["steps"] == len(schedule)
    assert torch.equal(out["plain"]["members"], out["pipelined"]["members"])
    # (float atomics in the blend backward: two runs differ in the last bits whatever the launch structure)
    for a, b in zip(out["plain"]["m"], out["pipelined"]["m"]):
        assert float((a - b).abs().sum()) <= 1e-3 * float(a.abs().sum()) + 1e-12
    for a, b in zip(out["plain"]["params"], out["pipelined"]["params"]):
        diff = (a - b).abs()
        assert float(diff.mean()) < 2e-6 and float((diff > 1e-4).float().mean()) < 0.01


def test_pipelined_graph_replay_and_leaving_the_pipeline(agslib):
    from active_gs_amd import raster_api as api
    n, h, w = 9000, 136, 240
    dev, cams, fn = _setup(n, h, w)
    cap = 1 << 20
    out = {}
    for mode in ("plain", "pipelined"):
        tr = _trainer(n, dev)
        tr.step([cams[0]], fn, cap)
        replay = tr.capture([cams[0]], fn, cap, repeat=3, pipeline=(mode == "pipelined"))
        assert replay.pipelined == (mode == "pipelined") and replay.steps == 3
        replay(); replay()
        tr.step([cams[1]], fn, cap)                  # an un-pipelined step: drops the prepared pass
        assert tr._prepared is None
        torch.cuda.synchronize()
        assert not api.read_status(tr.state_for(h, w, cap))["overflow"]
        if mode == "pipelined":
            with pytest.raises(RuntimeError):
                replay()                             # the graph starts at a tile sort: nothing is prepared any more
        out[mode] = dict(params=[p.clone() for p in tr.params], steps=int(tr.optim.device_clock[0].item()))
    # pipelined capture takes one eager priming step
    assert out["pipelined"]["steps"] == out["plain"]["steps"] + 1 == 1 + 1 + 6 + 1
    # one more step of the same view moves the parameters a little: compare against a plain run with that step too
    tr = _trainer(n, dev)
    for v in [0, 0, 0, 0, 0, 0, 0, 0, 1]:
        tr.step([cams[v]], fn, cap)
    torch.cuda.synchronize()
    for a, b in zip(tr.params, out["pipelined"]["params"]):
        diff = (a - b).abs()
        assert float(diff.mean()) < 2e-6 and float((diff > 1e-4).float().mean()) < 0.01


def test_fused_next_argument_checks(agslib):
    """ags_backward_fused_next refuses what it cannot do: no fused Adam, another binning mode, a foreign row set."""
    from active_gs_amd import raster_api as api
    from active_gs_amd import _lib
    n, h, w = 3000, 96, 128
    dev, cams, fn = _setup(n, h, w)
    tr = _trainer(n, dev)
    st = tr.state_for(h, w, 1 << 18)
    g = tr.gaussians()
    api.forward(cams[0], g, st, touched=tr.rows)
    d = fn(0, st)
    with pytest.raises(RuntimeError):       # no fused optimiser step
        api.backward(cams[0], g, st, *d, grads=tr.slab.grads, touched=tr.rows, next_view=(cams[1], st))
    other = api.RowSet(n, dev)
    fused = (tr.optim.tensors_struct(tr.slab.as_list()), tr.optim.eps)
    with pytest.raises(RuntimeError):       # the next pass must insert into the optimiser's row set
        lib = _lib.load()
        cs, gs = cams[0].c_struct(), g.c_struct()
        im, pg, ws = st.images_struct(), st.per_gaussian_struct(), st.ws_struct()
        import ctypes as C
        dout = _lib.AgsImageGrads(_lib.ptr(d[0]), _lib.ptr(d[1]), _lib.ptr(d[2]), None, None)
        din = _lib.AgsGaussianGrads(None, None, None, None, None, None, 0)
        clock, lrs, b1, b2 = tr.optim.tick_args()
        din.adam_clock = _lib.ptr(clock)
        din.touched = tr.rows.c_struct()
        din.fused_adam = C.cast(C.pointer(fused[0]), C.c_void_p)
        cs2, pg2, ws2 = cams[1].c_struct(), st.per_gaussian_struct(other), st.ws_struct()
        _lib.check(lib.ags_backward_fused_next(C.byref(cs), C.byref(gs), C.byref(im), C.byref(pg), C.byref(dout), C.byref(din),
                                               C.byref(ws), C.byref(cs2), C.byref(pg2), C.byref(ws2), 0,
                                               torch.cuda.current_stream().cuda_stream), "ags_backward_fused_next")
    torch.cuda.synchronize()


def test_prepared_pass_in_another_workspace(agslib):
    """The next view may be of another size: its per-Gaussian stage then goes into that size's own workspace
    (``next_ws`` != ``ws``) - resumed, it is bit for bit the plain forward."""
    from active_gs_amd import raster_api as api
    n = 6000
    dev, cams_a, fn = _setup(n, 136, 240)
    _, cams_b, _ = _setup(n, 96, 128)
    tr = _trainer(n, dev)
    cap = 1 << 20
    tr.step([cams_a[0]], fn, cap)
    tr.step([cams_a[1]], fn, cap, next_cam=cams_b[2])
    st_b = tr.state_for(96, 128, cap)
    assert tr._prepared is not None and tr._prepared[0] is st_b and st_b is not tr.state_for(136, 240, cap)
    g = tr.gaussians()
    api.forward(cams_b[2], g, st_b, touched=tr.rows, resume=True)
    torch.cuda.synchronize()
    got = {k: getattr(st_b, k).clone() for k in ("rgb", "normal", "depth", "opacity", "confidence", "radii")}
    api.init_workspace(st_b, n, 96, 128)
    api.forward(cams_b[2], g, st_b, touched=tr.rows)
    torch.cuda.synchronize()
    for k, v in got.items():
        assert torch.equal(v, getattr(st_b, k)), k
    assert int((st_b.radii > 0).sum()) > 0


def test_pipelined_step_reports_an_outgrown_workspace(agslib):
    """A prepared pass whose tile lists do not fit the workspace is flagged like any other pass (sticky status words):
    ``check_overflow`` raises instead of training on truncated lists."""
    n, h, w = 9000, 136, 240
    dev, cams, fn = _setup(n, h, w)
    tr = _trainer(n, dev, scale_boost=1.5)
    tiles = ((h + 15) // 16) * ((w + 15) // 16)
    cap = 4 * tiles                 # four key slots per tile: far too few
    tr.step([cams[0]], fn, cap, next_cam=cams[1])
    tr.step([cams[1]], fn, cap, next_cam=cams[1])
    with pytest.raises(RuntimeError, match="outgrew"):
        tr.check_overflow()


def test_every_binning_mode_replays_from_a_graph(agslib):
    """A forward + backward recorded into a hipGraph gives the eager pass's results on EVERY replay, in every binning
    mode: the workspace's counters are left clean by the kernels themselves (nothing a replay could find stale).  (Found
    while building: a hipMemsetAsync NODE wrote garbage patterns from the second replay on - ROCm 7.0.)"""
    from active_gs_amd import raster_api as api
    dev = torch.device("cuda:0")
    a, S = room_case(6000, 136, 240, view=2, seed=12, scale_mult=2.0)
    cam = api.Camera(S.image_height, S.image_width, S.tanfovx, S.tanfovy, S.viewmatrix.to(dev), S.projmatrix.to(dev),
                     S.bg.to(dev))
    g = api.Gaussians(*(a[k].to(dev).contiguous() for k in ("means", "scales", "rotations", "opacities", "colors",
                                                              "confidences")))
    gen = torch.Generator().manual_seed(3)
    d = [(torch.randn(c, cam.image_height, cam.image_width, generator=gen) / (cam.image_height * cam.image_width)).to(dev)
         for c in (3, 3, 1)]
    for mode in (api.BIN_TILE_SORT, api.BIN_RADIX, api.BIN_DIRECT):
        st = api.alloc_state(g.n, cam.image_height, cam.image_width, 1 << 20, dev, mode)
        api.forward(cam, g, st)
        ref_grads = api.backward(cam, g, st, d[0], d[1], d[2])
        ref = [t.clone() for t in (st.rgb, st.depth, st.radii, ref_grads.means3D, ref_grads.scales, ref_grads.colors)]
        info = api.read_status(st)
        assert not info["overflow"] and info["num_instances"] > 0
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
                api.forward(cam, g, st)
                grads = api.backward(cam, g, st, d[0], d[1], d[2])
        torch.cuda.synchronize()
        for replay in range(4):
            graph.replay()
            torch.cuda.synchronize()
            got = (st.rgb, st.depth, st.radii, grads.means3D, grads.scales, grads.colors)
            for k in range(3):
                assert torch.equal(got[k], ref[k]), (mode, replay, k)
            for k in range(3, 6):          # (float atomics: order-dependent last bits)
                assert float((got[k] - ref[k]).abs().sum()) <= 1e-5 * float(ref[k].abs().sum()) + 1e-12, (mode, replay, k)
            again = api.read_status(st)
            assert again["num_instances"] == info["num_instances"] and again["overflow_passes"] == 0


def test_more_views_than_one_rows_launch_joins(agslib):
    """A single-rank step over more views than ``ags_backward_rows`` joins in one launch (AGS_MAX_ROW_VIEWS = 16: e.g. all
    32 views of BASELINE.json's configuration 4 on one GPU) goes through its views in groups - the earlier groups sum into
    the gradient rows, the last group adds what they left and runs the fused Adam step.  Here: 4 views in groups of
    16 (one launch), 3 (two launches) and 1 (four): the same parameters and moments up to the order of the float sums."""
    from active_gs_amd.trainer import SurfelTrainer
    n, h, w = 9000, 136, 240
    dev, cams, fn = _setup(n, h, w)
    res = {}
    was = SurfelTrainer.MAX_ROW_VIEWS
    try:
        for groups in (16, 3, 1):
            SurfelTrainer.MAX_ROW_VIEWS = groups
            tr = _trainer(n, dev)
            init = [p.clone() for p in tr.params]
            for _ in range(3):
                tr.step(cams, fn, 1 << 21)
            tr.check_overflow()
            torch.cuda.synchronize()
            res[groups] = ([p.clone() for p in tr.params], [m.clone() for m in tr.optim.exp_avg], init, int(tr.rows.count.item()))
    finally:
        SurfelTrainer.MAX_ROW_VIEWS = was
    ref_p, ref_m, init, rows = res[16]
    assert rows > 1000
    for groups in (3, 1):
        p, m, _, r = res[groups]
        assert r == rows
        for a, b, i0 in zip(p, ref_p, init):
            travel = (b - i0).abs().mean()
            assert (a - b).abs().mean() <= 2e-3 * travel + 1e-9, groups
        for a, b in zip(m, ref_m):
            assert (a - b).abs().sum() <= 1e-4 * b.abs().sum() + 1e-12, groups


def test_trainer_asks_for_the_cull_first_kernel_when_its_views_show_little(agslib):
    """``SurfelTrainer._adapt_kernels``: every CHECK_EVERY steps the trainer reads its views' status blocks anyway; while they
    show less than a tenth of the map (here: 80 k rows in random order, a view of ~7 %) it asks for the cull-first
    per-Gaussian kernel (``AgsTuning.cull_first_min_n`` = 1; the library's own threshold is 2^20 rows) - bit-identical
    records, so training lands where the plain kernel's does; a caller's explicit selection is left alone."""
    import os
    from active_gs_amd import _lib
    from active_gs_amd.synthetic import make_room_scene
    from active_gs_amd.trainer import SurfelTrainer
    if os.environ.get("AGS_PRE_CULL_MIN_N") is not None:
        pytest.skip("the process pins the kernel choice")
    n, h, w = 80_000, 340, 600
    dev, cams, fn = _setup(n, h, w, views=1)
    res = {}
    for mode in ("adaptive", "pinned"):
        raw = {k: v.to(dev) for k, v in make_room_scene(n, seed=5).items()}
        tr = SurfelTrainer(raw, tuning=_lib.make_tuning(cull_first_min_n=-1) if mode == "pinned" else None)
        for _ in range(SurfelTrainer.CHECK_EVERY + 4):
            tr.step(cams, fn, 1 << 21)
        tr.check_overflow()
        torch.cuda.synchronize()
        res[mode] = (tr, [p.clone() for p in tr.params])
    ta, tp = res["adaptive"][0], res["pinned"][0]
    assert ta.tuning is not None and ta.tuning.cull_first_min_n == 1 and tp.tuning.cull_first_min_n == -1
    assert ta.cull_first_kernel() and not tp.cull_first_kernel()
    # hysteresis: on below 8 % of the rows, off above 12 %, unchanged in between - a trainer at the threshold does not flip
    for frac, want in ((0.09, 1), (0.11, 1), (0.13, 0), (0.11, 0), (0.09, 0), (0.07, 1)):
        ta._adapt_kernels(frac * n)
        assert ta.tuning.cull_first_min_n == want, (frac, want)
        tp._adapt_kernels(frac * n)
        assert tp.tuning.cull_first_min_n == -1          # a caller's explicit selection is never adapted
    assert 0 < int(ta.rows.count.item()) == int(tp.rows.count.item()) < 0.1 * n
    init = make_room_scene(n, seed=5)
    for a, b, key in zip(res["adaptive"][1], res["pinned"][1], ("means", "scales", "rotations", "opacities", "harmonics")):
        travel = float((b - init[key].to(dev).reshape(b.shape)).abs().mean())
        assert float((a - b).abs().mean()) <= 5e-3 * travel + 1e-9, key      # (the blend backward's float atomics are unordered)
