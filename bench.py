#!/usr/bin/env python3
"""bench.py — splatted-Gaussians/s (fwd+bwd) @1200x680 on MI355X (BASELINE.json metric).

One step = one optimisation step of the hot path on resident data: activations ->
forward rasterization -> backward rasterization (with fixed synthetic image gradients) ->
activation backward -> [exchange of the gradient rows when N>1] -> fused Adam.
Workload at every N: BASELINE.json configs[1] per GPU (office0 stand-in, 200k surfels,
1200x680, one view per rank, weak scaling).  Prints ONE JSON line on rank 0.

Timing: after W warm-up steps the K steps are timed R times (R >= 21, >= 0.25 s in total), every sample bracketed
by barrier + torch.cuda.synchronize() on both sides and the MAX taken over the ranks; `ms_per_step` / `value` are the
MEDIAN sample, min / max are printed beside it.  Everything else in the line (stage times, the un-pipelined and the
bf16-split forms of the same step, the drop-in module's call path, the larger configurations, the CPU oracle) is
measured AFTER that region and labelled.

``python bench.py --gpus N`` with N > 1 and no RANK in the environment launches the N ranks itself
(``python -m torch.distributed.run``, one process per GPU, RCCL) BEFORE this process touches the GPU and
relays rank 0's line; under ``torch.distributed.run`` (RANK set) it is one of the ranks.
``--check`` (with --gpus N): only initialise the process group, run one all-gather and the captured-collective probe
and print which exchange path every rank would take - a readiness check that costs seconds.
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

N_GAUSS = 200_000
H, W = 680, 1200
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
PMC_HBM_FILE = "pmc_hbm_bytes.json"   # per-stage HBM bytes per launch of this command (rocprofv3 --pmc, committed)
PMC_SQ_FILE = "sq_counters.json"      # per-stage SQ instruction counters per launch of this command
PMC5_HBM_FILE = "pmc5_hbm_bytes.json"  # the same HBM bytes for configuration 5's steady-state launches (c5_counters.sh)
DTYPE = "f32"   # every stage computes in fp32, like the reference's extension (the blend backward's per-surfel sums run on
                # exact f32 matrix instructions; the bf16 hi/lo-split form is an opt-in, timed as ms_per_step_bf16_split)


def stage_bytes(N, V, I, P, T, rows=None, direct=True, fused_single_view=True, bwd_extra_images=0):
    """ALGORITHMIC bytes per launch of each stage: every logical
    array moved once.  ``rows``: member rows of the sticky row set when the per-Gaussian backward
    runs in its row-set form with the Adam step fused in (single rank), else None.  ``direct``: one-pass
    binning (the per-Gaussian kernel writes the keys, the sort stage copies them into slot order).
    ``bwd_extra_images``: how many of d_opacity / d_confidence the backward is given (0 in the bench step, whose
    image gradients are d_rgb, d_normal, d_depth - like the reference's loss, gaussian_map.py:106-124)."""
    if rows is not None:
        # per member row: id 4 + radius 4 + parameters read 56 (the four geometry tensors 44 + harmonics 12) + gradient
        # record read 64 / re-zeroed 64 (visible rows) + interleaved Adam moments 112 read + 112 written + parameters
        # written 56 (+ the gradient row 56 when a slab is kept: several views per step)
        pbwd = (8 + 56 + 224 + 56 + (0 if fused_single_view else 56)) * rows + 128 * V
    else:
        pbwd = 44 * N + 128 * V + 68 * N                       # means/scales/rot/radii; dgeom read + re-zero; 5 grads
    if direct:
        pre = 60 * N + 4 * N + 128 * V + 8 * I                 # inputs; radii; geom 64 + zeroed dgeom 64; keys written at once
        binning = 4 * T + 16 * I + 8 * T                       # tile counts; keys read + written in slot order; slot headers
    else:
        pre = 60 * N + 8 * N + 136 * V                         # inputs; radii+tiles; geom 64 + rect 8 + zeroed dgeom 64
        binning = 8 * N + 12 * I + 24 * I + 8 * I + 8 * T      # tile counts/scan; key write; sort r+w once; ranges
    return {
        "preprocess": pre,
        "binning": binning,
        "render_fwd": 8 * T + 68 * I + 44 * P,                  # headers; id 4 + record 64; 9 ch + T + n_contrib
        # headers; id 4 + record 64; per pixel n_contrib 4 + final_T 4 + depth 4 + opacity 4 + d_rgb 12 + d_normal 12 +
        # d_depth 4 = 44 (+ 4 per extra image gradient); one 64-byte gradient record per visible surfel
        "render_bwd": 8 * T + 68 * I + (44 + 4 * bwd_extra_images) * P + 64 * V,
        "preprocess_bwd": pbwd,
    }


STAGE_IDS = {"preprocess": 0, "binning": 1, "render_fwd": 2, "render_bwd": 3, "preprocess_bwd": 4}


def time_samples(run_k, samples, dist_on, dev):
    """`samples` timings of one call of run_k(), each bracketed by barrier + synchronize on both sides; -> list of
    seconds, the MAX over the ranks per sample."""
    out = []
    for _ in range(samples):
        if dist_on:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_k()
        torch.cuda.synchronize()
        if dist_on:
            torch.distributed.barrier()
        out.append(time.perf_counter() - t0)
    if dist_on:
        from active_gs_amd.dist_util import all_reduce_
        t = torch.tensor(out, device=dev, dtype=torch.float64)
        all_reduce_(t, torch.distributed.ReduceOp.MAX)
        out = t.tolist()
    return out


def summarise(samples_s, k):
    per = sorted(s / k * 1e3 for s in samples_s)
    return dict(median=statistics.median(per), min=per[0], max=per[-1], samples=len(per))


def read_stage_times(lib):
    import ctypes as C
    from active_gs_amd import _lib
    med, mean = {}, {}
    for name, sid in STAGE_IDS.items():
        ms, md, cnt = C.c_float(), C.c_float(), C.c_int32()
        _lib.check(lib.ags_profile_read(sid, C.byref(ms), C.byref(md), C.byref(cnt)), "ags_profile_read")
        med[name], mean[name] = md.value, ms.value      # median over the eager steps (robust to host stalls)
    return med, mean


def measure_dropin(dev, n, h, w, views, iters=30, focal=None, deferred=False, samples=9):
    """The path an UNMODIFIED caller takes (operations.py:682-713, :854): ``GaussianRasterizer(settings)(...)`` per view
    under autograd + one backward through all of them - the module alone (no facade post-processing, no loss head,
    no optimiser).  ``deferred`` False: the module's default - every call reads its status block back and repairs an
    outgrown workspace before it returns (what the CUDA extension's num_rendered read-back does; safe for a caller that
    has no retry).  True: the opt-in for loops that settle once per iteration (``check_overflow()`` after the backward,
    inside the timed loop).  Sampled like the headline: `samples` timings of `iters` iterations each, median + min / max
    (one unsampled loop once read 0.49 ms on the driver's box against 0.09-0.13 in seven other sessions - a host stall
    nothing recorded).  -> dict(ms per view: median / min / max, host enqueue time, module waits per view)."""
    from active_gs_amd.camera import camera_matrices
    from active_gs_amd.synthetic import activate, make_camera, make_room_scene
    import active_gs_amd.rasterizer as R
    from diff_gaussian_rasterization_2d import GaussianRasterizationSettings, GaussianRasterizer, check_overflow
    raw = make_room_scene(n, seed=0)
    c2w, K = zip(*[make_camera(v, h, w, focal_px=focal) for v in range(views)])
    cm0 = camera_matrices(torch.stack(c2w), torch.stack(K), 0.001, 10.0)
    a = {k: v.to(dev) for k, v in activate(raw).items()}
    leaves = [a["means"].clone().requires_grad_(True), torch.zeros(n, 3, device=dev, requires_grad=True),
              a["opacities"][:, None].clone().requires_grad_(True), a["confidences"], a["colors"].clone().requires_grad_(True),
              a["scales"].clone().requires_grad_(True), a["rotations"].clone().requires_grad_(True)]
    gen = torch.Generator().manual_seed(0)
    gimg = [torch.randn(c, h, w, generator=gen).to(dev) / (h * w) for c in (3, 3, 1)]
    settings = [GaussianRasterizationSettings(
        image_height=h, image_width=w, tanfovx=float(cm0["tanfov"][v, 0]), tanfovy=float(cm0["tanfov"][v, 1]),
        bg=torch.zeros(4, device=dev), scale_modifier=1.0, viewmatrix=cm0["viewmatrix"][v].to(dev),
        projmatrix=cm0["projmatrix"][v].to(dev), sh_degree=0, campos=cm0["campos"][v].to(dev), prefiltered=False,
        render_mask=torch.tensor([], device=dev), weight_thres=0.03, debug=False,
        config=torch.tensor([1.0, 1, 1, 0, 0]).to(dev)) for v in range(views)]

    def iteration():
        outs = [GaussianRasterizer(s)(leaves[0], leaves[1], leaves[2], leaves[3], None, leaves[4], leaves[5], leaves[6], None)
                for s in settings]
        # rgb, normal, depth carry gradients (gaussian_map.py:106-124); opacity / confidence are used detached
        torch.autograd.backward([o[k] for o in outs for k in range(3)], [gimg[k] for _ in outs for k in range(3)])
        for t in leaves:
            t.grad = None

    was = R.get_option("always_check")
    R.set_option("always_check", 0.0 if deferred else 1.0)
    try:
        for _ in range(3):
            iteration()
        torch.cuda.synchronize()
        check_overflow()
        syncs0, early0 = R.counters()["status_syncs"], R.counters().get("early_waits", 0)
        per, host = [], []
        for _ in range(samples):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                iteration()
                if deferred:
                    check_overflow()      # one wait per iteration for the views' status copies (they have long landed)
            host_s = time.perf_counter() - t0
            torch.cuda.synchronize()
            per.append((time.perf_counter() - t0) / (iters * views) * 1e3)
            host.append(host_s / (iters * views) * 1e3)
        check_overflow()
    finally:
        R.set_option("always_check", was)
    calls = samples * iters * views
    return dict(ms_per_view=statistics.median(per), ms_per_view_min=min(per), ms_per_view_max=max(per), samples=samples,
                iterations_per_sample=iters, host_enqueue_ms_per_view=statistics.median(host),
                module_syncs_per_view=(R.counters()["status_syncs"] - syncs0) / calls,
                module_event_waits_per_view=(R.counters().get("early_waits", 0) - early0) / calls,
                workspace_checks="deferred, settled once per iteration (opt-in)" if deferred else
                                 "every call, before it returns (default): the call waits for the event behind its per-Gaussian "
                                 "kernel (the pass's overflow is known there), not for the stream")


def measure_config(tag, n, h, w, views, room, steps, dev, lrs=None, binning_mode=None):
    """One GPU's optimisation step on a synthetic scene of another size (BASELINE.json configs 4 and 5: the per-GPU
    share), same step as the headline: hipGraph replay timed over samples, then the per-stage eager pass."""
    from active_gs_amd import _lib, raster_api as api
    from active_gs_amd.camera import camera_matrices
    from active_gs_amd.synthetic import make_camera, make_room_scene
    from active_gs_amd.trainer import SurfelTrainer
    lib = _lib.load()
    raw = {k: v.to(dev) for k, v in make_room_scene(n, room=room, seed=0).items()}
    # (the views of a step on four streams: this bench's image_grads hands out fixed tensors - nothing shared, nothing allocated)
    trainer = SurfelTrainer(raw, lrs=lrs, binning_mode=api.BIN_DIRECT if binning_mode is None else binning_mode,
                            view_streams=int(os.environ.get("AGS_VIEW_STREAMS", "4")))
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
    V = I = 0
    while True:                                     # size the workspace: grow until no view overflows
        trainer.step(cams, fn, cap, device_clock=True)
        need, V, I = 0, 0, 0
        for cam in cams:
            st = trainer.state_for(h, w, cap)
            api.forward(cam, trainer.gaussians(), st)
            info = api.read_status(st)
            need = max(need, info["needed"]); V += info["num_visible"]; I += info["num_instances"]
        if need <= cap:
            break
        cap = int(need * 1.25)
    replay = trainer.capture(cams, fn, cap)
    for _ in range(5):
        replay()
    torch.cuda.synchronize()
    s = summarise(time_samples(lambda: [replay() for _ in range(steps)], 9, False, dev), steps)
    # per-stage times: the same step with its views on ONE stream (on several, the stages of different views overlap
    # and an event pair around one of them also times its neighbours)
    view_streams = min(int(trainer.VIEW_STREAMS), views) if views > 1 else 1
    trainer.VIEW_STREAMS = 1
    _lib.check(lib.ags_profile_enable(steps * views), "ags_profile_enable")
    for _ in range(steps):
        trainer.step(cams, fn, cap)
    torch.cuda.synchronize()
    med, _ = read_stage_times(lib)
    lib.ags_profile_enable(0)
    trainer.check_overflow()
    P, T = h * w, ((h + 15) // 16) * ((w + 15) // 16)
    rows = int(trainer.rows.count.item()) if trainer.rows is not None else None
    # per view: the stage times are medians over all (step, view) launches
    sb = stage_bytes(n, V / views, I / views, P, T, None if rows is None else rows / 1.0, direct=True, fused_single_view=True)
    if views > 1 and rows is not None:
        # several views per step: ONE per-Gaussian backward launch for all of them (ags_backward_rows): every member row
        # once (id, parameters, moments in and out) + every view's gradient records of the rows it shows (read + re-zeroed)
        sb["preprocess_bwd"] = (8 + 56 + 224 + 56) * rows + 128 * V
    frac = {k: (sb[k] / (med[k] * 1e-3) / 1e9 / HBM_PEAK_GBS if med[k] > 0 else 0.0) for k in sb}
    st = trainer.state_for(h, w, cap)
    out = dict(config=tag, surfels=n, image=[h, w], views_per_step=views, ms_per_step=round(s["median"], 4),
               ms_per_step_min=round(s["min"], 4), ms_per_step_max=round(s["max"], 4), samples=s["samples"], steps_per_sample=steps,
               gaussians_per_s=n * views / (s["median"] * 1e-3), visible_per_view=V // views, tile_instances_per_view=I // views,
               member_rows=rows, workspace_MB=round(st.workspace.numel() / 2**20, 1),
               stage_ms_per_view={k: round(v, 4) for k, v in med.items()},
               view_streams=view_streams,
               stage_note=("per view, except preprocess_bwd: ONE launch per step for all views (ags_backward_rows); measured with the views "
                           "on one stream - the timed step spreads them over view_streams streams, so the stages do not add up to it"
                           if views > 1 and rows is not None else "per view"),
               stage_hbm_frac={k: round(v, 4) for k, v in frac.items()},
               stage_algorithmic_bytes={k: int(v) for k, v in sb.items()},
               whole_step_hbm_frac=round((sum(v for k, v in sb.items() if k != "preprocess_bwd") * views +
                                          sb["preprocess_bwd"] * (1 if (views > 1 and rows is not None) else views))
                                         / (s["median"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4))
    del trainer, replay, raw
    torch.cuda.empty_cache()
    return out


STRONG_CONFIGS = {
    # BASELINE.json configs[3]: "room0, 1.5M Gaussians, 32 training views sharded view-parallel over 8 x MI355X, RCCL grad
    # all-reduce" and configs[4]: "5M-Gaussian scene, 2048x2048 ..., 1->8 GPU roofline sweep" (8 views, so that 8 ranks have
    # one each): the TOTAL work is fixed, rank r renders views r::N - "scaling": "strong" inside these objects
    "c4": dict(n=1_500_000, views=32, h=680, w=1200, room="room0",
               what="config 4: room0 stand-in, 1.5 M surfels, 32 views @1200x680 per optimisation step, 32/N per rank"),
    "c5": dict(n=5_000_000, views=8, h=2048, w=2048, room="office0",
               what="config 5: 5 M surfels, 8 views @2048x2048 per optimisation step, 8/N per rank"),
}


def strong_configs():
    """STRONG_CONFIGS, or the sizes a test asks for: AGS_BENCH_STRONG="c4=150000,8,680,1200;c5=400000,4,1024,1024"."""
    out = {k: dict(v) for k, v in STRONG_CONFIGS.items()}
    spec = os.environ.get("AGS_BENCH_STRONG")
    if spec:
        for part in spec.split(";"):
            k, v = part.split("=")
            n, views, h, w = (int(x) for x in v.split(","))
            out[k].update(n=n, views=views, h=h, w=w, reduced=True)
    return out


def replica_checksums(trainer, dist_on):
    """Are the replicas bit-identical?  Per parameter tensor the int64 sum of its bits, MIN and MAX over the ranks."""
    sums = torch.stack([t.detach().view(torch.int32).to(torch.int64).sum() for t in trainer.params])
    if not dist_on:
        return True, [int(x) for x in sums.tolist()]
    from active_gs_amd.dist_util import all_reduce_
    lo, hi = sums.clone(), sums.clone()
    all_reduce_(lo, torch.distributed.ReduceOp.MIN)
    all_reduce_(hi, torch.distributed.ReduceOp.MAX)
    return bool(torch.equal(lo, hi)), [int(x) for x in sums.tolist()]


def note(msg):
    """progress on stderr with AGS_BENCH_VERBOSE=1 (where a run is when a rank faults or hangs)"""
    if os.environ.get("AGS_BENCH_VERBOSE") == "1":
        print(f"[bench rank {os.environ.get('RANK', '0')}] {msg}", file=sys.stderr, flush=True)


def measure_strong(key, cfg, steps, dev, world, rank, dist_on, check_only=False, samples=7, probe_steps=6):
    """One optimisation step of a STRONG-scaled configuration on `world` ranks: the configuration's views are dealt out
    round-robin (rank r renders views r::world), every rank holds the whole map and the Adam state, ONE exchange of the
    gradients per step (the path `RowExchange.agree()` picks from the ranks' row coverage: all-gather of member rows, or
    the dense slab all-reduced in DENSE_CHUNKS row chunks on a communication stream under the next chunk's chain rule,
    /root/reference/mapping/gaussian_map.py:113-125 is the sum it shards), then the replicated Adam step.  Timed like
    the headline (samples of `steps` steps, barrier + synchronize on both sides, max over ranks, median); the exposed
    part of the exchange comes from HIP events on the main / communication streams of `probe_steps` eager steps."""
    from active_gs_amd import raster_api as api
    from active_gs_amd.camera import camera_matrices
    from active_gs_amd.dist_util import all_reduce_
    from active_gs_amd.synthetic import make_camera, make_room_scene
    from active_gs_amd.trainer import SurfelTrainer
    n, total, h, w = cfg["n"], cfg["views"], cfg["h"], cfg["w"]
    note(f"strong {key}: {n} surfels, {total} views")
    mine = list(range(rank, total, world))
    raw = {k: v.to(dev) for k, v in make_room_scene(n, room=cfg["room"], seed=0).items()}
    trainer = SurfelTrainer(raw, view_streams=int(os.environ.get("AGS_VIEW_STREAMS", "4")))
    cams = []
    for v in mine:
        c2w, K = make_camera(v, h, w)
        cm = camera_matrices(c2w[None].to(dev), K[None].to(dev), 0.001, 10.0)
        tan = cm["tanfov"][0].cpu()
        cams.append(api.Camera(h, w, float(tan[0]), float(tan[1]), cm["viewmatrix"][0].contiguous(),
                               cm["projmatrix"][0].contiguous(), torch.zeros(4, device=dev)))
    gen = torch.Generator().manual_seed(4)
    d_img = [(torch.randn(c, h, w, generator=gen) / (h * w * total)).to(dev) for c in (3, 3, 1)]
    fn = lambda v, st: (d_img[0], d_img[1], d_img[2], None, None)
    # size the workspaces from forward-only probes of this rank's views; every rank takes the largest need (one agreement)
    g = trainer.gaussians()
    cap, V, I = 1 << 22, 0, 0
    while True:
        probe = api.alloc_state(n, h, w, cap, dev)
        need, V, I = 0, 0, 0
        for cam in cams:
            api.forward(cam, g, probe)
            info = api.read_status(probe)
            need, V, I = max(need, info["needed"]), V + info["num_visible"], I + info["num_instances"]
        del probe
        t = torch.tensor([need], device=dev, dtype=torch.int64)
        if dist_on:
            all_reduce_(t, torch.distributed.ReduceOp.MAX)
        need = int(t.item())
        if need <= cap:
            break
        cap = int(need * 1.25) + 4096
    del g
    note(f"strong {key}: sized, cap {cap}")
    for _ in range(2):
        trainer.step(cams, fn, cap)              # the first step of a data-parallel trainer agrees on the exchange
    trainer.check_overflow()
    x = trainer.exchange
    slab_bytes = 4 * trainer.slab.flat.numel()
    if not dist_on:
        path, xbytes = "single rank: no exchange, row-set Adam fused into the per-Gaussian backward", 0
    elif x is not None and x.capacity:
        path, xbytes = f"rows: all-gather of {x.capacity}-row segments (64 B per row)", 4 * x.send.numel()
    else:
        K = max(1, min(int(trainer.DENSE_CHUNKS), n // trainer.DENSE_CHUNK_MIN_ROWS))
        path = (f"dense: all-reduce of the {slab_bytes / 1e6:.0f} MB gradient slab in {K} row chunks on a communication stream, "
                "chunk k under chunk k+1's chain rule, Adam per chunk" if trainer._dense_chunked(cams)
                else f"dense: one all-reduce of the {slab_bytes / 1e6:.0f} MB gradient slab")
        xbytes = slab_bytes
    out = dict(config=key, what=cfg["what"], scaling="strong", surfels=n, image=[h, w], views_total=total,
               views_this_rank=len(mine), world_size=world, exchange_path=path, exchange_bytes_per_rank=xbytes,
               dense_chunks=(max(1, min(int(trainer.DENSE_CHUNKS), n // trainer.DENSE_CHUNK_MIN_ROWS))
                             if dist_on and trainer.rows is None else None),
               member_rows_this_rank=None if trainer.rows is None else int(trainer.rows.count.item()),
               visible_per_view=V // max(1, len(mine)), tile_instances_per_view=I // max(1, len(mine)),
               view_streams=min(int(trainer.VIEW_STREAMS), max(1, len(mine))), reduced_size=bool(cfg.get("reduced")))
    if dist_on:                                     # every rank must have taken the same path
        paths = [None] * world
        torch.distributed.all_gather_object(paths, path)
        if len(set(paths)) != 1:
            raise RuntimeError(f"ranks disagree on the exchange path of {key}: {paths}")
    if not check_only:
        note(f"strong {key}: {path}; capturing")
        replay = trainer.capture(cams, fn, cap)
        for _ in range(3):
            replay()
        torch.cuda.synchronize()
        note(f"strong {key}: replays ok; timing")
        sm = summarise(time_samples(lambda: [replay() for _ in range(steps)], samples, dist_on, dev), steps)
        out.update(ms_per_step=round(sm["median"], 4), ms_per_step_min=round(sm["min"], 4), ms_per_step_max=round(sm["max"], 4),
                   samples=sm["samples"], steps_per_sample=steps, gaussians_per_s=n * total / (sm["median"] * 1e-3),
                   launch=("hipGraph replay" + (", exchange recorded in the graph" if getattr(replay, "collective_in_graph", False)
                                                else (": graph | collective | graph" if dist_on and trainer.rows is not None
                                                      else (", eager chunks (host-driven collectives)" if dist_on else "")))))
        if dist_on:
            # where the exchange sits in the step: events of a few EAGER steps (a replayed graph cannot be timed inside)
            note(f"strong {key}: timed; probing the exchange")
            trainer.tail_probe, trainer.exchange_probe = [], []
            for _ in range(probe_steps):
                trainer.step(cams, fn, cap)
            torch.cuda.synchronize()
            if trainer.tail_probe:
                tl = [SurfelTrainer.tail_timeline(r) for r in trainer.tail_probe]
                med = lambda k: statistics.median(t[k] for t in tl)
                out["exchange_timeline_ms"] = dict(chain_rule=round(med("rows_ms"), 4), all_reduce_sum=round(med("all_reduce_sum_ms"), 4),
                                                   adam=round(med("adam_ms"), 4), tail=round(med("tail_ms"), 4),
                                                   all_reduce_per_chunk=[round(statistics.median(t["all_reduce_ms"][c] for t in tl), 4)
                                                                         for c in range(len(tl[0]["all_reduce_ms"]))])
                out["all_reduce_exposed_ms"] = round(med("exposed_ms"), 4)
            elif trainer.exchange_probe:
                ex = [a.elapsed_time(b) for a, b in trainer.exchange_probe]
                out["all_reduce_exposed_ms"] = round(statistics.median(ex), 4)
                out["exchange_timeline_ms"] = dict(note="the all-gather sits on the main stream between the per-Gaussian backward and "
                                                        "the gathered Adam: all of it is exposed")
            trainer.tail_probe = trainer.exchange_probe = None
            out["exposed_note"] = ("HIP events on the main and communication streams of eager steps: what the main stream waited "
                                   "between the last chunk's chain rule and the last Adam update beyond the Adam updates themselves")
        trainer.check_overflow()
    note(f"strong {key}: checksums")
    ok, sums = replica_checksums(trainer, dist_on)
    out["replicas_identical"] = ok
    out["parameter_checksums"] = sums
    del trainer, raw, cams, d_img
    torch.cuda.empty_cache()
    return out


def measure_c3(dev):
    """BASELINE.json configs[2]: "full mapper loop (densify/prune + Adam), 1 MI355X, 500 iters" - 50 keyframes x 10
    iterations from an empty map through ``active_gs_amd.gaussian_map.GaussianMap.update`` (the class an untouched
    mapping.Mapper constructs and calls), frames rendered from the room stand-in.  The fraction of the loop during which
    kernels run comes from a committed rocprofv3 kernel trace of examples/mapper_loop.py (it cannot be collected from
    inside this process) and is labelled as such; the share of the wall time the GPU-bound phase covers is measured here."""
    from active_gs_amd.synthetic import make_keyframes, run_mapper_loop
    res = {}

    def first_pass(tag, h, w):
        frames = make_keyframes(50, h, w, dev)
        r = run_mapper_loop(frames, steps=10, draw="device", warmup_frames=2)
        res[tag] = dict(seconds=r["seconds"], ms_per_iteration=r["ms_per_iteration"], final_surfels=r["final_surfels"],
                        iterations=r["iterations"], mean_frame_error=r["mean_frame_error"], overflow_retries=r["overflow_retries"],
                        device_mallocs=r["device_mallocs"])
        return frames

    frames = first_pass("512x512", 512, 512)
    # Where the loop's wall time goes, from ONE run (no figure of a profiled run is divided by another run's clock): the
    # same 512x512 loop ONCE MORE in this process - right behind the first pass, so the caching allocator holds every block
    # the loop needs (device_mallocs: what a long-running process sees from its second mission on) - with a HIP event + the
    # host clock at every phase boundary.  The iterations are the GPU-bound phase (their GPU time is kernel time, the host
    # enqueues them in a fifth of it); growth, a call's set-up and post-processing are host-bound (GPU-timeline time ~ host
    # time: launches with the GPU waiting between them).
    busy = None
    try:
        r = run_mapper_loop(frames, steps=10, draw="device", warmup_frames=0, phases=True)
        busy = dict(seconds_with_marks=r["seconds"], gpu_bound_frac=r["gpu_bound_frac"], iterations_gpu_ms=r["iterations_gpu_ms"],
                    phases=r["phases"], device_mallocs=r["device_mallocs"],
                    what="this run, second pass of the 512x512 loop (allocator warm) with event marks at the phase boundaries (~7 records "
                         "per keyframe): gpu_bound_frac = GPU-timeline time of the iterations / that pass's wall time - a LOWER bound "
                         "of the kernels-busy fraction (the host-bound phases also run kernels)")
        try:
            kb = json.load(open(os.path.join(ROOT, "profiles", "c3_kernels_busy.json")))
            busy["kernels_busy_frac_under_profiler"] = kb.get("kernels_busy_frac")
            busy["under_profiler_source"] = (f"profiles/c3_kernels_busy.json (rocprofv3 --kernel-trace of examples/mapper_loop.py, session "
                                             f"{kb.get('_session', '?')}): union of the kernel intervals / that profiled run's own wall time")
        except Exception:
            pass
    except Exception as e:
        busy = f"{type(e).__name__}: {e}"
        torch.cuda.synchronize()
    del frames
    torch.cuda.empty_cache()
    frames = first_pass("1200x680", 680, 1200)
    del frames
    torch.cuda.empty_cache()
    return dict(seconds=res["512x512"]["seconds"], ms_per_iteration=res["512x512"]["ms_per_iteration"],
                final_surfels=res["512x512"]["final_surfels"], kernels_busy=busy, **res,
                what="GaussianMap(cfg, device).update(dataframe) x 50 keyframes (10 iterations each, batch 8 with 3 active frames, "
                     "prune every 5th keyframe) from an empty map, after two warm-up keyframes on a scratch map; the error-weighted "
                     "frame draw on the device (the same distribution as the reference's np.random.choice); the dataframe carries "
                     "the pose on the host next to the device copies (extrinsic_host / intrinsic_host / depth_range_host - what "
                     "mapper.py:94 holds before line 95; INTEGRATION.md section 3), images on the device",
                pose="host")


def pmc_entry(e):
    """(traffic, read, write, how) of one stage's entry of a committed counter file.  New files (profiles/pmc_exact_summary.py)
    carry the memory-side bytes by REQUEST SIZE; files of earlier rounds carry FETCH_SIZE, which counts every read request at
    64 bytes (profiles/r06_fetch_calibration.md): exact for isolated 64-byte gathers, half the bytes of whole 128-byte lines."""
    if "read" in e:
        return e.get("traffic"), e.get("read"), e.get("write"), (f"write = 64 x TCC_EA0_WRREQ_64B + 32 x the other write requests (exact); read between "
                                                                 f"{e.get('read_lo')} (64 B per read request = FETCH_SIZE) and {e.get('read_hi')} (128 B per "
                                                                 "request), the end this stage's access pattern sits at: profiles/r06_fetch_calibration.md")
    if "fetch_raw" in e:
        return e.get("traffic"), 2 * e["fetch_raw"], e.get("write"), ("read = 2 x FETCH_SIZE: an UPPER bound (FETCH_SIZE counts every "
                                                                      "request at 64 bytes; gathers move 64, streams 128)")
    return None, None, None, "no entry for this stage"


def flatten_scalars(out) -> None:
    """The driver's record keeps the SCALAR keys of `config`; the other workloads of the line live in nested objects
    (config.c3, config.dropin, config.secondary...).  Their headline figures are repeated here as top-level scalars of
    `config` so that they are in the record - the nested objects stay, these are copies."""
    cfg = out.get("config", {})

    def dig(obj, *path):
        for k in path:
            if not isinstance(obj, dict) or k not in obj:
                return None
            obj = obj[k]
        return obj if isinstance(obj, (int, float)) and not isinstance(obj, bool) else None

    flat = {"c3_seconds_512": dig(cfg, "c3", "512x512", "seconds"), "c3_seconds_1200x680": dig(cfg, "c3", "1200x680", "seconds"),
            "c3_ms_per_iteration_512": dig(cfg, "c3", "512x512", "ms_per_iteration"),
            "c3_final_surfels_512": dig(cfg, "c3", "512x512", "final_surfels"),
            "c3_device_mallocs_512": dig(cfg, "c3", "512x512", "device_mallocs"),
            "c3_gpu_bound_frac": dig(cfg, "c3", "kernels_busy", "gpu_bound_frac"),
            "c3_seconds_512_second_pass": dig(cfg, "c3", "kernels_busy", "seconds_with_marks"),
            "c3_device_mallocs_512_second_pass": dig(cfg, "c3", "kernels_busy", "device_mallocs"),
            "c4_share_ms": dig(cfg, "secondary", "c4_share_ms"), "c5_ms": dig(cfg, "secondary", "c5_ms"),
            "c4_strong_ms": dig(cfg, "secondary", "strong", "c4", "ms_per_step"),
            "c5_strong_ms": dig(cfg, "secondary", "strong", "c5", "ms_per_step"),
            "c4_strong_gaussians_per_s": dig(cfg, "secondary", "strong", "c4", "gaussians_per_s"),
            "c5_strong_gaussians_per_s": dig(cfg, "secondary", "strong", "c5", "gaussians_per_s"),
            "c5_roofline_frac_render_bwd": dig(out, "roofline", "c5", "frac"),
            "dropin_ms_per_view_512": dig(cfg, "dropin", "reference_shape_512x512_8_views", "ms_per_view"),
            "dropin_ms_per_view_1200x680": dig(cfg, "dropin", "c2_1200x680_1_view", "ms_per_view"),
            "ms_per_step_bf16_split": out.get("ms_per_step_bf16_split"), "ms_per_step_bf16x3": out.get("ms_per_step_bf16x3"),
            "ms_per_step_f32_mfma": out.get("ms_per_step_f32_mfma"), "ms_per_step_pipelined": out.get("ms_per_step_pipelined"),
            "parity_rgb_L1": dig(out, "parity", "parity_rgb_L1"), "parity_grad_rel": dig(out, "parity", "parity_grad_rel"),
            "cpu_baseline_gaussians_per_s": dig(out, "cpu_baseline", "value")}
    for k, v in flat.items():
        if v is not None and k not in cfg:
            cfg[k] = v


def cpu_baseline(raw_cpu, cam_cpu, d_img_cpu, gpu_check, budget_s=15.0, max_threads=16):
    """The oracle (oracle/surfel_oracle.py, PyTorch CPU, fp32) timed on this host: the full
    per-Gaussian stage + binning of the same view, then fwd+bwd of strided batches of tiles
    until ~budget_s of CPU time is spent, extrapolated to all non-empty tiles.

    The images and gradients the oracle computes on the way are not thrown away: ``gpu_check(tile_mask)``
    runs the HIP path on the same (initial) parameters with the same image gradients restricted to the tiles
    the oracle covered, and the two are compared -> the ``parity`` object of the JSON line (the same gates as
    tests/_parity.py: contract mean-L1, largest pixel error, worst-tile mean L1, relative L1 of the gradients)."""
    from active_gs_amd.synthetic import activate
    from oracle.surfel_oracle import OracleSettings, bin_instances, preprocess, render_tiles
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _parity
    cores = min(os.cpu_count() or 1, max_threads)  # many small ops: more threads only add overhead
    torch.set_num_threads(cores)
    a = activate(raw_cpu)
    S = OracleSettings(H, W, cam_cpu["tanx"], cam_cpu["tany"], cam_cpu["bg"], 1.0, cam_cpu["view"], cam_cpu["proj"])
    ins = [a["means"].clone().requires_grad_(True), torch.zeros(N_GAUSS, 3), a["opacities"][:, None].clone().requires_grad_(True),
           a["confidences"], a["colors"].clone().requires_grad_(True), a["scales"].clone().requires_grad_(True),
           a["rotations"].clone().requires_grad_(True)]
    t0 = time.perf_counter()
    G = preprocess(*ins, S)
    so, ranges = bin_instances(G)
    t_pre = time.perf_counter() - t0
    nonempty = torch.nonzero(ranges[:, 1] > ranges[:, 0]).flatten().tolist()
    order = nonempty[::37] + [t for k in range(1, 37) for t in nonempty[k::37]]  # strided, spatially spread
    done, t_tiles, batch = 0, 0.0, 16
    tiles_x = (W + 15) // 16
    covered = torch.zeros(H, W)
    images = {k: torch.zeros(c, H, W) for k, c in (("rgb", 3), ("normal", 3), ("depth", 1), ("opacity", 1))}
    while done < len(order) and t_tiles < budget_s:
        sample = order[done:done + batch]
        m = torch.zeros(H, W)
        for t in sample:
            ty, tx = divmod(t, tiles_x)
            m[ty * 16:(ty + 1) * 16, tx * 16:(tx + 1) * 16] = 1.0
        t0 = time.perf_counter()
        R = render_tiles(G, so, ranges, S, tiles=sample)
        # the bench step's own image gradients, on the tiles of this batch
        ((R["rgb"] * (d_img_cpu[0] * m)).sum() + (R["normal"] * (d_img_cpu[1] * m)).sum()
         + (R["depth"] * (d_img_cpu[2] * m)).sum()).backward(retain_graph=True)
        t_tiles += time.perf_counter() - t0
        done += len(sample)
        covered += m
        for k in images:
            images[k] += R[k].detach() * m
    est = t_pre + t_tiles * len(nonempty) / max(done, 1)
    base = {"value": N_GAUSS / est, "unit": "Gaussians/s", "cores": cores, "kind": "port",
            "sample": f"oracle (PyTorch CPU fp32, {cores} threads): full preprocess+binning of the 200k-surfel "
                      f"1200x680 view ({t_pre:.2f}s) + fwd+bwd of {done} of {len(nonempty)} non-empty tiles "
                      f"({t_tiles:.1f}s), extrapolated to all tiles"}
    # ---- parity of the HIP path against what the oracle just computed (checker role only)
    gpu_images, gpu_grads, gpu_radii = gpu_check(covered)
    stats = {k: _parity.image_stats(images[k], gpu_images[k], covered) for k in images}
    for k in stats:
        stats[k]["outliers"] = _parity.outlier_frac(stats[k], k)
        del stats[k]["per_pixel_max"]
    ref_grads = {"means3D": ins[0].grad, "opacities": ins[2].grad.reshape(-1), "colors": ins[4].grad, "scales": ins[5].grad,
                 "rotations": ins[6].grad}
    rel = _parity.grad_stats(ref_grads, gpu_grads)
    radii_mismatch = int((gpu_radii != G["radii"]).sum())
    ok = all(stats[k]["mean"] < _parity.MEAN_L1[k] and stats[k]["max"] < _parity.MAX_ABS[k] and stats[k]["tile"] < _parity.TILE_L1[k]
             and stats[k]["outliers"] < _parity.OUTLIER_FRAC for k in stats) and max(rel.values()) < _parity.GRAD_REL
    parity = {"parity_rgb_L1": stats["rgb"]["mean"], "parity_grad_rel": max(rel.values()),
              "image_L1": {k: s["mean"] for k, s in stats.items()}, "image_max_abs": {k: s["max"] for k, s in stats.items()},
              "image_worst_tile_L1": {k: s["tile"] for k, s in stats.items()},
              "image_outlier_frac": {k: s["outliers"] for k, s in stats.items()}, "grad_rel_L1": rel,
              "radii_rows_differing": radii_mismatch, "tiles_compared": done, "tiles_nonempty": len(nonempty),
              "tolerance": {"mean_L1": _parity.MEAN_L1["rgb"], "mean_L1_depth_m": _parity.MEAN_L1["depth"], "max_abs": _parity.MAX_ABS["rgb"],
                            "worst_tile_L1": _parity.TILE_L1["rgb"], "outlier_abs": _parity.OUTLIER_ABS["rgb"],
                            "outlier_frac": _parity.OUTLIER_FRAC, "grad_rel": _parity.GRAD_REL},
              "ok": bool(ok),
              "what": "HIP forward+backward of the bench view (initial parameters, the bench step's image gradients "
                      "restricted to the compared tiles) against the oracle's fp32 images / autograd gradients"}
    return base, parity


def launch_ranks(args) -> int:
    """``--gpus N`` without a launcher: start N ranks (one per GPU) as a child job and relay rank 0's JSON line.
    Runs before anything in this process has touched the GPU (no GPU state to share, no re-exec).  On failure the
    tail of the job's stderr (every rank's last lines, the watchdog's stacks) is printed."""
    import socket
    import subprocess
    share = os.environ.get("AGS_BENCH_SHARE_GPU") == "1"
    if not share:
        have = torch.cuda.device_count()            # counting devices does not initialise the GPU
        if have < args.gpus:
            print(f"bench.py: --gpus {args.gpus} but this node has {have} GPU(s)", file=sys.stderr)
            return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this pool
    env.setdefault("NCCL_SOCKET_IFNAME", "lo")          # one node: the bootstrap never needs a NIC picked by host name
    env.setdefault("OMP_NUM_THREADS", "8")
    for attempt in (0, 1):
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        # the port was free when it was picked; if another socket took it before the store bound it, once more on another
        if r.returncode == 0 or attempt == 1 or not any(m in r.stderr for m in ("Address already in use", "EADDRINUSE")):
            break
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    other = [l for l in r.stdout.splitlines() if not l.startswith("{")]
    if other:
        print("\n".join(other), file=sys.stderr)
    if r.returncode != 0 or len(lines) != 1:
        print("\n".join(r.stderr.splitlines()[-120:]), file=sys.stderr)
        print(f"bench.py: the {args.gpus}-rank job exited with code {r.returncode} and {len(lines)} result line(s)",
              file=sys.stderr)
        return r.returncode or 1
    print(lines[0])
    return 0


def rank_identity(dev, world, rank):
    """what every rank is, gathered on all ranks: (rank, host, device index, device uuid / name)"""
    import socket
    p = torch.cuda.get_device_properties(dev)
    me = dict(rank=rank, host=socket.gethostname(), device_index=dev.index, name=p.name,
              uuid=str(getattr(p, "uuid", "")) or None, pci_bus_id=getattr(p, "pci_bus_id", None))
    everyone = [None] * world
    torch.distributed.all_gather_object(everyone, me)
    return everyone


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--samples", type=int, default=0, help="timed samples of --steps steps each (0 = automatic: >= 21, >= 0.25 s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip what is measured after the timed region besides the stage times: pipelined / bf16-split forms, "
                         "the drop-in module's call path, configurations 3, 4 and 5")
    ap.add_argument("--binning", choices=["direct", "tile_sort", "radix"], default="direct")
    ap.add_argument("--eager", action="store_true", help="time eager launches instead of hipGraph replay")
    ap.add_argument("--pipeline", action="store_true",
                    help="single GPU: time the software-pipelined step (ags_backward_fused_next: the per-Gaussian backward + Adam "
                         "of step k and the per-Gaussian forward stage of step k+1 in ONE kernel, four launches per step, -7 %%) "
                         "as the headline.  Default: the five-launch step - what every trainer of this repository that runs the "
                         "reference's loop (FusedMapTrainer, GaussianMapTrainer) executes; the pipelined form is then reported as "
                         "`ms_per_step_pipelined`, measured after the timed region")
    ap.add_argument("--no-pipeline", action="store_true", help="(the default now; kept for old command lines)")
    ap.add_argument("--graph-steps", type=int, default=int(os.environ.get("AGS_BENCH_GRAPH_STEPS", "25")),
                    help="optimisation steps recorded per hipGraph (single GPU); K steps = K/this replays")
    ap.add_argument("--check", action="store_true", help="multi-GPU readiness check only (see the module docstring)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "RANK" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args))

    # The contract is ONE JSON line on stdout.  Libraries write to stdout too - RCCL prints a version banner (five lines,
    # flushed from the C library's buffer when the process EXITS, i.e. behind the JSON line), GaussianMap.prune prints like
    # the reference's - so from here on file descriptor 1 is stderr for everybody, and only emit_line() below writes to
    # the real stdout.
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    def emit_line(obj) -> None:
        real_stdout.write(json.dumps(obj) + "\n")
        real_stdout.flush()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # A rank that hangs (a collective that never completes) must not hang the job until the driver's timeout: with more
    # than one rank every rank dumps all its threads' stacks and exits non-zero after AGS_BENCH_WATCHDOG seconds
    # (default 600; 0 = off); the launcher above then prints the tail of the job's stderr.
    wd = int(os.environ.get("AGS_BENCH_WATCHDOG", "600" if world > 1 else "0"))
    if wd > 0:
        import faulthandler
        faulthandler.dump_traceback_later(wd, exit=True)
    if os.environ.get("AGS_BENCH_TEST_HANG") == str(rank) and world > 1:      # test hook: this rank stalls (see the tests)
        time.sleep(10 ** 6)
    # the package reads no environment variable: this launcher hands it the documented AGS_* selection variables
    from active_gs_amd import env_config
    env_config.apply_env(os.environ)
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         "(python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the rasterizer has no CPU fallback")
    # test-only hooks (tests/test_gpu_bench_multirank.py): several ranks on ONE GPU over gloo, to
    # exercise the N>1 code path where only a single-GPU box is available
    share_gpu = os.environ.get("AGS_BENCH_SHARE_GPU") == "1"
    backend = os.environ.get("AGS_BENCH_BACKEND", "nccl")
    dev_index = 0 if share_gpu else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # test-only hook: AGS_DP_FORCE=1 runs the data-parallel step (exchange, union Adam, collectives in the graph)
    # in a one-rank RCCL group - the overhead of that path without any wire time, measurable on a 1-GPU box
    dist_on = world > 1 or os.environ.get("AGS_DP_FORCE") == "1"
    identity = None
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")   # ONE node (the contract): RCCL's bootstrap over loopback, no NIC picked by host name
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)   # nccl IS RCCL on ROCm
        else:
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")   # test transport, ranks on one host: never pick a NIC by hostname
            dist.init_process_group(backend)
        if dist.get_world_size() != world:
            raise SystemExit(f"bench.py: the process group has {dist.get_world_size()} ranks, WORLD_SIZE says {world}")
        identity = rank_identity(dev, world, rank)
        if not share_gpu and world > 1:
            seen = {(e["host"], e["uuid"] or e["pci_bus_id"] or e["device_index"]) for e in identity}
            if len(seen) != world:
                raise SystemExit(f"bench.py: {world} ranks but {len(seen)} distinct devices: {identity}")

    from active_gs_amd import _lib, raster_api as api
    from active_gs_amd.camera import camera_matrices
    from active_gs_amd.synthetic import make_camera, make_room_scene
    from active_gs_amd.dist_util import all_reduce_
    from active_gs_amd.trainer import SurfelTrainer

    lib = _lib.load()
    raw_cpu = make_room_scene(N_GAUSS, "office0", seed=0)
    raw = {k: v.to(dev) for k, v in raw_cpu.items()}
    # one view per rank (weak scaling): rank r looks at the room through view r//8 reflected in the
    # room's symmetry planes (mirror r%8) - same workload on every rank (13.2-13.4 k visible surfels,
    # 164-167 k rect instances), different surfels
    c2w, K = make_camera(rank // 8, H, W, mirror=rank % 8)
    cm = camera_matrices(c2w[None], K[None], 0.001, 10.0)
    tanx, tany = cm["tanfov"][0, 0].item(), cm["tanfov"][0, 1].item()
    bg = torch.zeros(4)
    cam = api.Camera(H, W, tanx, tany, cm["viewmatrix"][0].to(dev), cm["projmatrix"][0].to(dev), bg.to(dev))
    bin_mode = {"direct": api.BIN_DIRECT, "tile_sort": api.BIN_TILE_SORT, "radix": api.BIN_RADIX}[args.binning]
    trainer = SurfelTrainer(raw, binning_mode=bin_mode)
    headline_reduce = int(_lib.default_tuning().bwd_reduce)     # which form of the blend backward's sums the timed region runs

    # size the workspace from one probing forward (outside the timed region)
    g = trainer.gaussians()
    probe = api.alloc_state(N_GAUSS, H, W, 16_000_000, dev, bin_mode)
    api.forward(cam, g, probe)
    info = api.read_status(probe)
    assert not info["overflow"], info
    I, V = info["num_instances"], info["num_visible"]
    del probe
    cap = int(info["needed"] * 1.3) + 4096   # direct binning: tiles x longest tile list; else the instance total

    gen = torch.Generator().manual_seed(1234 + rank)
    P = H * W
    scale = 1.0 / (P * world)
    d_img = [(torch.randn(c, H, W, generator=gen) * scale).to(dev) for c in (3, 3, 1)]
    grads_fn = lambda v, st: (d_img[0], d_img[1], d_img[2], None, None)

    # (AGS_BENCH_EAGER_PIPELINE=1: the eager steps are software-pipelined too - for counter passes over the fused
    # per-Gaussian kernel; the per-stage event profile wants the five-launch form, so it is not the default)
    eager_next = cam if os.environ.get("AGS_BENCH_EAGER_PIPELINE") == "1" else None

    def eager_step():
        trainer.step([cam], grads_fn, cap, next_cam=eager_next)

    for _ in range(3):
        eager_step()  # creates every buffer before capture
    torch.cuda.synchronize()

    if args.check:
        # readiness: the process group is up, the ranks are distinct devices, one eager data-parallel step (all-gather
        # of the rows or all-reduce of the slab) has run, and the captured-collective probe says which form a captured
        # step would take
        path = None
        if dist_on:
            in_graph = trainer._collectives_capturable()
            rows_x = trainer._row_exchange_on()
            path = ("rows" if rows_x else "dense") + (" in graph" if in_graph else ": graph | collective | graph")
            paths = [None] * world
            torch.distributed.all_gather_object(paths, path)
            if len(set(paths)) != 1:
                raise SystemExit(f"bench.py --check: ranks disagree on the exchange path: {paths}")
        refused0 = trainer.refused_steps() if dist_on else 0
        strong = None
        if dist_on and not args.no_extras:
            # the path every rank takes for the strong-scaled configurations 4 and 5 (one sizing pass + two eager steps each)
            del trainer
            torch.cuda.empty_cache()
            strong = {}
            for key, cfg in strong_configs().items():
                r = measure_strong(key, cfg, 0, dev, world, rank, dist_on, check_only=True)
                strong[key] = {k: r[k] for k in ("exchange_path", "exchange_bytes_per_rank", "views_this_rank", "surfels",
                                                 "member_rows_this_rank", "replicas_identical")}
        if rank == 0:
            emit_line({"check": "ok", "n_gpus": world, "backend": backend if dist_on else None, "exchange_path": path,
                       "ranks": identity, "refused_steps": refused0, "strong_configs": strong})
        if dist_on:
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()
        return

    def capture(pipe):
        """-> (one_step, many_steps, steps per replay of many_steps, description)"""
        mode = "hipGraph replay"
        timed_kernel["per_gaussian_forward"] = "ags_k_preprocess_cull" if trainer.cull_first_kernel() else "ags_k_preprocess<2>"
        one = trainer.capture([cam], grads_fn, cap, pipeline=pipe)
        in_graph = getattr(one, "collective_in_graph", False)
        if dist_on:
            mode = ("hipGraph replay, gradient exchange recorded in the graph" if in_graph
                    else "hipGraph replay: graph | collective | graph")
        # steps per graph: the largest divisor of K that is <= --graph-steps, so that a sample is whole replays of ONE graph
        rep = max(d for d in range(1, max(1, args.graph_steps) + 1) if args.steps % d == 0)
        many, per = None, 1
        if rep > 1 and (not dist_on or in_graph):
            many = trainer.capture([cam], grads_fn, cap, repeat=rep, pipeline=pipe)
            per = many.steps
        mode += f", {per} step(s) per graph, {args.steps // per} replay(s) per sample"
        if getattr(one, "pipelined", False):
            mode += ("; software-pipelined: 4 launches per step - tile sort, blend, blend backward, [chain rule + Adam "
                     "of this step and the per-Gaussian stage (cull, project, key emission) of the next step] - every "
                     "step still does one of each stage")
        elif not dist_on:
            mode += "; 5 launches per step: per-Gaussian stage, tile sort, blend, blend backward, chain rule + Adam"
        return one, many, per, mode

    launch_mode = "eager" if args.eager else "hipGraph replay"
    # which per-Gaussian forward kernel the timed graphs were recorded with (the trainer picks it from what its views show,
    # SurfelTrainer._adapt_kernels: a graph keeps the kernel it was captured with)
    timed_kernel = {"per_gaussian_forward": "ags_k_preprocess_cull" if trainer.cull_first_kernel() else "ags_k_preprocess<2>"}
    one_step, many_steps, per_replay = eager_step, None, 1
    pipe = args.pipeline and not dist_on and not args.no_pipeline and args.binning == "direct"
    if not args.eager:
        try:
            one_step, many_steps, per_replay, launch_mode = capture(pipe)
        except Exception as e:  # never lose the measurement to a capture problem
            launch_mode = f"eager (graph capture failed: {type(e).__name__}: {e})"
            one_step, many_steps, per_replay = eager_step, None, 1
            torch.cuda.synchronize()

    def make_runner(one, many, per):
        def run_steps(k):
            """exactly k optimisation steps"""
            if many is not None:
                for _ in range(k // per):
                    many()
                k = k % per
            for _ in range(k):
                one()
        return run_steps

    run_steps = make_runner(one_step, many_steps, per_replay)

    # bring the GPU to its sustained clock before the W warm-up steps (a step is ~0.1 ms:
    # W of them alone finish before DVFS has settled)
    t_pre = time.perf_counter()
    while True:
        for _ in range(20):
            one_step()
        torch.cuda.synchronize()
        go = time.perf_counter() - t_pre < 0.5
        if dist_on:   # every rank must run the same number of steps (each one is a collective): rank clocks differ
            flag = torch.tensor([1 if go else 0], device=dev, dtype=torch.int32)
            all_reduce_(flag, torch.distributed.ReduceOp.MAX)
            go = bool(flag.item())
        if not go:
            break
    run_steps(args.warmup)
    torch.cuda.synchronize()
    # one untimed sample tells how many samples make >= 0.25 s; every rank must take the same number
    est = time_samples(lambda: run_steps(args.steps), 1, dist_on, dev)[0]
    n_samples = args.samples if args.samples > 0 else int(min(201, max(21, 0.25 / max(est, 1e-6) + 1)))
    # ---------------------------------------------------------------- the timed region
    t0 = time.perf_counter()
    run_steps(args.steps)
    enqueue_s = time.perf_counter() - t0
    torch.cuda.synchronize()
    samples = time_samples(lambda: run_steps(args.steps), n_samples, dist_on, dev)
    # ----------------------------------------------------------------
    timing = summarise(samples, args.steps)
    st = trainer.state_for(H, W, cap)
    info = api.read_status(st)
    refused = trainer.refused_steps() if dist_on else 0      # steps the row exchange refused (segment outgrown)

    extras = {}
    if not dist_on and not args.eager and not args.no_extras and args.binning == "direct":
        # the OTHER form of the same step, sampled the same way: software-pipelined (four launches: a loop that can name
        # its next view one step early, SurfelTrainer.step(next_cam=)) when the headline is the five-launch step, and
        # vice versa
        other = "ms_per_step_no_pipeline" if pipe else "ms_per_step_pipelined"
        try:
            if pipe:
                trainer.step([cam], grads_fn, cap)      # an un-pipelined step consumes the prepared pass: the pipeline is left
            o2, m2, p2, _ = capture(not pipe)
            r2 = make_runner(o2, m2, p2)
            r2(args.warmup)
            extras[other] = summarise(time_samples(lambda: r2(args.steps), max(5, n_samples // 3), False, dev), args.steps)["median"]
            if not pipe:
                trainer.step([cam], grads_fn, cap)      # leave the pipeline again: the stage pass below is un-pipelined
        except Exception as e:
            extras[other] = None
            extras["other_form_note"] = f"{type(e).__name__}: {e}"
            torch.cuda.synchronize()

    # per-stage kernel time: K steps launched eagerly with library-owned HIP events
    # around every stage (events cannot be timed inside a replayed graph)
    k_prof = min(args.steps, 200)
    _lib.check(lib.ags_profile_enable(k_prof), "ags_profile_enable")
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    profile_note = None
    try:
        for _ in range(k_prof):
            eager_step()
    except RuntimeError as e:     # e.g. a very long run whose drifting synthetic scene outgrew the workspace sized at its start
        profile_note = f"per-stage pass stopped early: {e}"
    torch.cuda.synchronize()
    eager_elapsed = time.perf_counter() - t1
    stage_ms, stage_mean_ms = read_stage_times(lib)
    lib.ags_profile_enable(0)

    strong = None
    if dist_on and not args.no_extras and not args.eager:
        # BASELINE.json's configurations 4 and 5 at THIS number of ranks (strong scaling: the views of a step are dealt out
        # over the ranks); every rank takes part (collectives), rank 0 reports
        strong = {}
        # These configurations are SECONDARY fields of the line: a failure that every rank sees alike (same software, same
        # sizes: the likely kind) is recorded as text and the weak-scaled headline survives.  A failure only SOME ranks see
        # (out of memory with 32 workspaces, an overflow on one rank's views) cannot be recovered from here - the peers sit in
        # a collective the failed rank never joins, and the job ends with the watchdog's stack dump (AGS_BENCH_WATCHDOG).
        # The ranks agree on the outcome (one all-reduce of a flag) before anybody starts the next configuration, so a rank
        # that failed late - after the configuration's last collective - does not take its peers' next collectives apart.
        for key, cfg in strong_configs().items():
            try:
                strong[key] = measure_strong(key, cfg, 10, dev, world, rank, dist_on)
                failed = 0
            except Exception as e:
                strong[key] = f"{type(e).__name__}: {e}"
                failed = 1
                torch.cuda.synchronize()
            # agree on the outcome before anybody starts the next configuration's collectives
            flag = torch.tensor([failed], device=dev, dtype=torch.int32)
            all_reduce_(flag, torch.distributed.ReduceOp.MAX)
            if int(flag.item()) and not failed:
                strong[key] = "another rank failed in this configuration (see its stderr)"

    if rank == 0:
        T = ((H + 15) // 16) * ((W + 15) // 16)
        rows = int(trainer.rows.count.item()) if getattr(trainer, "rows", None) is not None else None
        x = getattr(trainer, "exchange", None)
        if not dist_on:
            exchange = None
        elif x is not None and x.capacity:
            exchange = {"kind": "all-gather of member rows (64 B per row)", "rows_per_segment": x.capacity,
                        "bytes_per_rank": 4 * x.send.numel(), "union_rows": int(x.union.count.item()),
                        "overflow": x.overflowed(), "refused_steps": refused, "regrowths": trainer.exchange_regrowths}
        else:
            exchange = {"kind": "all-reduce of the dense gradient slab", "bytes_per_rank": 4 * trainer.slab.flat.numel()}
        if exchange is not None:
            exchange["world_size"] = torch.distributed.get_world_size()
            exchange["backend"] = backend
            exchange["ranks"] = identity
        sb = stage_bytes(N_GAUSS, V, I, P, T, rows, direct=(args.binning == "direct"), fused_single_view=not dist_on)
        dom = max(stage_ms, key=lambda k: stage_ms[k])
        ach = sb[dom] / (stage_ms[dom] * 1e-3) / 1e9 if stage_ms[dom] > 0 else 0.0
        # Counter figures cannot be collected from inside this process: they come from the committed rocprofv3
        # --pmc runs of THIS command on this build (profiles/, made by profiles/experiments/pmc_run.sh) and are
        # labelled as such; per-launch instruction counts and HBM bytes of a fixed workload do not depend on the run.
        traffic = traffic_src = traffic_read = traffic_write = valu = None
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", PMC_HBM_FILE)))
            traffic, traffic_read, traffic_write, how = pmc_entry(pm.get(dom, {}))
            traffic_src = f"profiles/{PMC_HBM_FILE} (rocprofv3 --pmc passes of this command, session {pm.get('_session', '?')}; " \
                          f"not collected in this run; {how})"
        except Exception:
            pass
        try:
            sq = json.load(open(os.path.join(ROOT, "profiles", PMC_SQ_FILE)))
            blend = {}
            for k in ("render_fwd", "render_bwd"):
                act = sq.get(k, {}).get("SQ_ACTIVE_INST_VALU")
                if act and stage_ms[k] > 0:
                    # SQ_ACTIVE_INST_VALU counts quad-cycles summed over the SIMDs; 256 CUs x 4 SIMDs at 2.4 GHz
                    blend[k] = {"valu_busy_frac": act * 4.0 / (stage_ms[k] * 1e-3 * 2.4e9 * 1024),
                                "valu_insts_per_launch": sq[k].get("SQ_INSTS_VALU")}
            if blend:
                valu = {"bound": "valu", "kernel": dom if dom in blend else "render_bwd",
                        "achieved": blend.get(dom, blend.get("render_bwd", {})).get("valu_busy_frac"), "peak": 1.0,
                        "unit": "fraction of VALU issue cycles busy (1024 SIMDs x 2.4 GHz)", "kernels": blend,
                        "source": f"instruction counters from profiles/{PMC_SQ_FILE} (session {sq.get('_session', '?')}, "
                                  "not collected in this run) / this run's HIP-event stage times"}
        except Exception:
            pass
        ms_per_step = timing["median"]
        step_s = ms_per_step * 1e-3
        out = {
            "metric": "splatted-Gaussians/s (fwd+bwd) @1200x680; achieved HBM GB/s vs peak",
            "value": N_GAUSS * world / step_s,
            "unit": "Gaussians/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "ms_per_step_min": timing["min"], "ms_per_step_max": timing["max"], "samples": timing["samples"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": DTYPE,
            "data": "synthetic",
            "config": {"workload": "office0 stand-in (seeded box room), 200k surfels, 1200x680, 1 view per GPU (rank r: "
                                   "view 0 mirrored through the room's symmetry planes = equal work per rank), "
                                   "step = activations + fwd + bwd (fed fixed random image gradients d_rgb, d_normal, "
                                   "d_depth: no loss head in the step) + gradient-row exchange (N>1) + Adam",
                       "timing": f"{timing['samples']} samples of exactly {args.steps} steps each, every sample bracketed by "
                                 "barrier + torch.cuda.synchronize() on both sides, max over ranks per sample; ms_per_step "
                                 "and value are the MEDIAN sample",
                       "gaussians": N_GAUSS, "image": [H, W], "views_per_gpu": 1, "visible": V,
                       "tile_instances": I, "parallelism": f"view-parallel dp{world}",
                       "overflow": bool(info["overflow"]) or bool(info["overflow_passes"]) or refused > 0,
                       "overflow_passes": int(info["overflow_passes"]), "profile_note": profile_note,
                       "binning": args.binning,
                       "launch": launch_mode,
                       "per_gaussian_forward_kernel": timed_kernel["per_gaussian_forward"],
                       "blend_backward_reduce": {v: k for k, v in _lib._BWD_NAMES.items() if k != "bf16_split"}[headline_reduce],
                       "optimizer": ("row-set Adam fused into the per-Gaussian backward (exact: untouched rows "
                                     f"have zero gradient and moments), {rows} member rows") if (rows is not None and not dist_on)
                                    else ("row-set Adam over the union of the ranks' member rows" if rows is not None
                                          else "dense fused Adam kernel after the gradient all-reduce"),
                       "exchange": exchange,
                       # `value` counts SUBMITTED Gaussians (the metric's definition): most of them are culled
                       # by this view.  The rates below count what reaches the blend kernels.
                       "derived_rates": {"visible_gaussians_per_s": V * world / step_s,
                                         "tile_instances_per_s": I * world / step_s,
                                         "visible_fraction": V / N_GAUSS},
                       "host_enqueue_ms_per_step": round(enqueue_s / args.steps * 1e3, 4),
                       "eager_ms_per_step": round(eager_elapsed / max(k_prof, 1) * 1e3, 4),
                       "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
                       "stage_mean_ms": {k: round(v, 4) for k, v in stage_mean_ms.items()},
                       "stage_timing": f"HIP events on the launch stream around every stage of {k_prof} eager steps "
                                       "run right after the timed region (medians"
                                       + ("; the un-pipelined five-launch form of the step: `preprocess` and "
                                          "`preprocess_bwd` are separate kernels there, one launch in the timed region"
                                          if getattr(one_step, "pipelined", False) else "")
                                       + "); an event pair reads ~3 us more than rocprofv3's kernel duration"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_read": traffic_read,
                         "traffic_write": traffic_write, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": sb[dom],
                         "algorithmic_note": "render_bwd: 8 T + 68 I + 44 P + 64 V (the step passes d_rgb, d_normal, d_depth; "
                                             "d_opacity / d_confidence are not given and not read).  The kernel adds one "
                                             "64-byte gradient record per (surfel, wave that blended it) with atomics, not "
                                             "one per surfel: its WRITE traffic exceeds the algorithmic 64 V by that factor "
                                             "(traffic_write), its READ traffic matches the algorithmic reads",
                         "all_stages_GBps": {k: (sb[k] / (stage_ms[k] * 1e-3) / 1e9 if stage_ms[k] > 0 else 0.0)
                                             for k in sb},
                         "all_stages_frac": {k: (sb[k] / (stage_ms[k] * 1e-3) / 1e9 / HBM_PEAK_GBS if stage_ms[k] > 0 else 0.0)
                                             for k in sb},
                         "whole_step_GBps": sum(sb.values()) / step_s / 1e9},
        }
        out.update(extras)
        if valu is not None:
            out["roofline_valu"] = valu
        if strong is not None:
            out["config"]["secondary"] = {"strong": strong, **strong,
                                          "note": "configurations 4 and 5 strong-scaled over this run's ranks, measured after the "
                                                  "timed region; `value` / `ms_per_step` above are weak-scaled configuration 2"}
        if not dist_on and not args.no_extras and not args.eager:
            # -- the same step with the blend backward's per-surfel sums on the bf16 matrix pipe (hi/lo splits, f32
            #    accumulation: AgsTuning.bwd_reduce = AGS_BWD_BF16_SPLIT, an opt-in per workspace) - same process, same
            #    library, a second trainer on a fresh copy of the scene, sampled like the headline
            notes = {"bf16": ("ms_per_step_bf16_split", "bf16_split_note",
                              "opt-in (AgsTuning.bwd_reduce = AGS_BWD_BF16_SPLIT): the blend backward's per-surfel sums "
                              "from bf16 hi/lo splits of both operands, f32 accumulation; gradients move by ~1e-5 "
                              "relative; `value` / `ms_per_step` are the default form"),
                     "bf16x3": ("ms_per_step_bf16x3", "bf16x3_note",
                                "AgsTuning.bwd_reduce = AGS_BWD_BF16X3: the same sums from an exact three-way bf16 split of both "
                                "operands (six products, f32 accumulation; every multiply-add within 2^-24 of the exact product); "
                                "error against fp64 gradients: profiles/r06_bwd_reduce_error.md"),
                     "f32": ("ms_per_step_f32_mfma", "f32_mfma_note",
                             "AgsTuning.bwd_reduce = AGS_BWD_F32: exact f32 matrix instructions")}
            for mode, (key, note_key, text) in notes.items():
                if _lib._BWD_NAMES[mode] == headline_reduce:
                    continue                                   # (that form IS the headline)
                try:
                    t2 = SurfelTrainer({k: v.to(dev) for k, v in raw_cpu.items()}, binning_mode=bin_mode,
                                       tuning=_lib.make_tuning(bwd_reduce=mode))
                    t2._tuning_pinned = False          # (the per-Gaussian kernel is chosen from the views like the headline's)
                    for _ in range(3):
                        t2.step([cam], grads_fn, cap)
                    torch.cuda.synchronize()
                    rep = max(d for d in range(1, max(1, args.graph_steps) + 1) if args.steps % d == 0)
                    many2 = t2.capture([cam], grads_fn, cap, repeat=rep, pipeline=pipe)
                    run2 = lambda: [many2() for _ in range(args.steps // many2.steps)]
                    for _ in range(max(1, args.warmup // max(args.steps, 1))):
                        run2()
                    out[key] = summarise(time_samples(run2, max(5, n_samples // 3), False, dev), args.steps)["median"]
                    out[note_key] = text
                    t2.check_overflow()
                    del t2, many2
                except Exception as e:
                    out[key] = None
                    out[note_key] = f"{type(e).__name__}: {e}"
                    torch.cuda.synchronize()
            # -- the path an UNMODIFIED caller takes: the drop-in module under autograd, per view
            try:
                d1 = measure_dropin(dev, N_GAUSS, H, W, 1, iters=40)
                d2 = measure_dropin(dev, N_GAUSS, 512, 512, 8, iters=8, focal=0.5 * 512 / 0.57735)
                d1d = measure_dropin(dev, N_GAUSS, H, W, 1, iters=40, deferred=True)
                d2d = measure_dropin(dev, N_GAUSS, 512, 512, 8, iters=8, focal=0.5 * 512 / 0.57735, deferred=True)
                out["config"]["dropin_ms_per_view"] = round(d1["ms_per_view"], 4)
                out["config"]["dropin"] = {"c2_1200x680_1_view": d1, "reference_shape_512x512_8_views": d2,
                                           "c2_1200x680_1_view_deferred": d1d, "reference_shape_512x512_8_views_deferred": d2d,
                                           "what": "diff_gaussian_rasterization_2d.GaussianRasterizer under autograd, forward + "
                                                   "backward per view (no facade, no loss head, no optimiser), measured after "
                                                   "the timed region"}
            except Exception as e:
                out["config"]["dropin_ms_per_view"] = None
                out["config"]["dropin"] = f"{type(e).__name__}: {e}"
            # -- BASELINE.json's configuration 3: the full mapper loop (grow from every new keyframe, train 10 iterations,
            #    post-process / prune) for 500 iterations from an empty map, through the drop-in GaussianMap class - the
            #    calls /root/reference/mapping/mapper.py:44,101 makes - at the reference's 512x512 and at 1200x680
            try:
                del trainer, one_step, many_steps
                torch.cuda.empty_cache()
                out["config"]["c3"] = measure_c3(dev)
            except Exception as e:
                out["config"]["c3"] = f"{type(e).__name__}: {e}"
                torch.cuda.synchronize()
            # -- one GPU's share of BASELINE.json's configurations 4 and 5
            try:
                torch.cuda.empty_cache()
                sec = {"c4_share": measure_config("c4: one GPU's 4 of 32 views, 1.5 M surfels @1200x680", 1_500_000, 680, 1200, 4,
                                                  "room0", 20, dev),
                       "c5": measure_config("c5: 5 M surfels @2048x2048, 1 view", 5_000_000, 2048, 2048, 1, "office0", 20, dev)}
                out["config"]["secondary"] = {"c4_share_ms": sec["c4_share"]["ms_per_step"], "c5_ms": sec["c5"]["ms_per_step"], **sec}
                # the same roofline object for configuration 5's own dominant kernel (5 M surfels @2048x2048: the size at
                # which the blend backward dominates the step)
                c5 = sec["c5"]
                dom5 = max(c5["stage_ms_per_view"], key=lambda k: c5["stage_ms_per_view"][k])
                t5 = c5["stage_ms_per_view"][dom5] * 1e-3
                out["roofline"]["c5"] = {"bound": "hbm", "kernel": dom5, "achieved": c5["stage_algorithmic_bytes"][dom5] / t5 / 1e9,
                                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": c5["stage_algorithmic_bytes"][dom5] / t5 / 1e9 / HBM_PEAK_GBS,
                                         "algorithmic_bytes_per_launch": c5["stage_algorithmic_bytes"][dom5], "launch_ms": c5["stage_ms_per_view"][dom5],
                                         "traffic": None}
                try:      # counter figures of configuration 5's steady-state launches: a committed rocprofv3 --pmc run of
                          # profiles/experiments/c5_counters.sh (eager steps on a frozen scene; not collected in this run)
                    pm5 = json.load(open(os.path.join(ROOT, "profiles", PMC5_HBM_FILE)))
                    e5 = pm5.get(dom5, {})
                    t5_, r5_, w5_, how5 = pmc_entry(e5)
                    out["roofline"]["c5"].update(traffic=t5_, traffic_read=r5_, traffic_write=w5_,
                                                 traffic_source=f"profiles/{PMC5_HBM_FILE} (rocprofv3 --pmc passes over eager steps of "
                                                                f"configuration 5, session {pm5.get('_session', '?')}; not collected in this "
                                                                f"run; {how5})")
                except Exception:
                    pass
            except Exception as e:
                out["config"]["secondary"] = f"{type(e).__name__}: {e}"
            # -- the one-rank end of the strong-scaled forms of configurations 4 and 5 (ALL the step's views on this GPU):
            #    what `bench.py --gpus N` reports as config.secondary.c4 / .c5 at N ranks
            try:
                torch.cuda.empty_cache()
                st_ = {key: measure_strong(key, cfg, 10, dev, 1, 0, False) for key, cfg in strong_configs().items()}
                if isinstance(out["config"].get("secondary"), dict):
                    out["config"]["secondary"]["strong"] = st_
                else:
                    out["config"]["secondary_strong"] = st_
            except Exception as e:
                out["config"]["secondary_strong"] = f"{type(e).__name__}: {e}"
                torch.cuda.synchronize()
        if not args.no_cpu_baseline and world == 1:   # the CPU leg is timed at N=1 only (contract)
            d_cpu = [t.cpu() for t in d_img]

            def gpu_check(tile_mask):
                """HIP forward + backward on the INITIAL parameters (the trainer has moved its copy) with the step's
                image gradients restricted to the tiles the oracle covered."""
                from active_gs_amd.synthetic import activate
                a0 = activate(raw_cpu)
                g0 = api.Gaussians(*(a0[k].to(dev).contiguous() for k in ("means", "scales", "rotations", "opacities",
                                                                          "colors", "confidences")))
                s0 = api.alloc_state(N_GAUSS, H, W, cap, dev, bin_mode)
                api.forward(cam, g0, s0)
                m = tile_mask.to(dev)
                gr = api.backward(cam, g0, s0, (d_img[0] * m).contiguous(), (d_img[1] * m).contiguous(),
                                  (d_img[2] * m).contiguous())
                torch.cuda.synchronize()
                assert not api.read_status(s0)["overflow"]
                imgs = {"rgb": s0.rgb.cpu(), "normal": s0.normal.cpu(), "depth": s0.depth.cpu(), "opacity": s0.opacity.cpu()}
                grads = {k: getattr(gr, k).cpu() for k in ("means3D", "opacities", "colors", "scales", "rotations")}
                return imgs, grads, s0.radii.cpu()

            out["cpu_baseline"], out["parity"] = cpu_baseline(
                raw_cpu, dict(tanx=tanx, tany=tany, bg=bg, view=cm["viewmatrix"][0], proj=cm["projmatrix"][0]), d_cpu, gpu_check)
        flatten_scalars(out)
        emit_line(out)
    if dist_on:
        torch.distributed.barrier()  # rank 0 may still be in the CPU-baseline leg; leave together
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
