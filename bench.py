#!/usr/bin/env python3
"""bench.py — splatted-Gaussians/s (fwd+bwd) @1200x680 on MI355X (BASELINE.json metric).

One step = one optimisation step of the hot path on resident data: activations ->
forward rasterization -> backward rasterization (with fixed synthetic image gradients) ->
activation backward -> [exchange of the gradient rows when N>1] -> fused Adam.
Workload at every N: BASELINE.json configs[1] per GPU (office0 stand-in, 200k surfels,
1200x680, one view per rank, weak scaling).  Prints ONE JSON line on rank 0.

``python bench.py --gpus N`` with N > 1 and no RANK in the environment launches the N ranks itself
(``python -m torch.distributed.run``, one process per GPU, RCCL) BEFORE this process touches the GPU and
relays rank 0's line; under ``torch.distributed.run`` (RANK set) it is one of the ranks.
"""
import argparse
import json
import os
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


def stage_bytes(N, V, I, P, T, rows=None, direct=True, fused_single_view=True):
    """ALGORITHMIC bytes per launch of each stage: every logical
    array moved once.  ``rows``: member rows of the sticky row set when the per-Gaussian backward
    runs in its row-set form with the Adam step fused in (single rank), else None.  ``direct``: one-pass
    binning (the per-Gaussian kernel writes the keys, the sort stage copies them into slot order)."""
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
        "render_bwd": 8 * T + 68 * I + 52 * P + 64 * V,         # + 9 grads, depth, opac, T, n; accumulate dgeom
        "preprocess_bwd": pbwd,
    }


STAGE_IDS = {"preprocess": 0, "binning": 1, "render_fwd": 2, "render_bwd": 3, "preprocess_bwd": 4}


def cpu_baseline(raw_cpu, cam_cpu, d_img_cpu, gpu_check, budget_s=15.0, max_threads=16):
    """The oracle (oracle/surfel_oracle.py, PyTorch CPU, fp32) timed on this host: the full
    per-Gaussian stage + binning of the same view, then fwd+bwd of strided batches of tiles
    until ~budget_s of CPU time is spent, extrapolated to all non-empty tiles.

    The images and gradients the oracle computes on the way are not thrown away: ``gpu_check(tile_mask)``
    runs the HIP path on the same (initial) parameters with the same image gradients restricted to the tiles
    the oracle covered, and the two are compared -> the ``parity`` object of the JSON line."""
    from active_gs_amd.synthetic import activate
    from oracle.surfel_oracle import OracleSettings, bin_instances, preprocess, render_tiles
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
                      f"1200x680 view ({t_pre:.1f}s) + fwd+bwd of {done} of {len(nonempty)} non-empty tiles "
                      f"({t_tiles:.1f}s), extrapolated to all tiles"}
    # ---- parity of the HIP path against what the oracle just computed (checker role only)
    gpu_images, gpu_grads = gpu_check(covered)
    npx = float(covered.sum())
    l1 = {k: float(((gpu_images[k] - images[k]) * covered).abs().sum() / (npx * images[k].shape[0])) for k in images}
    ref_grads = {"means3D": ins[0].grad, "opacities": ins[2].grad.reshape(-1), "colors": ins[4].grad, "scales": ins[5].grad,
                 "rotations": ins[6].grad}
    rel = {k: float((gpu_grads[k] - r).abs().sum() / r.abs().sum().clamp_min(1e-30)) for k, r in ref_grads.items()}
    parity = {"parity_rgb_L1": l1["rgb"], "parity_grad_rel": max(rel.values()),
              "image_L1": l1, "grad_rel_L1": rel, "tiles_compared": done, "tiles_nonempty": len(nonempty),
              "tolerance": {"rgb_L1": 1e-4, "grad_rel": 1e-3},
              "ok": bool(l1["rgb"] < 1e-4 and max(rel.values()) < 1e-3),
              "what": "HIP forward+backward of the bench view (initial parameters, the bench step's image gradients "
                      "restricted to the compared tiles) against the oracle's fp32 images / autograd gradients"}
    return base, parity


def launch_ranks(args) -> int:
    """``--gpus N`` without a launcher: start N ranks (one per GPU) as a child job and relay rank 0's JSON line.
    Runs before anything in this process has touched the GPU (no GPU state to share, no re-exec)."""
    import socket
    import subprocess
    share = os.environ.get("AGS_BENCH_SHARE_GPU") == "1"
    if not share:
        have = torch.cuda.device_count()            # counting devices does not initialise the GPU
        if have < args.gpus:
            print(f"bench.py: --gpus {args.gpus} but this node has {have} GPU(s)", file=sys.stderr)
            return 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this pool
    env.setdefault("OMP_NUM_THREADS", "8")
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    other = [l for l in r.stdout.splitlines() if not l.startswith("{")]
    if other:
        print("\n".join(other), file=sys.stderr)
    if r.returncode != 0 or len(lines) != 1:
        print(f"bench.py: the {args.gpus}-rank job exited with code {r.returncode} and {len(lines)} result line(s)",
              file=sys.stderr)
        return r.returncode or 1
    print(lines[0])
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--binning", choices=["direct", "tile_sort", "radix"], default="direct")
    ap.add_argument("--eager", action="store_true", help="time eager launches instead of hipGraph replay")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="single GPU: five launches per step (the per-Gaussian backward + Adam of step k and the per-Gaussian "
                         "forward stage of step k+1 as two kernels) instead of the software-pipelined four "
                         "(ags_backward_fused_next: both in ONE kernel, -4 %% step time)")
    ap.add_argument("--graph-steps", type=int, default=int(os.environ.get("AGS_BENCH_GRAPH_STEPS", "25")),
                    help="optimisation steps recorded per hipGraph (single GPU); K steps = K/this replays")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "RANK" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args))
    if os.environ.get("AGS_BENCH_WATCHDOG"):   # debugging aid: dump every thread's stack and exit after N s
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["AGS_BENCH_WATCHDOG"]), exit=True)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
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
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)   # nccl IS RCCL on ROCm
        else:
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")   # test transport, ranks on one host: never pick a NIC by hostname
            dist.init_process_group(backend)

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
    launch_mode = "eager" if args.eager else "hipGraph replay"
    one_step = eager_step
    many_steps, per_replay = None, 1
    if not args.eager:
        try:
            pipe = not dist_on and not args.no_pipeline and args.binning == "direct"
            one_step = trainer.capture([cam], grads_fn, cap, pipeline=pipe)
            in_graph = getattr(one_step, "collective_in_graph", False)
            if dist_on:
                launch_mode = ("hipGraph replay, gradient exchange recorded in the graph" if in_graph
                               else "hipGraph replay: graph | collective | graph")
            # steps per graph: the largest divisor of K that is <= --graph-steps, so that the timed region is
            # whole replays of ONE graph and this string describes exactly what was timed
            rep = max(d for d in range(1, max(1, args.graph_steps) + 1) if args.steps % d == 0)
            if rep > 1 and (not dist_on or in_graph):
                many_steps = trainer.capture([cam], grads_fn, cap, repeat=rep, pipeline=pipe)
                per_replay = many_steps.steps
            launch_mode += f", {per_replay} step(s) per graph, {args.steps // per_replay} replay(s) timed"
            if getattr(one_step, "pipelined", False):
                launch_mode += ("; software-pipelined: 4 launches per step - tile sort, blend, blend backward, [chain rule + Adam "
                                "of this step and the per-Gaussian stage (cull, project, key emission) of the next step] - every "
                                "step still does one of each stage")
        except Exception as e:  # never lose the measurement to a capture problem
            launch_mode = f"eager (graph capture failed: {type(e).__name__})"
            many_steps, per_replay = None, 1
            torch.cuda.synchronize()

    def run_steps(k):
        """exactly k optimisation steps"""
        if many_steps is not None:
            for _ in range(k // per_replay):
                many_steps()
            k = k % per_replay
        for _ in range(k):
            one_step()

    def barrier():
        if dist_on:
            torch.distributed.barrier()

    # bring the GPU to its sustained clock before the W warm-up steps (a step is ~0.2 ms:
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
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(args.steps)
    enqueue_s = time.perf_counter() - t0
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist_on:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        all_reduce_(t, torch.distributed.ReduceOp.MAX)
        elapsed = t.item()
    st = trainer.state_for(H, W, cap)
    info = api.read_status(st)
    refused = trainer.refused_steps() if dist_on else 0      # steps the row exchange refused (segment outgrown)

    # per-stage kernel time: the same K steps launched eagerly with library-owned HIP events
    # around every stage (events cannot be timed inside a replayed graph)
    _lib.check(lib.ags_profile_enable(args.steps), "ags_profile_enable")
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    profile_note = None
    try:
        for _ in range(args.steps):
            eager_step()
    except RuntimeError as e:     # e.g. a very long run whose drifting synthetic scene outgrew the workspace sized at its start
        profile_note = f"per-stage pass stopped early: {e}"
    torch.cuda.synchronize()
    eager_elapsed = time.perf_counter() - t1

    import ctypes as C
    stage_ms, stage_mean_ms = {}, {}
    for name, sid in STAGE_IDS.items():
        ms, med, cnt = C.c_float(), C.c_float(), C.c_int32()
        _lib.check(lib.ags_profile_read(sid, C.byref(ms), C.byref(med), C.byref(cnt)), "ags_profile_read")
        stage_ms[name] = med.value       # median over the K eager steps (robust to host stalls)
        stage_mean_ms[name] = ms.value
    lib.ags_profile_enable(0)

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
        sb = stage_bytes(N_GAUSS, V, I, P, T, rows, direct=(args.binning == "direct"), fused_single_view=not dist_on)
        dom = max(stage_ms, key=lambda k: stage_ms[k])
        ach = sb[dom] / (stage_ms[dom] * 1e-3) / 1e9 if stage_ms[dom] > 0 else 0.0
        # Counter figures cannot be collected from inside this process: they come from the committed rocprofv3
        # --pmc runs of THIS command on this build (profiles/, made by profiles/experiments/pmc_run.sh) and are
        # labelled as such; per-launch instruction counts and HBM bytes of a fixed workload do not depend on the run.
        traffic, traffic_src, valu = None, None, None
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", PMC_HBM_FILE)))
            traffic = pm.get(dom, {}).get("traffic")
            traffic_src = f"profiles/{PMC_HBM_FILE} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, " \
                          f"session {pm.get('_session', '?')}; not collected in this run)"
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
        ms_per_step = elapsed / args.steps * 1e3
        step_s = elapsed / args.steps
        out = {
            "metric": "splatted-Gaussians/s (fwd+bwd) @1200x680; achieved HBM GB/s vs peak",
            "value": N_GAUSS * world / step_s,
            "unit": "Gaussians/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "office0 stand-in (seeded box room), 200k surfels, 1200x680, 1 view per GPU (rank r: "
                                   "view 0 mirrored through the room's symmetry planes = equal work per rank), "
                                   "step = activations + fwd + bwd (fed fixed random image gradients d_rgb, d_normal, "
                                   "d_depth: no loss head in the step) + gradient-row exchange (N>1) + Adam",
                       "gaussians": N_GAUSS, "image": [H, W], "views_per_gpu": 1, "visible": V,
                       "tile_instances": I, "parallelism": f"view-parallel dp{world}",
                       "overflow": bool(info["overflow"]) or bool(info["overflow_passes"]) or refused > 0,
                       "overflow_passes": int(info["overflow_passes"]), "profile_note": profile_note,
                       "binning": args.binning,
                       "launch": launch_mode,
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
                       "eager_ms_per_step": round(eager_elapsed / args.steps * 1e3, 4),
                       "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
                       "stage_mean_ms": {k: round(v, 4) for k, v in stage_mean_ms.items()},
                       "stage_timing": f"HIP events on the launch stream around every stage of {args.steps} eager steps "
                                       "run right after the timed region (medians"
                                       + ("; the un-pipelined five-launch form of the step: `preprocess` and "
                                          "`preprocess_bwd` are separate kernels there, one launch in the timed region"
                                          if getattr(one_step, "pipelined", False) else "")
                                       + "); an event pair reads ~3 us more than rocprofv3's kernel duration"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": sb[dom],
                         "all_stages_GBps": {k: (sb[k] / (stage_ms[k] * 1e-3) / 1e9 if stage_ms[k] > 0 else 0.0)
                                             for k in sb},
                         "whole_step_GBps": sum(sb.values()) / step_s / 1e9},
        }
        if valu is not None:
            out["roofline_valu"] = valu
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
                return imgs, grads

            out["cpu_baseline"], out["parity"] = cpu_baseline(
                raw_cpu, dict(tanx=tanx, tany=tany, bg=bg, view=cm["viewmatrix"][0], proj=cm["projmatrix"][0]), d_cpu, gpu_check)
        print(json.dumps(out))
    if dist_on:
        torch.distributed.barrier()  # rank 0 may still be in the CPU-baseline leg; leave together
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
