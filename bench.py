#!/usr/bin/env python3
"""bench.py — splatted-Gaussians/s (fwd+bwd) @1200x680 on MI355X (BASELINE.json metric).

One step = one optimisation step of the hot path on resident data: activations ->
forward rasterization -> backward rasterization (with fixed synthetic image gradients) ->
activation backward -> [all-reduce of the gradient slab when N>1] -> fused Adam.
Workload at every N: BASELINE.json configs[1] per GPU (office0 stand-in, 200k surfels,
1200x680, one view per rank, weak scaling).  Prints ONE JSON line on rank 0.
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


def stage_bytes(N, V, I, P, T, rows=None):
    """ALGORITHMIC bytes per launch of each stage: every logical
    array moved once.  ``rows``: member rows of the sticky row set when the per-Gaussian backward
    runs in its row-set form with the Adam step fused in (single rank), else None."""
    if rows is not None:
        # per member row: id 4 + radius 4 + params 44 + gradient record read 64 / re-zeroed 64 (visible
        # rows) + gradient row 56 + Adam m, v read+write 224 + parameters written 56
        pbwd = 8 * rows + 128 * V + (44 + 56 + 224 + 56) * rows
    else:
        pbwd = 44 * N + 128 * V + 68 * N                       # means/scales/rot/radii; dgeom read + re-zero; 5 grads
    return {
        "preprocess": 60 * N + 8 * N + 136 * V,                # inputs; radii+tiles; geom 64 + rect 8 + zeroed dgeom 64
        "binning": 8 * N + 12 * I + 24 * I + 8 * I + 8 * T,    # tile counts/scan; key write; sort r+w once; ranges
        "render_fwd": 8 * T + 68 * I + 44 * P,                  # ranges; id 4 + record 64; 9 ch + T + n_contrib
        "render_bwd": 8 * T + 68 * I + 52 * P + 64 * V,         # + 9 grads, depth, opac, T, n; accumulate dgeom
        "preprocess_bwd": pbwd,
    }


STAGE_IDS = {"preprocess": 0, "binning": 1, "render_fwd": 2, "render_bwd": 3, "preprocess_bwd": 4}


def cpu_baseline(raw_cpu, cam_cpu, budget_s=15.0, max_threads=16):
    """The oracle (oracle/surfel_oracle.py, PyTorch CPU, fp32) timed on this host: the full
    per-Gaussian stage + binning of the same view, then fwd+bwd of strided batches of tiles
    until ~budget_s of CPU time is spent, extrapolated to all non-empty tiles."""
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
    while done < len(order) and t_tiles < budget_s:
        sample = order[done:done + batch]
        t0 = time.perf_counter()
        R = render_tiles(G, so, ranges, S, tiles=sample)
        (R["rgb"].sum() + R["depth"].sum() + R["normal"].sum()).backward(retain_graph=True)
        t_tiles += time.perf_counter() - t0
        done += len(sample)
    est = t_pre + t_tiles * len(nonempty) / max(done, 1)
    return {"value": N_GAUSS / est, "unit": "Gaussians/s", "cores": cores, "kind": "port",
            "sample": f"oracle (PyTorch CPU fp32, {cores} threads): full preprocess+binning of the 200k-surfel "
                      f"1200x680 view ({t_pre:.1f}s) + fwd+bwd of {done} of {len(nonempty)} non-empty tiles "
                      f"({t_tiles:.1f}s), extrapolated to all tiles"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--binning", choices=["tile_sort", "radix"], default="tile_sort")
    ap.add_argument("--eager", action="store_true", help="time eager launches instead of hipGraph replay")
    ap.add_argument("--graph-steps", type=int, default=int(os.environ.get("AGS_BENCH_GRAPH_STEPS", "25")),
                    help="optimisation steps recorded per hipGraph (single GPU); K steps = K/this replays")
    args = ap.parse_args()
    if os.environ.get("AGS_BENCH_WATCHDOG"):   # debugging aid: dump every thread's stack and exit after N s
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["AGS_BENCH_WATCHDOG"]), exit=True)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
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
    trainer = SurfelTrainer(raw, binning_mode=api.BIN_RADIX if args.binning == "radix" else api.BIN_TILE_SORT)

    # size the workspace from one probing forward (outside the timed region)
    g = trainer.gaussians()
    probe = api.alloc_state(N_GAUSS, H, W, 16_000_000, dev)
    api.forward(cam, g, probe)
    info = api.read_status(probe)
    assert not info["overflow"], info
    I, V = info["num_instances"], info["num_visible"]
    del probe
    cap = int(I * 1.3) + 4096

    gen = torch.Generator().manual_seed(1234 + rank)
    P = H * W
    scale = 1.0 / (P * world)
    d_img = [(torch.randn(c, H, W, generator=gen) * scale).to(dev) for c in (3, 3, 1)]
    grads_fn = lambda v, st: (d_img[0], d_img[1], d_img[2], None, None)

    def eager_step():
        trainer.step([cam], grads_fn, cap, world_views=world, device_clock=True)

    for _ in range(3):
        eager_step()  # creates every buffer before capture
    torch.cuda.synchronize()
    launch_mode = "eager" if args.eager else "hipGraph replay"
    one_step = eager_step
    many_steps, per_replay = None, 1
    if not args.eager:
        try:
            one_step = trainer.capture([cam], grads_fn, cap)
            in_graph = getattr(one_step, "collective_in_graph", False)
            if dist_on:
                launch_mode = ("hipGraph replay, gradient exchange recorded in the graph" if in_graph
                               else "hipGraph replay: graph | collective | graph")
            if args.graph_steps > 1 and (not dist_on or in_graph):
                many_steps = trainer.capture([cam], grads_fn, cap, repeat=args.graph_steps)
                per_replay = many_steps.steps
                launch_mode += f", {per_replay} steps per graph"
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

    # per-stage kernel time: the same K steps launched eagerly with library-owned HIP events
    # around every stage (events cannot be timed inside a replayed graph)
    _lib.check(lib.ags_profile_enable(args.steps), "ags_profile_enable")
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        eager_step()
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
                        "overflow": x.overflowed()}
        else:
            exchange = {"kind": "all-reduce of the dense gradient slab", "bytes_per_rank": 4 * trainer.slab.flat.numel()}
        sb = stage_bytes(N_GAUSS, V, I, P, T, rows)
        dom = max(stage_ms, key=lambda k: stage_ms[k])
        ach = sb[dom] / (stage_ms[dom] * 1e-3) / 1e9 if stage_ms[dom] > 0 else 0.0
        traffic = None
        pmc_path = os.path.join(ROOT, "profiles", "pmc_hbm_bytes.json")
        if os.path.exists(pmc_path):
            try:
                traffic = json.load(open(pmc_path)).get(dom, {}).get("traffic")
            except Exception:
                traffic = None
        ms_per_step = elapsed / args.steps * 1e3
        out = {
            "metric": "splatted-Gaussians/s (fwd+bwd) @1200x680; achieved HBM GB/s vs peak",
            "value": N_GAUSS * world / (elapsed / args.steps),
            "unit": "Gaussians/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "office0 stand-in (seeded box room), 200k surfels, 1200x680, 1 view per GPU (rank r: "
                                   "view 0 mirrored through the room's symmetry planes = equal work per rank), "
                                   "step = activations + fwd + bwd + grad all-reduce (N>1) + Adam",
                       "gaussians": N_GAUSS, "image": [H, W], "views_per_gpu": 1, "visible": V,
                       "tile_instances": I, "parallelism": f"view-parallel dp{world}",
                       "overflow": bool(info["overflow"]), "binning": args.binning,
                       "launch": launch_mode,
                       "optimizer": ("row-set Adam fused into the per-Gaussian backward (exact: untouched rows "
                                     f"have zero gradient and moments), {rows} member rows") if (rows is not None and not dist_on)
                                    else ("row-set Adam over the union of the ranks' member rows" if rows is not None
                                          else "dense fused Adam kernel after the gradient all-reduce"),
                       "exchange": exchange,
                       "host_enqueue_ms_per_step": round(enqueue_s / args.steps * 1e3, 4),
                       "eager_ms_per_step": round(eager_elapsed / args.steps * 1e3, 4),
                       "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
                       "stage_mean_ms": {k: round(v, 4) for k, v in stage_mean_ms.items()}},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": sb[dom],
                         "all_stages_GBps": {k: (sb[k] / (stage_ms[k] * 1e-3) / 1e9 if stage_ms[k] > 0 else 0.0)
                                             for k in sb}},
        }
        if not args.no_cpu_baseline and world == 1:   # the CPU leg is timed at N=1 only (contract)
            out["cpu_baseline"] = cpu_baseline(raw_cpu, dict(tanx=tanx, tany=tany, bg=bg, view=cm["viewmatrix"][0],
                                                             proj=cm["projmatrix"][0]))
        print(json.dumps(out))
    if dist_on:
        torch.distributed.barrier()  # rank 0 may still be in the CPU-baseline leg; leave together
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
