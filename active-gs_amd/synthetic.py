"""Seeded synthetic stand-ins for the Replica scenes.

Replica is not available offline (/root/reference/data holds only download
scripts), so every benchmark / parity config uses a box room whose inner faces carry
surfels, with the parameterisation of /root/reference/mapping/gaussian_map.py:
raw ``_scales`` (z = -1e10, :373), raw ``_opacities``, ``_rotations`` from the face
normal (``normal2rotation``, operations.py:481-500), ``_harmonics`` (N,1,3).
"""
from __future__ import annotations

import math

import torch

ROOMS = {"office0": (5.0, 4.0, 3.0), "room0": (8.0, 5.0, 3.0)}
SCALE_FACTOR = 0.01  # /root/reference/config/mapper/incremental.yaml:15


def _quat_from_frame(x, y, z):
    """Rotation with columns (x,y,z) -> (w,x,y,z), w >= 0, branch on the largest
    diagonal term for conditioning."""
    m00, m10, m20 = x.unbind(-1)
    m01, m11, m21 = y.unbind(-1)
    m02, m12, m22 = z.unbind(-1)
    tr = m00 + m11 + m22
    q = torch.zeros(x.shape[0], 4, dtype=x.dtype)
    c0 = tr > 0
    c1 = (~c0) & (m00 >= m11) & (m00 >= m22)
    c2 = (~c0) & (~c1) & (m11 >= m22)
    c3 = ~(c0 | c1 | c2)
    s = torch.sqrt(torch.clamp(tr + 1.0, min=1e-12)) * 2
    q[c0] = torch.stack([0.25 * s, (m21 - m12) / s, (m02 - m20) / s, (m10 - m01) / s], -1)[c0]
    s = torch.sqrt(torch.clamp(1.0 + m00 - m11 - m22, min=1e-12)) * 2
    q[c1] = torch.stack([(m21 - m12) / s, 0.25 * s, (m01 + m10) / s, (m02 + m20) / s], -1)[c1]
    s = torch.sqrt(torch.clamp(1.0 + m11 - m00 - m22, min=1e-12)) * 2
    q[c2] = torch.stack([(m02 - m20) / s, (m01 + m10) / s, 0.25 * s, (m12 + m21) / s], -1)[c2]
    s = torch.sqrt(torch.clamp(1.0 + m22 - m00 - m11, min=1e-12)) * 2
    q[c3] = torch.stack([(m10 - m01) / s, (m02 + m20) / s, (m12 + m21) / s, 0.25 * s], -1)[c3]
    q = torch.where(q[:, :1] < 0, -q, q)
    return torch.nn.functional.normalize(q, dim=-1)


def rotation_from_normal(n: torch.Tensor) -> torch.Tensor:
    """Unit normals (N,3) -> quaternions whose 3rd rotation column is the normal."""
    z = torch.nn.functional.normalize(n, dim=-1)
    ref = torch.zeros_like(z)
    ref[:, 0] = 1.0
    par = z[:, 0].abs() > 0.99
    ref[par] = torch.tensor([0.0, 1.0, 0.0], dtype=z.dtype)
    x = torch.nn.functional.normalize(ref - (ref * z).sum(-1, keepdim=True) * z, dim=-1)
    y = torch.nn.functional.normalize(torch.cross(z, x, dim=-1), dim=-1)
    return _quat_from_frame(x, y, z)


def make_room_scene(n: int, room: str = "office0", seed: int = 0):
    """Raw (pre-activation) map parameters for ``n`` surfels on the six inner faces of
    the room box, plus per-Gaussian confidences.  Room is centred at the origin, z up."""
    g = torch.Generator().manual_seed(seed)
    lx, ly, lz = ROOMS[room]
    areas = torch.tensor([ly * lz, ly * lz, lx * lz, lx * lz, lx * ly, lx * ly])
    face = torch.multinomial(areas / areas.sum(), n, replacement=True, generator=g)
    u = torch.rand(n, 2, generator=g)
    half = torch.tensor([lx, ly, lz]) / 2
    pos = torch.zeros(n, 3)
    nrm = torch.zeros(n, 3)
    for f in range(6):
        axis, sign = f // 2, (-1.0 if f % 2 == 0 else 1.0)
        m = face == f
        others = [a for a in range(3) if a != axis]
        pos[m, axis] = sign * half[axis]
        pos[m, others[0]] = (u[m, 0] * 2 - 1) * half[others[0]]
        pos[m, others[1]] = (u[m, 1] * 2 - 1) * half[others[1]]
        nrm[m, axis] = -sign
    nrm = torch.nn.functional.normalize(nrm + 0.05 * torch.randn(n, 3, generator=g), dim=-1)
    scales = torch.empty(n, 3)
    lo, hi = math.log(0.5), math.log(3.0)
    scales[:, :2] = lo + (hi - lo) * torch.rand(n, 2, generator=g)
    scales[:, 2] = -1e10
    return dict(
        means=pos.contiguous(),
        scales=scales,
        rotations=rotation_from_normal(nrm),
        opacities=1.5 * torch.randn(n, generator=g),
        harmonics=torch.rand(n, 1, 3, generator=g),
        confidences=torch.rand(n, generator=g),
    )


def activate(raw: dict):
    """Activations of /root/reference/mapping/gaussian_map.py:529-549."""
    return dict(
        means=raw["means"],
        scales=torch.clamp(SCALE_FACTOR * torch.exp(raw["scales"]), min=0, max=0.05),
        rotations=torch.nn.functional.normalize(raw["rotations"], dim=-1),
        opacities=torch.sigmoid(raw["opacities"]),
        colors=raw["harmonics"][:, 0, :],
        confidences=raw["confidences"],
    )


def make_camera(view: int, height: int, width: int, focal_px: float | None = None, seed_base: int = 1,
                room: str = "office0", mirror: int = 0):
    """OpenCV c2w (4,4) and normalised intrinsics (3,3) for training view ``view``:
    position = room centre + U(-1,1) m in x,y (±0.3 m in z), yaw uniform, pitch 0.
    ``mirror`` (bits 0/1/2 = x/y/z): the same pose reflected through the room's symmetry planes - the
    box rooms of ``make_room_scene`` are statistically mirror-symmetric, so the eight mirrors of a
    view carry the same workload while looking at different surfels (bench.py's weak scaling)."""
    g = torch.Generator().manual_seed(seed_base + view)
    r = torch.rand(4, generator=g)
    sx, sy, sz = (-1.0 if mirror & 1 else 1.0), (-1.0 if mirror & 2 else 1.0), (-1.0 if mirror & 4 else 1.0)
    pos = torch.tensor([sx * (r[0] * 2 - 1).item(), sy * (r[1] * 2 - 1).item(), sz * (r[2] * 0.6 - 0.3).item()])
    yaw = (r[3] * 2 * math.pi).item()
    fwd = torch.tensor([sx * math.cos(yaw), sy * math.sin(yaw), 0.0])
    down = torch.tensor([0.0, 0.0, -1.0])
    right = torch.linalg.cross(down, fwd)
    c2w = torch.eye(4)
    c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = right, down, fwd, pos
    if focal_px is None:
        focal_px = 0.5 * width  # 90 deg horizontal, as 600 px at 1200x680
    K = torch.tensor([[focal_px / width, 0, 0.5], [0, focal_px / height, 0.5], [0, 0, 1.0]], dtype=torch.float32)
    return c2w, K


def make_keyframes(count: int, height: int, width: int, device, gt_surfels: int = 400_000, room: str = "office0",
                   host_pose: bool = True):
    """``count`` RGB-D keyframes of the room stand-in as the simulator would hand them to ``GaussianMap.update``
    (/root/reference/mapping/mapper.py:95-101: dict of ``rgb (3,H,W)``, ``depth (1,H,W)``, ``extrinsic (4,4)`` camera-to-world,
    ``intrinsic (3,3)`` normalised, ``depth_range (2,)``), rendered from a dense, opaque ground-truth surfel room with this
    library's own rasterizer (Replica / habitat are not available offline).  GPU only.
    ``host_pose``: the frames also carry the pose as the simulator made it - on the HOST (``extrinsic_host``, ``intrinsic_host``,
    ``depth_range_host``; /root/reference/mapping/mapper.py:94 has exactly these before line 95 moves the dict to the device):
    the map then needs no read-back of the pose per keyframe."""
    from . import raster_api as api
    from .camera import camera_matrices
    dev = torch.device(device)
    gt = {k: v.to(dev) for k, v in make_room_scene(gt_surfels, room=room, seed=0).items()}
    gt["scales"][:, :2] += 0.6                       # dense coverage: the ground truth is a closed room
    gt["opacities"] += 4.0
    a = activate(gt)
    g = api.Gaussians(a["means"], a["scales"], a["rotations"], a["opacities"], gt["harmonics"].view(-1, 3).contiguous(),
                      a["confidences"])
    st = api.alloc_state(gt_surfels, height, width, 1 << 24, dev)
    frames = []
    for v in range(count):
        c2w, K = make_camera(v, height, width, room=room)
        cm = camera_matrices(c2w[None].to(dev), K[None].to(dev), 0.001, 10.0)
        tan = cm["tanfov"][0].cpu()
        cam = api.Camera(height, width, float(tan[0]), float(tan[1]), cm["viewmatrix"][0].contiguous(),
                         cm["projmatrix"][0].contiguous(), torch.zeros(4, device=dev))
        api.forward(cam, g, st)
        if api.read_status(st)["overflow"]:
            raise RuntimeError("make_keyframes: the ground-truth render outgrew its workspace")
        depth = torch.where(st.opacity > 0.5, st.depth, torch.zeros_like(st.depth))
        frames.append(dict(rgb=st.rgb.clone().clamp(0, 1), depth=depth.clone(), extrinsic=c2w.to(dev),
                           intrinsic=K.to(dev), depth_range=torch.tensor([0.001, 10.0], device=dev)))
        if host_pose:
            frames[-1].update(extrinsic_host=c2w.clone(), intrinsic_host=K.clone(), depth_range_host=torch.tensor([0.001, 10.0]))
    return frames


def mapper_cfg(optimization_steps: int = 10, draw: str = "device"):
    """config/mapper/incremental.yaml:12-32 (``cfg.gaussian_map`` of the reference's mapper) as an attribute-style object."""
    from types import SimpleNamespace as NS
    return NS(bound=[0.001, 10.0], background=[0.0, 0.0, 0.0, 0.0], sparse_ratio=0.1, error_thres=0.25, scale_factor=0.01,
              optimization_steps=optimization_steps, prune_interval=5, use_view_distribution=True,
              sampler=NS(sampler_type="weighted", batch_size=8, active_size=3, draw=draw),
              optimizer=NS(mean_lr=0.0005, rotation_lr=0.0005, opacity_lr=0.01, scale_lr=0.01, harmonic_lr=0.0001))


def run_mapper_loop(frames, steps: int = 10, draw: str = "device", warmup_frames: int = 2, split: bool = False,
                    phases: bool = False):
    """BASELINE.json configuration 3: what /root/reference/mapping/mapper.py:98-104 does per keyframe -
    ``gaussian_map.update(dataframe)`` - from an EMPTY map over ``frames``, through the drop-in ``GaussianMap`` class.
    ``warmup_frames`` keyframes first go through a scratch map so that every kernel module is loaded before the timed
    loop.  ``split``: synchronise around growth and training to report them separately (two more waits per keyframe).
    ``phases``: a HIP event and the host clock at every phase boundary of the loop (FusedMapTrainer.phase_hook; ~7 event
    records per keyframe) -> ``phases``: per phase the GPU-timeline time between its marks (idle included) and the host
    time to enqueue it, from THIS run.  A phase whose GPU time is about its host time is host-bound (the GPU waits for
    launches); the iterations' GPU time is kernel time (their host time is a fifth of it).
    -> dict(seconds, iterations, ms_per_iteration, final_surfels, ...)."""
    import time
    from .gaussian_map import GaussianMap
    dev = frames[0]["rgb"].device
    if warmup_frames:
        warm = GaussianMap(mapper_cfg(steps, draw), dev)
        for f in frames[:warmup_frames]:
            warm.update(f)
        del warm
    torch.cuda.synchronize(dev)
    gm = GaussianMap(mapper_cfg(steps, draw), dev)
    marks = []
    if phases:
        def mark(label):
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            marks.append((label, e, time.perf_counter()))
        gm._fused().phase_hook = mark
    ms0 = torch.cuda.memory_stats(dev)
    t_grow = t_train = 0.0
    sizes = []
    t0 = time.perf_counter()
    for f in frames:
        if split:
            torch.cuda.synchronize(dev); a = time.perf_counter()
            gm.add_gaussians(f)
            torch.cuda.synchronize(dev); b = time.perf_counter()
            gm.train()
            torch.cuda.synchronize(dev); c = time.perf_counter()
            t_grow += b - a; t_train += c - b
        else:
            gm.update(f)
        sizes.append(gm.num_gaussians)       # (the row count: a read of the map itself would settle the call first)
    gm.settle()                              # the last call's pending workspace check belongs to the loop
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    iters = len(frames) * steps
    h, w = frames[0]["rgb"].shape[-2:]
    tr = gm._trainer
    out = dict(workload=f"mapper loop through GaussianMap.update(): {len(frames)} keyframes x {steps} iterations @{h}x{w}, batch 8 with "
                        f"3 active frames, prune every 5th keyframe, from an empty map",
               frame_sampler=draw, iterations=iters, seconds=round(dt, 4), ms_per_iteration=round(1e3 * dt / iters, 4),
               final_surfels=sizes[-1], surfels_after_10=sizes[min(9, len(sizes) - 1)],
               mean_frame_error=round(float(gm.training_performance.mean()), 5), last_loss=round(tr.last_losses[-1], 5),
               device_mallocs=int(torch.cuda.memory_stats(dev)["num_device_alloc"] - ms0["num_device_alloc"]),
               overflow_retries=int(getattr(tr, "overflow_retries", 0)))
    if phases and len(marks) > 1:
        tr.phase_hook = None
        acc = {}
        for (la, ea, ta), (lb, eb, tb) in zip(marks[:-1], marks[1:]):
            a = acc.setdefault(la, dict(gpu_timeline_ms=0.0, host_enqueue_ms=0.0, times=0))
            a["gpu_timeline_ms"] += ea.elapsed_time(eb); a["host_enqueue_ms"] += (tb - ta) * 1e3; a["times"] += 1
        for a in acc.values():
            a["gpu_timeline_ms"], a["host_enqueue_ms"] = round(a["gpu_timeline_ms"], 2), round(a["host_enqueue_ms"], 2)
        out["phases"] = acc
        it_ms = acc.get("iterations", {}).get("gpu_timeline_ms", 0.0)
        out["iterations_gpu_ms"] = it_ms
        out["gpu_bound_frac"] = round(it_ms / (dt * 1e3), 4)      # share of THIS run's wall time the GPU-bound phase covers
    if split:
        out.update(grow_ms_per_keyframe=round(1e3 * t_grow / len(frames), 3), train_ms_per_keyframe=round(1e3 * t_train / len(frames), 3))
    return out
