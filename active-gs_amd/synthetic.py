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
