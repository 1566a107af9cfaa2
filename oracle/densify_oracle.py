"""CPU oracle for the map-growth path: depth smoothing, ``add_gaussians``, ``cal_mask``, voxel
down-sampling and ``prune``.

TEST INFRASTRUCTURE ONLY - see oracle/surfel_oracle.py: nothing under ``active-gs_amd/`` imports it.

What is restated, and how it is pinned:
  * ``add_gaussians`` / ``cal_mask`` / ``prune`` / ``voxel_downsample`` / ``depth2normal`` /
    ``normal2rotation`` / ``get_world_rays`` are Python in the reference
    (/root/reference/mapping/gaussian_map.py:234-246,294-489,
    /root/reference/utils/operations.py:172-219,464-500,526-569,603-625).  They are PINNED:
    tests/golden/make_golden.py runs the reference's own ``GaussianMap.add_gaussians`` and
    ``prune`` on seeded frames and tests/test_cpu_oracle.py compares this file with those outputs.
  * ``voxel_downsample`` keeps one RANDOM point per occupied 2 cm voxel (torch.randperm).  A random
    choice cannot be matched, so the fixture is generated with ``torch.randperm`` replaced by the
    identity permutation - a valid sample of the reference's behaviour, for which its index
    assignment keeps the HIGHEST index of every voxel on CPU.  That rule ("one point per voxel,
    the last in pixel order") is what this oracle and the HIP kernels implement.
  * ``get_smooth_depth`` (/root/reference/utils/operations.py:161-169) calls
    ``cv2.bilateralFilter(depth, 15, 0.5, 20)`` from opencv-python==4.6.0.66
    (/root/reference/envs/requirements.txt:23), which is NOT installed here: PARITY UNPINNED for
    this one function.  ``bilateral_depth`` restates the published filter (Tomasi & Manduchi 1998
    as implemented by OpenCV for CV_32F: circular 15x15 support, BORDER_REFLECT_101, weight
    exp(-r^2/(2 sigma_s^2)) * exp(-dv^2/(2 sigma_c^2))).  OpenCV evaluates the range kernel through
    a 4096-bin linearly interpolated table; this restatement uses expf directly (weights agree to
    ~1e-5).  The fixture generator plugs this function in for the absent cv2.
"""
from __future__ import annotations

import math

import numpy as np
import torch

VOXEL_SIZE = 0.02          # operations.py:603
BILATERAL_D, BILATERAL_SIGMA_COLOR, BILATERAL_SIGMA_SPACE = 15, 0.5, 20.0   # operations.py:164-166


def bilateral_filter(src: np.ndarray, d: int = BILATERAL_D, sigma_color: float = BILATERAL_SIGMA_COLOR,
                     sigma_space: float = BILATERAL_SIGMA_SPACE) -> np.ndarray:
    """cv2.bilateralFilter for one float32 channel (see the header for what is and is not pinned)."""
    src = np.asarray(src, dtype=np.float32)
    radius = d // 2
    h, w = src.shape
    pad = np.pad(src, radius, mode="reflect")          # numpy 'reflect' == BORDER_REFLECT_101
    cs, cc = np.float32(-0.5 / (sigma_space * sigma_space)), np.float32(-0.5 / (sigma_color * sigma_color))
    acc = np.zeros_like(src)
    wsum = np.zeros_like(src)
    for dy in range(-radius, radius + 1):
        for dx in range(-radius, radius + 1):
            r2 = dy * dy + dx * dx
            if math.sqrt(r2) > radius:
                continue
            v = pad[radius + dy:radius + dy + h, radius + dx:radius + dx + w]
            dv = v - src
            wgt = np.exp(np.float32(r2) * cs, dtype=np.float32) * np.exp(dv * dv * cc, dtype=np.float32)
            acc += v * wgt
            wsum += wgt
    return (acc / wsum).astype(np.float32)


def smooth_depth(depth: np.ndarray, tolerance: float = BILATERAL_SIGMA_COLOR) -> np.ndarray:
    """get_smooth_depth (operations.py:161-169): invalid (<0) pixels enter the filter as 0 and
    come out as -1."""
    depth = np.asarray(depth, dtype=np.float32)
    invalid = depth < 0.0
    work = depth.copy()
    work[invalid] = 0.0
    out = bilateral_filter(work, BILATERAL_D, tolerance, BILATERAL_SIGMA_SPACE)
    out[invalid] = -1.0
    return out


def depth_to_normal(depth: torch.Tensor, mask: torch.Tensor, fov) -> torch.Tensor:
    """depth2normal (operations.py:172-219) incl. its focal pairing K00 <- (fov[0], H), K11 <- (fov[1], W)."""
    _, H, W = depth.shape
    d = depth[0]
    m = mask[0].to(torch.float32)
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    k00 = H / (2 * math.tan(fov[0] / 2))
    k11 = W / (2 * math.tan(fov[1] / 2))
    px = (xs - 0.5 * W) * d / k00
    py = (ys - 0.5 * H) * d / k11
    cam = torch.stack([px, py, d], -1)                                  # (H,W,3)
    pp = torch.nn.functional.pad(cam.permute(2, 0, 1)[None], [1, 1, 1, 1], mode="replicate")[0].permute(1, 2, 0)
    mp = torch.nn.functional.pad(m[None, None], [1, 1, 1, 1], mode="replicate")[0, 0].bool().float()[..., None]
    pc = pp[1:-1, 1:-1] * mp[1:-1, 1:-1]
    pu = (pp[:-2, 1:-1] - pc) * mp[:-2, 1:-1]
    pl = (pp[1:-1, :-2] - pc) * mp[1:-1, :-2]
    pb = (pp[2:, 1:-1] - pc) * mp[2:, 1:-1]
    pr = (pp[1:-1, 2:] - pc) * mp[1:-1, 2:]
    n = torch.cross(pu, pl, dim=-1) + torch.cross(pr, pu, dim=-1) + torch.cross(pb, pr, dim=-1) + \
        torch.cross(pl, pb, dim=-1)
    n = torch.nn.functional.normalize(n, dim=-1)
    return (n * m[..., None]).permute(2, 0, 1)


def normal_to_rotation(z: torch.Tensor) -> torch.Tensor:
    """normal2rotation + rotmat2quaternion (operations.py:481-500,526-541) -> (n,4) wxyz."""
    z = z / z.norm(dim=1, keepdim=True)
    ref = torch.zeros_like(z)
    ref[:, 0] = 1.0
    par = z[:, 0].abs() > 0.99
    ref[par] = torch.tensor([0.0, 1.0, 0.0])
    x = ref - (ref * z).sum(1, keepdim=True) * z
    x = x / x.norm(dim=1, keepdim=True)
    y = torch.cross(z, x, dim=1)
    y = y / y.norm(dim=1, keepdim=True)
    R = torch.stack([x, y, z], -1)
    tr = R[:, 0, 0] + R[:, 1, 1] + R[:, 2, 2] + 1e-6
    r = torch.sqrt(1 + tr) / 2
    q = torch.stack([r, (R[:, 2, 1] - R[:, 1, 2]) / (4 * r), (R[:, 0, 2] - R[:, 2, 0]) / (4 * r),
                     (R[:, 1, 0] - R[:, 0, 1]) / (4 * r)], -1)
    return torch.nn.functional.normalize(q, dim=-1)


def candidates(rgb: torch.Tensor, depth: torch.Tensor, intrinsic: torch.Tensor, extrinsic: torch.Tensor,
               depth_smooth: torch.Tensor, pred: dict | None, error_thres: float) -> dict:
    """Everything add_gaussians (gaussian_map.py:294-400) derives per pixel, before the voxel filter:
    ``means`` (P,3), ``rotations`` (P,4), ``harmonics`` (P,3) and the boolean ``select`` mask."""
    _, H, W = rgb.shape
    P = H * W
    valid = (depth > 0.0).view(-1).clone()
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    uv1 = torch.stack([(xs.float() + 0.5) / W, (ys.float() + 0.5) / H, torch.ones(H, W)], -1).view(P, 3)
    dir_cam = uv1 @ intrinsic.inverse().T
    R, t = extrinsic[:3, :3], extrinsic[:3, 3]
    dir_w = dir_cam @ R.T
    pcd = t[None] + dir_w * depth.view(-1, 1)
    n_cam = depth_to_normal(depth_smooth, valid.view(1, H, W), (math.pi / 3, math.pi / 3)).permute(1, 2, 0).reshape(P, 3)
    valid &= (n_cam ** 2).sum(-1) > 0.0
    n_w = n_cam @ R.T
    normals = torch.zeros(P, 3)
    normals[:, 2] = 1.0
    normals[valid] = n_w[valid]
    cos = (torch.nn.functional.normalize(dir_w, dim=1) * normals).sum(-1)
    valid &= cos < -0.01
    rot = normal_to_rotation(normals)
    valid &= ~torch.any(rot.isnan(), dim=1)
    if pred is not None:                                              # cal_mask, gaussian_map.py:470-489
        err = ((rgb - pred["rgb"]) ** 2).mean(0)
        m = err > error_thres
        m |= pred["opacity"] < 0.5
        m |= (depth[0] - pred["depth"]) < -0.05 * depth[0]
        m = m.view(-1)
    else:
        m = torch.ones(P, dtype=torch.bool)
    return dict(means=pcd, rotations=rot, harmonics=rgb.permute(1, 2, 0).reshape(P, 3).clone(), select=m & valid,
                normals=normals)


def voxel_keys(points: torch.Tensor, voxel: float = VOXEL_SIZE) -> torch.Tensor:
    return torch.floor(points / voxel).long()


def voxel_select_last(points: torch.Tensor, select: torch.Tensor, voxel: float = VOXEL_SIZE) -> torch.Tensor:
    """One point per occupied voxel among ``select``: the highest pixel index (see the header)."""
    idx = torch.nonzero(select).flatten()
    keys = voxel_keys(points[idx], voxel)
    _, inv = torch.unique(keys, return_inverse=True, dim=0)
    last = torch.full((int(inv.max()) + 1 if inv.numel() else 0,), -1, dtype=torch.long)
    last.scatter_reduce_(0, inv, idx, reduce="amax")
    out = torch.zeros_like(select)
    out[last] = True
    return out


def add_gaussians(state: dict, frame: dict, pred: dict | None, error_thres: float) -> dict:
    """``state``: means, scales, rotations, opacities, harmonics (n,1,3), view_scores, view_supports,
    view_means.  Returns the grown state (gaussian_map.py:400-462)."""
    ds = torch.from_numpy(smooth_depth(frame["depth"][0].numpy()))[None]
    c = candidates(frame["rgb"], frame["depth"], frame["intrinsic"], frame["extrinsic"], ds, pred, error_thres)
    keep = voxel_select_last(c["means"], c["select"])
    k = int(keep.sum())
    new_scales = torch.zeros(k, 3)
    new_scales[:, 2] = -1e10
    out = dict(state)
    out["means"] = torch.cat([state["means"], c["means"][keep]])
    out["scales"] = torch.cat([state["scales"], new_scales])
    out["rotations"] = torch.cat([state["rotations"], c["rotations"][keep]])
    out["opacities"] = torch.cat([state["opacities"], torch.zeros(k)])
    out["harmonics"] = torch.cat([state["harmonics"], c["harmonics"][keep][:, None, :]])
    out["view_scores"] = torch.cat([state["view_scores"], torch.zeros(k)])
    out["view_supports"] = torch.cat([state["view_supports"], torch.zeros(k)])
    out["view_means"] = torch.cat([state["view_means"], torch.zeros(k, 3)])
    return out


def prune(state: dict, prune_mask: torch.Tensor) -> dict:
    """gaussian_map.py:234-246: also drops surfels whose activated opacity is < 0.1."""
    drop = prune_mask.bool() | (torch.sigmoid(state["opacities"]) < 0.1)
    return {k: (v[~drop] if torch.is_tensor(v) and v.shape[:1] == drop.shape else v) for k, v in state.items()}
