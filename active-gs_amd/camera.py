"""Camera conventions at the rasterizer boundary.

Host-side mirror of what ``GaussianRenderer.__init__`` feeds the rasterizer
(/root/reference/utils/operations.py:724-778): normalised intrinsics -> fov
(``get_fov``, :628-642), symmetric-frustum projection with z in [0,1] and w = z
(``get_projection_matrix``, :572-600), ``viewmatrix = inverse(c2w)^T`` and
``projmatrix = viewmatrix @ P^T`` — i.e. row-vector matrices, translation in the last
row (``p_hom = [x y z 1] @ projmatrix``).  Pure torch, no GPU needed.
"""
from __future__ import annotations

import torch


def fov_from_intrinsics(K: torch.Tensor) -> torch.Tensor:
    """(B,3,3) normalised intrinsics -> (B,2) [fov_x, fov_y]: the angle between the
    rays through the mid-points of opposite image edges (operations.py:628-642)."""
    Kinv = torch.linalg.inv_ex(K).inverse     # (inv() reads an error flag back from the device: a stream synchronisation)

    def ray(v):
        # Kinv @ v without bringing v to the device (four small host-to-device copies per batch otherwise)
        d = Kinv[:, :, 0] * v[0] + Kinv[:, :, 1] * v[1] + Kinv[:, :, 2] * v[2]
        return d / d.norm(dim=-1, keepdim=True)

    fov_x = (ray([0.0, 0.5, 1.0]) * ray([1.0, 0.5, 1.0])).sum(-1).acos()
    fov_y = (ray([0.5, 0.0, 1.0]) * ray([0.5, 1.0, 1.0])).sum(-1).acos()
    return torch.stack([fov_x, fov_y], -1)


def projection_matrix(near: float, far: float, fov_x: torch.Tensor, fov_y: torch.Tensor) -> torch.Tensor:
    """(B,) fovs -> (B,4,4) column-vector projection (operations.py:572-600)."""
    tx = (0.5 * fov_x).tan()
    ty = (0.5 * fov_y).tan()
    B = fov_x.shape[0]
    P = torch.zeros(B, 4, 4, dtype=torch.float32, device=fov_x.device)
    P[:, 0, 0] = 1.0 / tx
    P[:, 1, 1] = 1.0 / ty
    P[:, 3, 2] = 1.0
    P[:, 2, 2] = far / (far - near)
    P[:, 2, 3] = -(far * near) / (far - near)
    return P


def camera_matrices(c2w: torch.Tensor, K: torch.Tensor, near: float, far: float):
    """(B,4,4) OpenCV camera-to-world, (B,3,3) normalised intrinsics ->
    dict(viewmatrix (B,4,4), projmatrix (B,4,4), tanfov (B,2), campos (B,3)) in the
    row-vector convention the rasterizer takes (operations.py:749-762)."""
    fov = fov_from_intrinsics(K)
    P = projection_matrix(near, far, fov[:, 0], fov[:, 1])
    view = torch.linalg.inv_ex(c2w).inverse.transpose(1, 2).contiguous()
    proj = (view @ P.transpose(1, 2)).contiguous()
    return dict(viewmatrix=view, projmatrix=proj, tanfov=(0.5 * fov).tan(), campos=c2w[:, :3, 3].contiguous())
