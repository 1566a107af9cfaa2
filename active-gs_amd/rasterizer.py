"""Drop-in replacement for ``diff_gaussian_rasterization_2d`` (the un-vendored CUDA
extension imported at /root/reference/utils/operations.py:22-25).

Same public names and call contract as the single call site operations.py:682-713:
``GaussianRasterizationSettings(**15 keyword fields)`` and
``GaussianRasterizer(settings)(means3D, means2D, opacities, confidences, shs,
colors_precomp, scales, rotations, cov3D_precomp)`` returning the 8-tuple
``(rgb, normal, depth, opacity, confidence, importance, count, radii)``.
The arithmetic runs in libags_raster.so (HIP, gfx950); there is no CPU fallback.
"""
from __future__ import annotations

from typing import NamedTuple

import torch
import torch.nn as nn

from . import raster_api as api

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
SH_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
         1.445305721320277, -0.5900435899266435)


def eval_sh(degree: int, sh: torch.Tensor, dirs: torch.Tensor) -> torch.Tensor:
    """Real spherical harmonics of degree 0..3 evaluated along unit directions: ``sh`` (N, >= (degree+1)^2, 3),
    ``dirs`` (N, 3) -> (N, 3), before the +0.5 offset.  The basis, its ordering and signs are the ones the reference tree
    itself evaluates in its viewer (/root/reference/visualization/gl_render/shaders/gau_vert.glsl:3-18,173-205; the
    3DGS convention).  Plain torch on whatever device the inputs live on: autograd carries the gradients to the
    coefficients and, through the direction, to the means - the reference's mapper never takes this branch (it passes
    ``colors_precomp``, operations.py:709), so there is no kernel for it."""
    if not 0 <= degree <= 3:
        raise ValueError("sh_degree must be 0..3")
    if sh.shape[1] < (degree + 1) ** 2:
        raise ValueError(f"degree {degree} needs {(degree + 1) ** 2} coefficients per channel, got {sh.shape[1]}")
    out = SH_C0 * sh[:, 0]
    if degree > 0:
        x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
        out = out - SH_C1 * y * sh[:, 1] + SH_C1 * z * sh[:, 2] - SH_C1 * x * sh[:, 3]
        if degree > 1:
            xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
            out = (out + SH_C2[0] * xy * sh[:, 4] + SH_C2[1] * yz * sh[:, 5] + SH_C2[2] * (2.0 * zz - xx - yy) * sh[:, 6]
                   + SH_C2[3] * xz * sh[:, 7] + SH_C2[4] * (xx - yy) * sh[:, 8])
            if degree > 2:
                out = (out + SH_C3[0] * y * (3.0 * xx - yy) * sh[:, 9] + SH_C3[1] * xy * z * sh[:, 10]
                       + SH_C3[2] * y * (4.0 * zz - xx - yy) * sh[:, 11]
                       + SH_C3[3] * z * (2.0 * zz - 3.0 * xx - 3.0 * yy) * sh[:, 12]
                       + SH_C3[4] * x * (4.0 * zz - xx - yy) * sh[:, 13] + SH_C3[5] * z * (xx - yy) * sh[:, 14]
                       + SH_C3[6] * x * (xx - 3.0 * yy) * sh[:, 15])
    return out


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    render_mask: torch.Tensor
    weight_thres: float
    debug: bool
    config: torch.Tensor


# Binning algorithm used by the drop-in module (raster_api.BIN_DIRECT | BIN_TILE_SORT | BIN_RADIX).
BINNING_MODE = api.BIN_DIRECT

# Tile-instance capacity remembered per (device, image size): the forward pass sizes its
# workspace from the last view's need and re-runs only when a view overflows it.
_capacity_hint: dict = {}

# Workspaces (the library's internal state of a view: projected records, keys, ranges, per-pixel blend state, gradient
# records - tens of MB) are pooled per (device, surfels, image size): a call takes one, and it goes back when the
# autograd graph that may still need it for the backward pass is freed (a forward under no_grad returns it at once).
# Reuse is safe without re-initialisation: the forward pass leaves its counters clean, and everything runs on the
# caller's stream in order.  The image / per-Gaussian OUTPUT tensors are fresh per call - the caller owns them.
_workspace_pool: dict = {}
_POOL_MAX_PER_KEY = 16


def _take_workspace(key, n, h, w, cap, dev):
    free = _workspace_pool.setdefault(key, [])
    while free:
        ws, ws_cap = free.pop()
        if ws_cap >= cap:
            return ws, ws_cap
    ws = torch.empty(api.workspace_bytes(n, h, w, cap), device=dev, dtype=torch.uint8)
    st = api.ForwardState(None, None, None, None, None, None, None, None, ws, int(cap), BINNING_MODE)
    api.init_workspace(st, n, h, w)
    return ws, int(cap)


def _give_workspace(key, ws, cap):
    free = _workspace_pool.setdefault(key, [])
    if len(free) < _POOL_MAX_PER_KEY:
        free.append((ws, cap))


def _config_flags(cfg_tensor):
    """config = [_, normalize_depth, perpix_depth, importance?, front_only?] (operations.py:699).  The reference
    builds it on the host and moves it to the GPU per call; reading it back costs a stream synchronisation, so the
    combinations are remembered by the tensor's storage and version: a tensor that is still alive and unmodified
    cannot have changed its five floats."""
    if cfg_tensor is None:
        return [1.0, 1.0, 1.0, 0.0, 0.0]
    cfg = [float(v) for v in cfg_tensor.detach().cpu().tolist()]
    while len(cfg) < 5:
        cfg.append(0.0)
    return cfg


def _camera_from_settings(s: GaussianRasterizationSettings, device) -> api.Camera:
    cfg = _config_flags(s.config)
    f32 = lambda t: t.detach().to(device=device, dtype=torch.float32).contiguous()
    mask = None
    if s.render_mask is not None and s.render_mask.numel() > 0:
        mask = f32(s.render_mask)
    return api.Camera(int(s.image_height), int(s.image_width), float(s.tanfovx), float(s.tanfovy),
                      f32(s.viewmatrix), f32(s.projmatrix), f32(s.bg), float(s.scale_modifier),
                      float(s.weight_thres), cfg[1] > 0, cfg[2] > 0, cfg[3] > 0, cfg[4] > 0, mask)


class _RasterizeSurfels(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, opacities, confidences, colors, scales, rotations, settings):
        dev = means3D.device
        if not means3D.is_cuda:
            raise RuntimeError("diff_gaussian_rasterization_2d (MI355X build): tensors must be on the GPU; "
                               "there is no CPU fallback")
        cam = _camera_from_settings(settings, dev)
        f32 = lambda t: t.detach().to(dtype=torch.float32).contiguous()
        g = api.Gaussians(f32(means3D), f32(scales), f32(rotations), f32(opacities).reshape(-1), f32(colors),
                          f32(confidences).reshape(-1))
        n, h, w = g.n, cam.image_height, cam.image_width
        key = (dev.index, h, w)
        pkey = (dev.index, n, h, w, BINNING_MODE)
        cap = max(_capacity_hint.get(key, 0), 1 << 16, 2 * n)
        while True:
            ws, ws_cap = _take_workspace(pkey, n, h, w, cap, dev)
            state = api.alloc_outputs(n, h, w, dev, ws, ws_cap, BINNING_MODE, stats=cam.want_stats)
            api.forward(cam, g, state)
            st = api.read_status(state)  # one 64-byte D2H, like upstream's num_rendered read-back
            if not st["overflow"]:
                break
            cap = int(st["needed"] * 1.25) + 1024      # too small: this workspace is dropped, a larger one is made
        # monotone: views of one loop differ in what they need, and a hint that followed the last view would make
        # the next one reject every pooled workspace sized for a lighter view
        _capacity_hint[key] = max(_capacity_hint.get(key, 0), int(st["needed"] * 1.5) + 1024, 1 << 16)
        if any(ctx.needs_input_grad):      # (all False under no_grad)
            import weakref
            weakref.finalize(ctx, _give_workspace, pkey, ws, ws_cap)   # back to the pool when the graph is freed
        else:
            _give_workspace(pkey, ws, ws_cap)
        # What the backward needs of the forward's OUTPUTS goes through save_for_backward: an output tensor kept as a
        # plain attribute of ctx would close a cycle (ctx -> tensor -> grad_fn = ctx) that only the cyclic garbage
        # collector breaks - the view's workspace (tens of MB) would outlive its graph by many iterations.
        ctx.save_for_backward(state.depth, state.opacity, state.radii)
        ctx.cam, ctx.g = cam, g
        ctx.ws, ctx.ws_cap = ws, ws_cap
        ctx.need_m2d = means2D.requires_grad
        ctx.opac_shape = opacities.shape
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(state.importance, state.count, state.radii)
        return (state.rgb, state.normal, state.depth, state.opacity, state.confidence, state.importance,
                state.count, state.radii)

    @staticmethod
    def backward(ctx, d_rgb, d_normal, d_depth, d_opacity, d_conf, *_unused):
        cam, g = ctx.cam, ctx.g
        depth, opacity, radii = ctx.saved_tensors
        # (the blend backward reads the forward's depth and opacity images, its per-pixel state in the workspace and radii)
        state = api.ForwardState(None, None, depth, opacity, None, None, None, radii, ctx.ws, ctx.ws_cap, BINNING_MODE)
        c = lambda t: None if t is None else t.to(dtype=torch.float32).contiguous()
        grads = api.alloc_grads(g.n, g.means3D.device, with_means2d=ctx.need_m2d)
        api.backward(cam, g, state, c(d_rgb), c(d_normal), c(d_depth), c(d_opacity), c(d_conf), grads)
        return (grads.means3D, grads.means2D, grads.opacities.reshape(ctx.opac_shape), None, grads.colors,
                grads.scales, grads.rotations, None)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings

    def forward(self, means3D, means2D, opacities, confidences, shs=None, colors_precomp=None, scales=None,
                rotations=None, cov3D_precomp=None):
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or (
                (scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        if cov3D_precomp is not None:
            raise NotImplementedError("surfels need scale+rotation (the normal is R[:,2]); cov3D_precomp is unsupported")
        if shs is not None:
            # view-dependent colour from SH coefficients (N, K, 3): direction = mean - camera centre, normalised; the
            # result + 0.5 is clamped at 0 (the clamp passes no gradient where it is active, as upstream's `clamped` flag)
            deg = int(self.raster_settings.sh_degree)
            campos = self.raster_settings.campos.to(means3D.device, means3D.dtype).reshape(1, 3)
            dirs = means3D - campos
            dirs = dirs / dirs.norm(dim=1, keepdim=True).clamp_min(1e-20)
            colors_precomp = torch.clamp_min(eval_sh(deg, shs, dirs) + 0.5, 0.0)
        return _RasterizeSurfels.apply(means3D, means2D, opacities, confidences, colors_precomp, scales, rotations,
                                       self.raster_settings)
