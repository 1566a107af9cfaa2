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

import os
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _lib, raster_api as api

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


# ---- the module's host side is native code: csrc/torch_binding.cpp (autograd node, workspace pool, deferred workspace
# checks) over the C ABI of libags_raster.so.  This file keeps the public names, the argument checks the CUDA extension
# makes in Python, and the spherical-harmonics branch.
_binding = None


DROPIN_STATUS = "always"     # or "deferred"; read when the binding is first loaded (set_option("always_check", ..) afterwards)


def _load_binding():
    """lib/ags_torch_binding.so (built by active_gs_amd.build.build_torch_binding); raises if it is missing."""
    global _binding
    if _binding is not None:
        return _binding
    import importlib.util
    from . import build as _build
    if not os.path.exists(_build.BINDING):
        raise RuntimeError(f"{_build.BINDING} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`. "
                           "There is no Python or CPU fallback for the rasterizer module.")
    spec = importlib.util.spec_from_file_location(_build.BINDING_NAME, _build.BINDING)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    _lib.load()                                  # the ctypes handle of the same file: fails loudly if the library is absent
    mod.init(_lib.library_path())
    # Default: every call reads its status block back before it returns and repairs an outgrown workspace on the spot
    # (one stream synchronisation per view - what the CUDA extension's num_rendered read-back costs): an unmodified
    # caller never sees truncated tile lists.  DROPIN_STATUS = "deferred" (set before the first call): checks one call late, for loops that call
    # check_overflow() every iteration (see torch_binding.cpp); deferred_status() does the same for one block of code.
    mode = DROPIN_STATUS
    if mode not in ("always", "deferred"):
        raise ValueError(f"DROPIN_STATUS={mode!r}: 'always' (default) or 'deferred'")
    mod.set_option("always_check", 0.0 if mode == "deferred" else 1.0)
    # the library reads no environment variable: this binding hands it the process's kernel selection (AgsTuning)
    t = _lib.default_tuning()
    mod.set_tuning(t.bwd_reduce, t.render_slots, t.cull_first_min_n, t.tile_sort_no_wave, t.bucket_no_scan)
    _binding = mod
    return mod


def set_option(name: str, value) -> None:
    """binning_mode (raster_api.BIN_*), always_check (0/1), headroom, min_headroom, skew_factor, direct_budget_bytes,
    max_pending - see csrc/torch_binding.cpp: Options."""
    _load_binding().set_option(name, float(value))


def get_option(name: str) -> float:
    return _load_binding().get_option(name)


def counters() -> dict:
    """forward_calls, status_syncs (calls that read the status block back), deferred_checks, overflows, mode_switches."""
    return dict(_load_binding().counters())


def state() -> dict:
    need, modes, pooled = _load_binding().state()
    return dict(need_seen={tuple(r[:4]): r[4] for r in need}, mode_for={tuple(r[:3]): r[3] for r in modes},
                pooled={tuple(r[:4]): dict(count=r[4], bytes=r[5]) for r in pooled})


def reset_state() -> None:
    """Forget the sizes and binning modes learnt so far and drop the pooled workspaces."""
    _load_binding().reset_state()


def check_overflow() -> None:
    """Deferred checks only (AGS_DROPIN_STATUS=deferred / set_option("always_check", 0)): wait for the status copies of
    all forward passes issued so far and raise if one of them outgrew its workspace.  A training loop calls this where
    it synchronises anyway (e.g. once per iteration); without it the NEXT module call raises.  With the default
    (every call checked and repaired before it returns) there is never anything to report."""
    _load_binding().check_overflow()


def set_tuning(tuning: "_lib.AgsTuning") -> None:
    """The kernel selection the module hands to the library with every workspace (default: _lib.default_tuning())."""
    _load_binding().set_tuning(tuning.bwd_reduce, tuning.render_slots, tuning.cull_first_min_n, tuning.tile_sort_no_wave,
                               tuning.bucket_no_scan)


import threading as _threading

_tls = _threading.local()


class deferred_status:
    """``with deferred_status() as d: ... module calls ...; ok = d.settle()`` - inside the block (this thread) the
    module's calls do not wait for their status blocks; ``d.settle()`` waits once for all of them and returns True when
    every pass fitted its workspace.  False: at least one pass of the block was truncated - the sizes have been raised,
    REPEAT the block's passes (facade.SurfelRenderer does exactly that around its view loops: one wait per batch of views
    instead of one per view, and no exception reaches its caller).  A block that ends without having settled is settled
    then, and raises if a pass was truncated."""

    def __enter__(self):
        self._outer = getattr(_tls, "deferred", False)
        _tls.deferred = True
        self.reports, self._settled = [], False
        return self

    def settle(self) -> bool:
        self.reports = list(_load_binding().settle())
        self._settled = True
        return not self.reports

    def __exit__(self, exc_type, exc, tb):
        _tls.deferred = self._outer
        if exc_type is None and not self._settled and not self.settle():
            raise RuntimeError("diff_gaussian_rasterization_2d: " + "; ".join(self.reports) + ": the tile lists of that pass "
                               "were truncated - repeat the block's passes (the workspace size has been raised)")
        return False


_EMPTY = {}


def _empty(dev):
    t = _EMPTY.get(dev)
    if t is None:
        t = _EMPTY[dev] = torch.empty(0, device=dev)
    return t


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings

    def forward(self, means3D, means2D, opacities, confidences, shs=None, colors_precomp=None, scales=None,
                rotations=None, cov3D_precomp=None):
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            # (this and the next message are the upstream package's own strings, typo included - typed from memory of
            #  the public diff-gaussian-rasterization Python shim, which is not on this filesystem - so that a caller
            #  matching on them behaves the same)
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or (
                (scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        if cov3D_precomp is not None:
            raise NotImplementedError("surfels need scale+rotation (the normal is R[:,2]); cov3D_precomp is unsupported")
        s = self.raster_settings
        if shs is not None:
            # view-dependent colour from SH coefficients (N, K, 3): direction = mean - camera centre, normalised; the
            # result + 0.5 is clamped at 0 (the clamp passes no gradient where it is active, as upstream's `clamped` flag)
            deg = int(s.sh_degree)
            campos = s.campos.to(means3D.device, means3D.dtype).reshape(1, 3)
            dirs = means3D - campos
            dirs = dirs / dirs.norm(dim=1, keepdim=True).clamp_min(1e-20)
            colors_precomp = torch.clamp_min(eval_sh(deg, shs, dirs) + 0.5, 0.0)
        if not means3D.is_cuda:
            raise RuntimeError("diff_gaussian_rasterization_2d (MI355X build): tensors must be on the GPU; "
                               "there is no CPU fallback")
        b = _binding or _load_binding()
        none = _empty(means3D.device)
        mask = s.render_mask if s.render_mask is not None else none
        cfg = s.config if s.config is not None else none
        if means2D is None:
            means2D = none
        return tuple(b.rasterize(means3D, means2D, opacities, confidences, colors_precomp, scales, rotations, s.bg,
                                 s.viewmatrix, s.projmatrix, mask, cfg, int(s.image_height), int(s.image_width),
                                 float(s.tanfovx), float(s.tanfovy), float(s.scale_modifier), float(s.weight_thres),
                                 getattr(_tls, "deferred", False)))

    # (nn.Module.__call__ goes through the hook machinery: ~3 us per call that this module has no use for)
    __call__ = forward
