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

import collections
import ctypes as C
import os
import weakref
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


# Binning algorithm the drop-in module starts a view size with (raster_api.BIN_DIRECT | BIN_TILE_SORT | BIN_RADIX).
# BIN_DIRECT needs tiles x the LONGEST tile list of key slots; a view size whose lists are badly skewed (a zoomed-out or
# distant camera: most surfels in a few tiles) is moved to BIN_TILE_SORT, which needs the instance TOTAL (same images).
BINNING_MODE = api.BIN_DIRECT
SKEW_FACTOR = 8                    # direct binning is left when it needs this many times the instance total ...
DIRECT_BUDGET_BYTES = 1 << 30      # ... AND more than this much workspace for its key slots (24 B each)

# How a forward pass learns whether its workspace was large enough (the device notes it in the status block):
#   "deferred" (default)  a call that has to MAKE a workspace (the first views of a map size / image size - allocation
#                         is the expensive part anyway) reads the status back and repairs an overflow on the spot, like
#                         the CUDA extension does on every call (its num_rendered read-back).  A call that finds a
#                         pooled workspace - sized HEADROOM x the largest need seen so far - copies the status block to
#                         page-locked memory WITHOUT waiting, and a later call (or ``check_overflow()``) looks at it
#                         once the copy has landed: no stream synchronisation per view, the host runs ahead of the GPU.
#                         A view that did outgrow its workspace (its tile lists were truncated: images and gradients
#                         of THAT call are invalid) is reported by a RuntimeError from the next call / check_overflow(),
#                         and the size is raised so that repeating the iteration succeeds.
#   "always"              read the status back after every forward pass (one stream synchronisation per view).
STATUS_CHECK = os.environ.get("AGS_DROPIN_STATUS", "deferred")
HEADROOM = 2.0        # a new workspace holds this many times the largest need seen so far
MIN_HEADROOM = 1.25   # a pooled workspace is reused while it holds at least this many times that need
MAX_PENDING = 64

# Largest capacity need seen per (device, h, w, binning mode) - monotone: views of one loop differ in what they need,
# and a figure that followed the last view would make the next one reject every pooled workspace sized for a lighter view.
_need_seen: dict = {}
_mode_for: dict = {}               # (device, h, w) -> binning mode in use for that view size
counters = {"forward_calls": 0, "status_syncs": 0, "deferred_checks": 0, "overflows": 0, "mode_switches": 0}

# Workspaces (the library's internal state of a view: projected records, keys, ranges, per-pixel blend state, gradient
# records - tens of MB) are pooled per (device, surfels, image size, mode): a call takes one, and it goes back when the
# autograd graph that may still need it for the backward pass is freed (a forward under no_grad returns it at once).
# Reuse is safe without re-initialisation: the forward pass leaves its counters clean, and everything runs on the
# caller's stream in order.  The image / per-Gaussian OUTPUT tensors are fresh per call - the caller owns them.
_workspace_pool: dict = {}
_POOL_MAX_PER_KEY = 16
_U32 = 0xFFFFFFFF


def _take_workspace(key, n, h, w, min_cap, cap, dev, mode):
    """A pooled workspace of at least ``min_cap`` key slots, or a new one of ``cap``; -> (tensor, slots, newly made)."""
    free = _workspace_pool.get(key)
    if free:
        while free:                # too-small ones are dropped: the need only grows
            ws, ws_cap = free.pop()
            if ws_cap >= min_cap:
                return ws, ws_cap, False
    ws = torch.empty(api.workspace_bytes(n, h, w, cap), device=dev, dtype=torch.uint8)
    wss = _lib.AgsWorkspace(ws.data_ptr(), ws.numel(), int(cap), int(mode))
    _lib.check(_lib.load().ags_workspace_init(C.byref(wss), n, h, w, torch.cuda.current_stream().cuda_stream),
               "ags_workspace_init")
    return ws, int(cap), True


def _give_workspace(key, ws, cap):
    free = _workspace_pool.setdefault(key, [])
    if len(free) < _POOL_MAX_PER_KEY:
        free.append((ws, cap))


# ---- deferred status checks: (event, page-locked status copy, what it belongs to) in call order
class _Slot:
    __slots__ = ("host", "words", "event")

    def __init__(self):
        self.host = torch.zeros(16, dtype=torch.int32).pin_memory()
        self.words = self.host.numpy()
        self.event = torch.cuda.Event()


_free_slots: list = []
_pending: collections.deque = collections.deque()
_overflow_reports: list = []


def _note_need(key, mode, n, words):
    """Fold one view's status words into the sizing state; returns a description if the view overflowed."""
    instances, overflow, needed = int(words[0]) & _U32, int(words[2]), int(words[7]) & _U32
    hk = key + (mode,)
    if needed > _need_seen.get(hk, 0):
        _need_seen[hk] = needed
    if mode == api.BIN_DIRECT and needed > SKEW_FACTOR * max(instances, 1 << 16) and needed * 24 > DIRECT_BUDGET_BYTES:
        # skewed tile lists: this view size goes on with the scan-based binning, sized by the instance total
        _mode_for[key] = api.BIN_TILE_SORT
        counters["mode_switches"] += 1
        tk = key + (api.BIN_TILE_SORT,)
        _need_seen[tk] = max(_need_seen.get(tk, 0), instances)
    if overflow:
        return (f"a {key[1]}x{key[2]} view of {n} surfels needed {needed} tile-instance slots "
                f"({instances} instances, binning mode {mode})")
    return None


def _poll_pending(block: bool = False) -> None:
    while _pending:
        slot, key, mode, n, cap = _pending[0]
        if not slot.event.query():
            if not block and len(_pending) <= MAX_PENDING:
                break
            slot.event.synchronize()
        _pending.popleft()
        counters["deferred_checks"] += 1
        what = _note_need(key, mode, n, slot.words)
        if what:
            counters["overflows"] += 1
            _overflow_reports.append(what + f" but its workspace held {cap}")
        _free_slots.append(slot)
    if _overflow_reports:
        msg = "; ".join(_overflow_reports)
        _overflow_reports.clear()
        raise RuntimeError("diff_gaussian_rasterization_2d: " + msg + ": the tile lists of that call were truncated, its "
                           "images and gradients are invalid.  The workspace size has been raised - repeat the iteration "
                           "(AGS_DROPIN_STATUS=always checks every call before it returns).")


def check_overflow() -> None:
    """Wait for the status copies of all forward passes issued so far and raise if one of them outgrew its workspace
    (see STATUS_CHECK).  A training loop calls this where it synchronises anyway (e.g. once per iteration)."""
    _poll_pending(block=True)


def _f32c(t, dev):
    if t.device != dev:
        if not t.is_cuda and dev.type == "cuda" and t.numel() <= 16:
            t = t.to(dev)          # tiny host-side settings tensors (bg, campos, ...) are moved; big ones are an error
        else:
            raise RuntimeError("diff_gaussian_rasterization_2d (MI355X build): every tensor must live on the GPU of "
                               f"means3D ({dev}), got {t.device}; there is no CPU fallback")
    if t.dtype is not torch.float32:
        t = t.to(torch.float32)
    return t if t.is_contiguous() else t.contiguous()


def _camera_struct(s: GaussianRasterizationSettings, dev):
    """AgsCamera of a settings record + the tensors it points into (kept alive by the caller).
    ``config`` (operations.py:697-699) stays on the device when it is a device tensor: the kernels read the four
    flags there (AgsCamera.config) and nothing is read back.  A host tensor / None is decoded here."""
    V, P, bg = _f32c(s.viewmatrix, dev), _f32c(s.projmatrix, dev), _f32c(s.bg, dev)
    mask = s.render_mask
    if mask is not None and mask.numel() > 0:
        mask = _f32c(mask, dev)
        if mask.numel() != int(s.image_height) * int(s.image_width):
            raise ValueError("render_mask must hold image_height*image_width values")
    else:
        mask = None
    cfg = s.config
    flags = (1, 1, 0, 0)
    if cfg is not None and cfg.is_cuda:
        if cfg.numel() < 5:
            raise ValueError("config must hold 5 values")
        cfg = _f32c(cfg, dev)
    elif cfg is not None:
        v = [float(x) for x in cfg.tolist()] + [0.0] * 5
        flags = (int(v[1] > 0), int(v[2] > 0), int(v[3] > 0), int(v[4] > 0))
        cfg = None
    cs = _lib.AgsCamera(int(s.image_height), int(s.image_width), float(s.tanfovx), float(s.tanfovy),
                        float(s.scale_modifier), float(s.weight_thres), flags[0], flags[1], flags[2], flags[3],
                        V.data_ptr(), P.data_ptr(), bg.data_ptr(), None if mask is None else mask.data_ptr(),
                        None if cfg is None else cfg.data_ptr())
    return cs, (V, P, bg, mask, cfg)


class _RasterizeSurfels(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, opacities, confidences, colors, scales, rotations, settings):
        if not means3D.is_cuda:
            raise RuntimeError("diff_gaussian_rasterization_2d (MI355X build): tensors must be on the GPU; "
                               "there is no CPU fallback")
        dev = means3D.device
        lib = _lib.load()
        if _pending or _overflow_reports:
            _poll_pending()                    # non-blocking look at earlier calls' status copies
        cs, cam_keep = _camera_struct(settings, dev)
        m3, sc, rot = _f32c(means3D, dev), _f32c(scales, dev), _f32c(rotations, dev)
        op, col, conf = _f32c(opacities, dev).reshape(-1), _f32c(colors, dev), _f32c(confidences, dev).reshape(-1)
        n, h, w = m3.shape[0], cs.image_height, cs.image_width
        gs = _lib.AgsGaussians(n, m3.data_ptr(), sc.data_ptr(), rot.data_ptr(), op.data_ptr(), col.data_ptr(),
                               conf.data_ptr(), 0, 0.01, 0.05)
        f = dict(device=dev, dtype=torch.float32)
        rgb, normal = torch.empty(3, h, w, **f), torch.empty(3, h, w, **f)
        depth, opacity, confidence = torch.empty(1, h, w, **f), torch.empty(1, h, w, **f), torch.empty(1, h, w, **f)
        radii = torch.empty(n, device=dev, dtype=torch.int32)
        # device-side flags: the per-Gaussian kernel clears the statistics itself; host flags: zero-filled here
        mk = torch.empty if cs.config else torch.zeros
        importance, count = mk(n, **f), mk(n, device=dev, dtype=torch.int32)
        im = _lib.AgsImages(rgb.data_ptr(), normal.data_ptr(), depth.data_ptr(), opacity.data_ptr(), confidence.data_ptr())
        pg = _lib.AgsPerGaussian(importance.data_ptr(), count.data_ptr(), radii.data_ptr(), _lib.AgsRowSet(None, None, None))
        stream = torch.cuda.current_stream().cuda_stream
        key = (dev.index, h, w)
        counters["forward_calls"] += 1
        must_sync, attempts = STATUS_CHECK == "always", 0
        while True:
            mode = _mode_for.get(key, BINNING_MODE)
            seen = _need_seen.get(key + (mode,), 0)
            floor = max(1 << 16, 2 * n)
            pkey = (dev.index, n, h, w, mode)
            ws, ws_cap, fresh = _take_workspace(pkey, n, h, w, min(max(int(seen * MIN_HEADROOM) + 1024, floor), _U32),
                                                min(max(int(seen * HEADROOM) + 1024, floor), _U32), dev, mode)
            wss = _lib.AgsWorkspace(ws.data_ptr(), ws.numel(), ws_cap, mode)
            _lib.check(lib.ags_forward(C.byref(cs), C.byref(gs), C.byref(im), C.byref(pg), C.byref(wss), stream), "ags_forward")
            if fresh or must_sync:
                # a workspace had to be made (first views of this map size / image size, or the need has outgrown the pooled ones)
                # or every call is to be checked: read the need back, like upstream's num_rendered read-back
                st = _lib.AgsStatus()
                _lib.check(lib.ags_read_status(C.byref(wss), C.byref(st), stream), "ags_read_status")
                counters["status_syncs"] += 1
                over = _note_need(key, mode, n, (st.num_instances, st.num_sorted, st.overflow, st.num_visible,
                                                st.peak_instances, st.overflow_passes, st.max_tile_instances,
                                                st.needed_instances))
                if over is None:
                    break
                # repaired here: re-run (checked again) in a workspace of the size just learnt; this one is dropped
                must_sync, attempts = True, attempts + 1
                if attempts > 4 or (ws_cap >= _U32 and _mode_for.get(key, BINNING_MODE) == mode):
                    raise RuntimeError("diff_gaussian_rasterization_2d: " + over + " - more than a workspace can hold")
                continue
            slot = _free_slots.pop() if _free_slots else _Slot()
            _lib.check(lib.ags_read_status_async(C.byref(wss), slot.host.data_ptr(), stream), "ags_read_status_async")
            slot.event.record()
            _pending.append((slot, key, mode, n, ws_cap))
            break
        if any(ctx.needs_input_grad):      # (all False under no_grad)
            weakref.finalize(ctx, _give_workspace, pkey, ws, ws_cap)   # back to the pool when the graph is freed
            # What the backward needs of the forward's OUTPUTS goes through save_for_backward: an output tensor kept as
            # a plain attribute of ctx would close a cycle (ctx -> tensor -> grad_fn = ctx) that only the cyclic garbage
            # collector breaks - the view's workspace (tens of MB) would outlive its graph by many iterations.
            ctx.save_for_backward(depth, opacity, radii)
            ctx.cs, ctx.gs, ctx.wss = cs, gs, wss
            ctx.keep = (cam_keep, m3, sc, rot, op, col, conf, ws)     # what the raw pointers in cs / gs / wss point into
            ctx.need_m2d = means2D is not None and means2D.requires_grad
            ctx.opac_shape = opacities.shape
        else:
            _give_workspace(pkey, ws, ws_cap)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(importance, count, radii)
        return rgb, normal, depth, opacity, confidence, importance, count, radii

    @staticmethod
    def backward(ctx, d_rgb, d_normal, d_depth, d_opacity, d_conf, *_unused):
        depth, opacity, radii = ctx.saved_tensors
        # (the blend backward reads the forward's depth and opacity images, its per-pixel state in the workspace and radii)
        gs = ctx.gs
        n, dev = gs.n, depth.device
        f = dict(device=dev, dtype=torch.float32)
        g_m, g_s, g_r = torch.empty(n, 3, **f), torch.empty(n, 3, **f), torch.empty(n, 4, **f)
        g_o, g_c = torch.empty(n, **f), torch.empty(n, 3, **f)
        g_m2 = torch.empty(n, 3, **f) if ctx.need_m2d else None
        c = lambda t: None if t is None else _f32c(t, dev)
        d_rgb, d_normal, d_depth, d_opacity, d_conf = c(d_rgb), c(d_normal), c(d_depth), c(d_opacity), c(d_conf)
        dout = _lib.AgsImageGrads(_lib.ptr(d_rgb), _lib.ptr(d_normal), _lib.ptr(d_depth), _lib.ptr(d_opacity), _lib.ptr(d_conf))
        din = _lib.AgsGaussianGrads(g_m.data_ptr(), g_s.data_ptr(), g_r.data_ptr(), g_o.data_ptr(), g_c.data_ptr(),
                                    _lib.ptr(g_m2), 0)
        im = _lib.AgsImages(None, None, depth.data_ptr(), opacity.data_ptr(), None)
        pg = _lib.AgsPerGaussian(None, None, radii.data_ptr(), _lib.AgsRowSet(None, None, None))
        _lib.check(_lib.load().ags_backward(C.byref(ctx.cs), C.byref(gs), C.byref(im), C.byref(pg), C.byref(dout), C.byref(din),
                                            C.byref(ctx.wss), torch.cuda.current_stream().cuda_stream), "ags_backward")
        return g_m, g_m2, g_o.reshape(ctx.opac_shape), None, g_c, g_s, g_r, None


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
