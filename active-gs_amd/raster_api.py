"""Functional host API over the C ABI (include/ags_raster.h): no autograd, no host sync.

``forward`` / ``backward`` enqueue the HIP kernels on the current torch stream and return
immediately; buffers are torch tensors only because torch owns device memory here.
Used directly by bench.py and the trainer, and wrapped by ``rasterizer.py`` for the
drop-in ``diff_gaussian_rasterization_2d`` module.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional, Sequence

import torch

from . import _lib
from ._lib import ptr


@dataclass
class Camera:
    """What GaussianRasterizationSettings carries (operations.py:682-700), flags decoded."""
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    bg: torch.Tensor
    scale_modifier: float = 1.0
    weight_thres: float = 0.03
    normalize_depth: bool = True
    perpix_depth: bool = True
    want_stats: bool = False
    front_only: bool = False
    render_mask: Optional[torch.Tensor] = None
    # the reference's `config` tensor itself (device, 5 floats): the kernels read the four flags above from it on the
    # device (AgsCamera.config) - what the drop-in module passes so that it never has to read the tensor back
    config: Optional[torch.Tensor] = None

    def c_struct(self) -> _lib.AgsCamera:
        m = self.render_mask
        if m is not None and m.numel() == 0:
            m = None
        if m is not None and m.numel() != self.image_height * self.image_width:
            raise ValueError("render_mask must hold image_height*image_width values")
        return _lib.AgsCamera(self.image_height, self.image_width, self.tanfovx, self.tanfovy,
                              self.scale_modifier, self.weight_thres, int(self.normalize_depth),
                              int(self.perpix_depth), int(self.want_stats), int(self.front_only),
                              ptr(self.viewmatrix), ptr(self.projmatrix), ptr(self.bg), ptr(m), ptr(self.config))


@dataclass
class Gaussians:
    means3D: torch.Tensor
    scales: torch.Tensor
    rotations: torch.Tensor
    opacities: torch.Tensor
    colors: torch.Tensor
    confidences: torch.Tensor
    raw_params: bool = False   # scales/rotations/opacities are raw map parameters (activated in-kernel)
    scale_factor: float = 0.01
    max_scale: float = 0.05

    @property
    def n(self) -> int:
        return self.means3D.shape[0]

    def c_struct(self) -> _lib.AgsGaussians:
        return _lib.AgsGaussians(self.n, ptr(self.means3D), ptr(self.scales), ptr(self.rotations),
                                 ptr(self.opacities), ptr(self.colors), ptr(self.confidences), int(self.raw_params),
                                 float(self.scale_factor), float(self.max_scale))


class RowSet:
    """Sticky set of surfels an optimisation loop has seen (``AgsRowSet`` in include/ags_raster.h):
    ``forward`` inserts the visible ones, ``backward`` and the fused Adam then touch only member
    rows.  ``reset()`` empties it - do that together with zeroing the gradient slab and the Adam
    moments (the reference re-creates its optimiser per train() call, gaussian_map.py:259-292)."""

    def __init__(self, n: int, device, zero: bool = True):
        self.buf = (torch.zeros if zero else torch.empty)(2 * n + 1, device=device, dtype=torch.int32)
        self.n = n
        self.member, self.rows, self.count = self.buf[:n], self.buf[n:2 * n], self.buf[2 * n:]

    def reset(self) -> None:
        self.buf.zero_()

    def c_struct(self) -> _lib.AgsRowSet:
        return _lib.AgsRowSet(ptr(self.member), ptr(self.rows), ptr(self.count))


def _rowset_struct(rs) -> _lib.AgsRowSet:
    return rs.c_struct() if rs is not None else _lib.AgsRowSet(None, None, None)


@dataclass
class ForwardState:
    rgb: torch.Tensor
    normal: torch.Tensor
    depth: torch.Tensor
    opacity: torch.Tensor
    confidence: torch.Tensor
    importance: torch.Tensor
    count: torch.Tensor
    radii: torch.Tensor
    workspace: torch.Tensor
    max_instances: int
    binning_mode: int = 2  # AGS_BIN_DIRECT (default) | 0 = AGS_BIN_TILE_SORT | 1 = AGS_BIN_RADIX
    tuning: Optional["_lib.AgsTuning"] = None   # kernel selection handed over with the workspace (None: the process default)

    def images_struct(self) -> _lib.AgsImages:
        return _lib.AgsImages(ptr(self.rgb), ptr(self.normal), ptr(self.depth), ptr(self.opacity), ptr(self.confidence))

    def per_gaussian_struct(self, touched=None) -> _lib.AgsPerGaussian:
        return _lib.AgsPerGaussian(ptr(self.importance), ptr(self.count), ptr(self.radii), _rowset_struct(touched))

    def ws_struct(self) -> _lib.AgsWorkspace:
        return _lib.workspace(ptr(self.workspace), self.workspace.numel(), self.max_instances,
                              int(self.binning_mode), self.tuning)


def _require_cuda(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on the GPU: the rasterizer has no CPU path")
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous float32")


def _stream() -> int:
    return _lib.current_stream()


def workspace_bytes(n: int, h: int, w: int, max_instances: int) -> int:
    return int(_lib.load().ags_workspace_bytes(n, h, w, max_instances))


BIN_TILE_SORT, BIN_RADIX, BIN_DIRECT = 0, 1, 2
# Camera.want_stats = STATS_SEEN (AGS_STATS_SEEN): count[i] = 1 for every surfel with a counted pixel, importance untouched -
# all the mapper's post-processing reads of its count render
STATS_SEEN = 2
REGION_FINAL_T, REGION_N_CONTRIB, REGION_GEOM, REGION_RANGES, REGION_KEYS, REGION_IDS = range(6)


def workspace_region(state: "ForwardState", n: int, h: int, w: int, region: int, dtype) -> torch.Tensor:
    """A typed view of one region of ``state.workspace`` (``ags_workspace_region``: the per-pixel blend state, the
    projected records, the tile ranges, the sorted keys) - diagnostics and parity tests."""
    off, nbytes = C.c_size_t(), C.c_size_t()
    _lib.check(_lib.load().ags_workspace_region(n, h, w, state.max_instances, int(state.binning_mode), int(region),
                                                C.byref(off), C.byref(nbytes)), "ags_workspace_region")
    return state.workspace[off.value:off.value + nbytes.value].view(dtype)


def last_contributor(state: "ForwardState", n: int, h: int, w: int) -> torch.Tensor:
    """(H, W) int64: id of the last surfel every pixel blended (-1: none), from the forward pass's per-pixel list
    position (``n_contrib``), the tile ranges and the sorted ids - comparable across binning modes and with a
    rasterizer whose tile lists are longer (every tile of a surfel's rect) but in the same order."""
    tx, ty = (w + 15) // 16, (h + 15) // 16
    T = tx * ty
    last = workspace_region(state, n, h, w, REGION_N_CONTRIB, torch.int32).view(h, w).long()
    rg = workspace_region(state, n, h, w, REGION_RANGES, torch.int32).view(T, 2).long() & 0xFFFFFFFF
    mode = int(state.binning_mode)
    if mode == BIN_RADIX:
        ids = workspace_region(state, n, h, w, REGION_IDS, torch.int32).long() & 0xFFFFFFFF
        begin = rg[:, 0]
    else:
        ids = workspace_region(state, n, h, w, REGION_KEYS, torch.int64) & 0xFFFFFFFF
        if mode == BIN_DIRECT:     # rg[slot] = {tile, count}; the list of slot s starts at s * tile_cap
            tile_cap = state.max_instances // T
            begin = torch.zeros(T, dtype=torch.long, device=rg.device)
            begin[rg[:, 0]] = torch.arange(T, device=rg.device) * tile_cap
        else:
            begin = rg[:, 0]
    ys, xs = torch.meshgrid(torch.arange(h, device=last.device), torch.arange(w, device=last.device), indexing="ij")
    tile = (ys // 16) * tx + xs // 16
    pos = (begin[tile] + last - 1).clamp_min(0)
    return torch.where(last > 0, ids[pos.clamp_max(ids.numel() - 1)], torch.full_like(last, -1))


def alloc_state(n: int, h: int, w: int, max_instances: int, device, binning_mode: int = BIN_DIRECT,
                tuning: Optional["_lib.AgsTuning"] = None) -> ForwardState:
    """Buffers + workspace for views of one size.  ``max_instances``: key slots of the workspace; what a view needs is
    ``read_status(state)["needed"]`` (mode-aware: total tile instances for the scan-based modes, tiles x longest
    tile list for ``BIN_DIRECT``, whose tiles own ``max_instances // tiles`` slots each)."""
    f = dict(device=device, dtype=torch.float32)
    st = ForwardState(
        rgb=torch.empty(3, h, w, **f), normal=torch.empty(3, h, w, **f), depth=torch.empty(1, h, w, **f),
        opacity=torch.empty(1, h, w, **f), confidence=torch.empty(1, h, w, **f),
        importance=torch.zeros(n, **f), count=torch.zeros(n, device=device, dtype=torch.int32),
        radii=torch.empty(n, device=device, dtype=torch.int32),
        workspace=torch.empty(workspace_bytes(n, h, w, max_instances), device=device, dtype=torch.uint8),
        max_instances=int(max_instances), binning_mode=int(binning_mode), tuning=tuning)
    ws = st.ws_struct()
    _lib.check(_lib.load().ags_workspace_init(C.byref(ws), n, h, w, _stream()), "ags_workspace_init")
    return st


def init_workspace(state: ForwardState, n: int, h: int, w: int) -> None:
    """``ags_workspace_init`` on a freshly allocated workspace (once per allocation: forwards leave it clean)."""
    ws = state.ws_struct()
    _lib.check(_lib.load().ags_workspace_init(C.byref(ws), n, h, w, _stream()), "ags_workspace_init")


def discard_pass(state: ForwardState, n: int, h: int, w: int) -> None:
    """``ags_workspace_discard_pass``: forget a prepared per-Gaussian stage (leaving the software pipeline) without
    touching the status block's sticky overflow notes."""
    ws = state.ws_struct()
    _lib.check(_lib.load().ags_workspace_discard_pass(C.byref(ws), n, h, w, _stream()), "ags_workspace_discard_pass")


def alloc_outputs(n: int, h: int, w: int, device, workspace: torch.Tensor, max_instances: int,
                  binning_mode: int = BIN_DIRECT, stats: bool = True, tuning: Optional["_lib.AgsTuning"] = None) -> ForwardState:
    """Fresh output tensors around an existing (initialised) workspace - what the drop-in module hands to its caller
    per call while the workspace itself is pooled."""
    f = dict(device=device, dtype=torch.float32)
    return ForwardState(
        rgb=torch.empty(3, h, w, **f), normal=torch.empty(3, h, w, **f), depth=torch.empty(1, h, w, **f),
        opacity=torch.empty(1, h, w, **f), confidence=torch.empty(1, h, w, **f),
        importance=torch.zeros(n, **f), count=torch.zeros(n, device=device, dtype=torch.int32),
        radii=torch.empty(n, device=device, dtype=torch.int32), workspace=workspace,
        max_instances=int(max_instances), binning_mode=int(binning_mode), tuning=tuning)


def forward(cam: Camera, g: Gaussians, state: ForwardState, stream: Optional[int] = None, checked: bool = False,
            touched: Optional[RowSet] = None, resume: bool = False) -> ForwardState:
    """Enqueue one forward pass into ``state`` (see ``alloc_state``). Asynchronous.
    ``touched``: a ``RowSet`` the visible surfels are inserted into (training loops).
    ``stream``: raw HIP stream handle (default: torch's current stream); ``checked``: the caller
    vouches for contiguous float32 GPU inputs (hot loops skip the per-call validation).
    ``resume``: the per-Gaussian stage of this pass has already run (``backward(..., next_view=(cam, state))`` of the
    previous optimisation step): only the tile sort and the blend are launched (``ags_forward_resume``)."""
    lib = _lib.load()
    if not checked:
        for name in ("means3D", "scales", "rotations", "opacities", "colors", "confidences"):
            _require_cuda(getattr(g, name), name)
    if cam.want_stats and cam.config is None:     # (device-side configuration: the per-Gaussian kernel clears them)
        if int(cam.want_stats) != STATS_SEEN:       # (seen flags: importance is not touched)
            state.importance.zero_()
        state.count.zero_()
    cs, gs = cam.c_struct(), g.c_struct()
    im, pg, ws = state.images_struct(), state.per_gaussian_struct(touched), state.ws_struct()
    if resume:
        _lib.check(lib.ags_forward_resume(C.byref(cs), C.byref(gs), C.byref(im), C.byref(pg), C.byref(ws),
                                          _stream() if stream is None else stream), "ags_forward_resume")
        return state
    _lib.check(lib.ags_forward(C.byref(cs), C.byref(gs), C.byref(im), C.byref(pg), C.byref(ws),
                               _stream() if stream is None else stream), "ags_forward")
    return state


def read_status(state: ForwardState) -> dict:
    """Blocking read of the device status block (instances needed, overflow flag, ...)."""
    lib = _lib.load()
    st = _lib.AgsStatus()
    ws = state.ws_struct()
    _lib.check(lib.ags_read_status(C.byref(ws), C.byref(st), _stream()), "ags_read_status")
    return dict(num_instances=st.num_instances, num_sorted=st.num_sorted, overflow=bool(st.overflow),
                num_visible=st.num_visible, peak_instances=st.peak_instances, overflow_passes=st.overflow_passes,
                max_tile_instances=st.max_tile_instances, needed=st.needed_instances)


@dataclass
class GaussianGrads:
    means3D: torch.Tensor
    scales: torch.Tensor
    rotations: torch.Tensor
    opacities: torch.Tensor
    colors: torch.Tensor
    means2D: Optional[torch.Tensor] = None


def alloc_grads(n: int, device, with_means2d: bool = False, zero: bool = False) -> GaussianGrads:
    mk = torch.zeros if zero else torch.empty
    f = dict(device=device, dtype=torch.float32)
    return GaussianGrads(mk(n, 3, **f), mk(n, 3, **f), mk(n, 4, **f), mk(n, **f), mk(n, 3, **f),
                         mk(n, 3, **f) if with_means2d else None)


def backward(cam: Camera, g: Gaussians, state: ForwardState, d_rgb=None, d_normal=None, d_depth=None,
             d_opacity=None, d_confidence=None, grads: Optional[GaussianGrads] = None,
             accumulate: bool = False, adam_tick=None, stream: Optional[int] = None,
             touched: Optional[RowSet] = None, fused_adam=None, pack=None, next_view=None,
             defer_rows: bool = False) -> GaussianGrads:
    """Enqueue the backward pass of the view held in ``state``. Asynchronous.
    ``adam_tick`` = (device_clock_tensor, lrs, beta1, beta2): also advance that Adam clock.
    ``touched``: the ``RowSet`` given to this view's ``forward``: only its rows are written.
    ``fused_adam`` = (AgsAdamTensors struct, eps): the launch also performs the optimiser step for
    the member rows (last view of a single-GPU step; needs ``touched`` and ``adam_tick``).
    ``pack`` = (segment tensor, capacity): the member rows' totals leave as the rank's exchange segment
    (last view of a rank's data-parallel step; needs ``touched``), the gradient rows are left zeroed.
    ``next_view`` = (Camera, ForwardState[, rows_hint]) with ``fused_adam``: software-pipelined step - the per-Gaussian launch also
    runs the per-Gaussian stage of the NEXT forward pass (``ags_backward_fused_next``); render that view with
    ``forward(..., resume=True)``.
    ``defer_rows``: run the blend backward only; ``backward_rows`` later turns the gradient records of all the step's views
    into parameter gradients in one launch (every such view needs its own ``state`` until then)."""
    lib = _lib.load()
    if defer_rows:
        grads = grads or GaussianGrads(None, None, None, None, None)
    if grads is None:
        grads = alloc_grads(g.n, g.means3D.device)
        accumulate = False
    for name, t in (("d_rgb", d_rgb), ("d_normal", d_normal), ("d_depth", d_depth), ("d_opacity", d_opacity),
                    ("d_confidence", d_confidence)):
        if t is not None:
            _require_cuda(t, name)
    cs, gs = cam.c_struct(), g.c_struct()
    im, pg, ws = state.images_struct(), state.per_gaussian_struct(), state.ws_struct()
    dout = _lib.AgsImageGrads(ptr(d_rgb), ptr(d_normal), ptr(d_depth), ptr(d_opacity), ptr(d_confidence))
    din = _lib.AgsGaussianGrads(ptr(grads.means3D), ptr(grads.scales), ptr(grads.rotations), ptr(grads.opacities),
                                ptr(grads.colors), ptr(grads.means2D), int(accumulate))  # 2 = atomic accumulate
    if adam_tick is not None:
        clock, lrs, b1, b2 = adam_tick
        din.adam_clock = ptr(clock)
        for k in range(5):
            din.adam_lr[k] = float(lrs[k])
        din.adam_beta1, din.adam_beta2 = float(b1), float(b2)
    din.touched = _rowset_struct(touched)
    din.defer_rows = int(defer_rows)
    if fused_adam is not None:
        tensors, eps = fused_adam
        din.fused_adam = C.cast(C.pointer(tensors), C.c_void_p)
        din.adam_eps = float(eps)
    if pack is not None:
        segment, capacity = pack
        din.pack_segment = ptr(segment)
        din.pack_capacity = int(capacity)
    if next_view is not None:
        cam2, state2 = next_view[:2]
        rows_hint = int(next_view[2]) if len(next_view) > 2 else 0
        cs2, pg2, ws2 = cam2.c_struct(), state2.per_gaussian_struct(touched), state2.ws_struct()
        _lib.check(lib.ags_backward_fused_next(C.byref(cs), C.byref(gs), C.byref(im), C.byref(pg), C.byref(dout), C.byref(din),
                                               C.byref(ws), C.byref(cs2), C.byref(pg2), C.byref(ws2), rows_hint,
                                               _stream() if stream is None else stream), "ags_backward_fused_next")
        return grads
    _lib.check(lib.ags_backward(C.byref(cs), C.byref(gs), C.byref(im), C.byref(pg), C.byref(dout), C.byref(din),
                                C.byref(ws), _stream() if stream is None else stream), "ags_backward")
    return grads


def backward_rows(views: Sequence, g: Gaussians, grads: GaussianGrads, touched: Optional[RowSet], accumulate: bool = False,
                  adam_clock=None, fused_adam=None, pack=None, stream: Optional[int] = None,
                  row_range: Optional[tuple] = None) -> GaussianGrads:
    """``ags_backward_rows``: the per-Gaussian backward of ALL the views of a step (``views`` = [(Camera, ForwardState)],
    each rendered and taken through ``backward(..., defer_rows=True)``) in one launch over the member rows of ``touched``:
    the views' gradients are summed in registers, then written to ``grads`` / packed as the rank's exchange segment
    (``pack``) / consumed by the fused Adam step (``fused_adam`` = (AgsAdamTensors, eps), ``adam_clock`` = the optimiser's
    ``tick_args()`` whose clock one of the views' ``backward`` calls has already advanced).
    ``touched=None`` with ``row_range=(begin, end)``: no row set - the rows [begin, end) of the map, ``grads`` overwritten
    (the dense form of data-parallel ranks, cut into row chunks whose all-reduces overlap the next chunk's launch)."""
    lib = _lib.load()
    if isinstance(views, tuple) and len(views) == 2 and isinstance(views[0], ViewRefs):
        refs, n = views[0].refs, int(views[1])              # (ViewBatch.view_refs(), the first n of them)
    else:
        n = len(views)
        refs = (_lib.AgsViewRef * n)()
        keep = []
        for k, (cam, state) in enumerate(views):
            cs, ws = cam.c_struct(), state.ws_struct()
            keep.append((cs, ws))
            refs[k].cam, refs[k].radii, refs[k].ws = C.pointer(cs), ptr(state.radii), C.pointer(ws)
    gs = g.c_struct()
    din = _lib.AgsGaussianGrads(ptr(grads.means3D), ptr(grads.scales), ptr(grads.rotations), ptr(grads.opacities),
                                ptr(grads.colors), ptr(grads.means2D), int(accumulate))
    din.touched = _rowset_struct(touched)
    if row_range is not None:
        din.row_begin, din.row_end = int(row_range[0]), int(row_range[1])
    if adam_clock is not None:
        clock, lrs, b1, b2 = adam_clock
        din.adam_clock = ptr(clock)
        for k in range(5):
            din.adam_lr[k] = float(lrs[k])
        din.adam_beta1, din.adam_beta2 = float(b1), float(b2)
    if fused_adam is not None:
        tensors, eps = fused_adam
        din.fused_adam = C.cast(C.pointer(tensors), C.c_void_p)
        din.adam_eps = float(eps)
    if pack is not None:
        segment, capacity = pack
        din.pack_segment = ptr(segment)
        din.pack_capacity = int(capacity)
    _lib.check(lib.ags_backward_rows(refs, n, C.byref(gs), C.byref(din), _stream() if stream is None else stream),
               "ags_backward_rows")
    return grads


class StreamPool:
    """A few HIP streams that fork from and join back into torch's current stream."""

    def __init__(self, n: int = 4):
        self.streams = [torch.cuda.Stream() for _ in range(max(1, n))]

    def fork(self):
        main = torch.cuda.current_stream()
        for s in self.streams:
            s.wait_stream(main)
        return [s.cuda_stream for s in self.streams]

    def join(self):
        main = torch.cuda.current_stream()
        for s in self.streams:
            main.wait_stream(s)


def forward_many(cams: Sequence[Camera], g: Gaussians, states: Sequence[ForwardState],
                 pool: Optional[StreamPool] = None) -> None:
    """Forward-only render of many (small) views, e.g. the ~100 candidate views at 128x128 of the
    planners' utility pass (/root/reference/planning/confidence.py:24-46,
    /root/reference/config/planner/confidence.yaml:9,15) or the K keyframes of the prune pass
    (/root/reference/mapping/gaussian_map.py:149-192).  One view is far too small to fill the GPU
    (64 tiles), so views are enqueued round-robin on the pool's streams and run concurrently; the
    streams join torch's current stream before returning.  Asynchronous, no host sync."""
    if len(cams) != len(states):
        raise ValueError("one ForwardState per camera")
    for name in ("means3D", "scales", "rotations", "opacities", "colors", "confidences"):
        _require_cuda(getattr(g, name), name)
    if pool is None or len(cams) < 2:
        for cam, st in zip(cams, states):
            forward(cam, g, st, checked=True)
        return
    for cam, st in zip(cams, states):   # stats buffers are zeroed on the main stream, before the fork
        if cam.want_stats:
            st.importance.zero_()
            st.count.zero_()
    handles = pool.fork()
    lib = _lib.load()
    for k, (cam, st) in enumerate(zip(cams, states)):
        cs, gs = cam.c_struct(), g.c_struct()
        im, pg, ws = st.images_struct(), st.per_gaussian_struct(), st.ws_struct()
        _lib.check(lib.ags_forward(C.byref(cs), C.byref(gs), C.byref(im), C.byref(pg), C.byref(ws),
                                   handles[k % len(handles)]), "ags_forward")
    pool.join()


class ViewRefs:
    """A prebuilt ``AgsViewRef`` array (``ViewBatch.view_refs()``): what ``backward_rows`` builds from a list of
    (Camera, ForwardState) at every call, kept for the life of a binding."""

    def __init__(self, refs, keep):
        self.refs, self.keep = refs, keep


class ViewBatch:
    """``V`` forward-only views of one size and field of view rendered by ONE set of launches
    (``ags_forward_batch``: blockIdx.y = view).

    The planners' utility pass renders ~100 candidate views at 128x128 per planning step
    (/root/reference/planning/confidence.py:24-46) and the prune pass renders every keyframe
    (/root/reference/mapping/gaussian_map.py:149-192).  One such view is 64 tiles - it cannot fill
    256 CUs, and 600 separate launches are host-bound besides; a batch is one big grid.
    Outputs are batched tensors (``rgb`` (V,3,H,W), ``depth`` (V,1,H,W), ..., ``count`` (V,n));
    ``states[v]`` is a per-view ``ForwardState`` of slices of them.  ``mode="streams"`` keeps the
    older path (per-view launches on a stream pool, replayed from a hipGraph) for comparison."""

    def __init__(self, g: Gaussians, num_views: int, height: int, width: int, tanfovx: float, tanfovy: float,
                 bg: torch.Tensor, max_instances: int, num_streams: int = 8, want_stats: bool = False,
                 front_only: bool = False, render_masks: Optional[torch.Tensor] = None,
                 binning_mode: int = BIN_DIRECT, mode: str = "batched", capacity_n: Optional[int] = None,
                 tuning: Optional["_lib.AgsTuning"] = None):
        self.tuning = tuning
        if mode not in ("batched", "streams"):
            raise ValueError("mode is 'batched' or 'streams'")
        dev = g.means3D.device
        V, h, w = int(num_views), int(height), int(width)
        ncap = max(int(capacity_n or 0), g.n)          # buffers sized for maps of up to this many surfels
        self.num_views, self.mode, self.want_stats = V, mode, want_stats
        self.capacity_n, self.max_instances, self.binning_mode = ncap, int(max_instances), int(binning_mode)
        self._hw = (h, w)
        self.viewmats = torch.zeros(V, 4, 4, device=dev)
        self.projmats = torch.zeros(V, 4, 4, device=dev)
        self.masks = None if render_masks is None else render_masks.to(dev).float().reshape(V, h, w).contiguous()
        f = dict(device=dev, dtype=torch.float32)
        self.rgb, self.normal = torch.empty(V, 3, h, w, **f), torch.empty(V, 3, h, w, **f)
        self.depth, self.opacity, self.confidence = (torch.empty(V, 1, h, w, **f) for _ in range(3))
        self._importance = torch.zeros(V * ncap, **f)
        self._count = torch.zeros(V * ncap, device=dev, dtype=torch.int32)
        self._radii = torch.empty(V * ncap, device=dev, dtype=torch.int32)
        self.workspace = torch.empty(V * workspace_bytes(ncap, h, w, max_instances), device=dev, dtype=torch.uint8)
        self._graph = None
        self.bind(g)
        self.cam = Camera(h, w, tanfovx, tanfovy, self.viewmats[0], self.projmats[0], bg, want_stats=want_stats,
                          front_only=front_only, render_mask=None if self.masks is None else self.masks[0])
        self.cams = [Camera(h, w, tanfovx, tanfovy, self.viewmats[v], self.projmats[v], bg, want_stats=want_stats,
                            front_only=front_only, render_mask=None if self.masks is None else self.masks[v])
                     for v in range(V)]
        self.pool = StreamPool(min(num_streams, max(1, V))) if mode == "streams" else None

    def bind(self, g: Gaussians) -> None:
        """(Re)attach a map of ``g.n <= capacity_n`` surfels: the per-view workspaces are laid out for
        that size inside the allocated buffers and re-initialised.  Lets a training loop whose map grows
        every keyframe keep one allocation."""
        if g.n > self.capacity_n:
            raise ValueError("map larger than this ViewBatch's capacity_n")
        for name in ("means3D", "scales", "rotations", "opacities", "colors", "confidences"):
            _require_cuda(getattr(g, name), name)
        V, n, (h, w) = self.num_views, g.n, self._hw
        self.g = g
        self._graph = None                     # a recorded replay (mode "streams") points at the old layout
        per = workspace_bytes(n, h, w, self.max_instances)
        self._per = per
        self.importance = self._importance[:V * n].view(V, n)
        self.count = self._count[:V * n].view(V, n)
        self.radii = self._radii[:V * n].view(V, n)
        self._states = None                    # per-view ForwardStates (slices of the batch buffers): made when somebody asks
        self._refs = None                      # view_refs()
        ws = _lib.workspace(ptr(self.workspace), V * per, self.max_instances, self.binning_mode, self.tuning)
        _lib.check(_lib.load().ags_workspace_init_batch(C.byref(ws), V, n, h, w, _stream()), "ags_workspace_init_batch")

    @property
    def states(self):
        """``states[v]``: view v's outputs and workspace as a ``ForwardState`` of slices (built on first use after a
        ``bind``: a training loop that re-binds at every keyframe never needs them - ninety tensor slices)."""
        if self._states is None:
            per = self._per
            self._states = [ForwardState(self.rgb[v], self.normal[v], self.depth[v], self.opacity[v], self.confidence[v],
                                         self.importance[v], self.count[v], self.radii[v],
                                         self.workspace[v * per:(v + 1) * per], self.max_instances, self.binning_mode,
                                         self.tuning)
                            for v in range(self.num_views)]
        return self._states

    def view_refs(self) -> "ViewRefs":
        """The views of this batch as ``backward_rows`` takes them (``AgsViewRef`` array over cameras, radii and per-view
        workspaces), from the batch buffers' addresses - no tensor slices; valid until the next ``bind``."""
        if self._refs is None:
            V, n, per = self.num_views, self.g.n, self._per
            refs = (_lib.AgsViewRef * V)()
            keep = []
            base_ws, base_radii = ptr(self.workspace), ptr(self._radii)
            for v in range(V):
                cs = self.cams[v].c_struct()
                ws = _lib.workspace(base_ws + v * per, per, self.max_instances, self.binning_mode, self.tuning)
                keep.append((cs, ws))
                refs[v].cam, refs[v].radii, refs[v].ws = C.pointer(cs), base_radii + 4 * v * n, C.pointer(ws)
            self._refs = ViewRefs(refs, keep)
        return self._refs

    def _structs(self, touched=None):
        im = _lib.AgsImages(ptr(self.rgb), ptr(self.normal), ptr(self.depth), ptr(self.opacity), ptr(self.confidence))
        pg = _lib.AgsPerGaussian(ptr(self.importance), ptr(self.count), ptr(self.radii), _rowset_struct(touched))
        ws = _lib.workspace(ptr(self.workspace), self.workspace.numel(), self.max_instances, self.binning_mode, self.tuning)
        return im, pg, ws

    def _enqueue_batched(self, views: Optional[int] = None, touched: Optional[RowSet] = None, loss=None) -> None:
        lib = _lib.load()
        if self.want_stats:
            if int(self.want_stats) != STATS_SEEN:
                self.importance.zero_()
            self.count.zero_()
        cs, gs = self.cam.c_struct(), self.g.c_struct()
        im, pg, ws = self._structs(touched)
        if loss is not None:     # an _lib.AgsLossEpilogue: stage 1 of the loss head rides in the blend kernel's epilogue
            _lib.check(lib.ags_forward_batch_loss(C.byref(cs), self.num_views if views is None else int(views), C.byref(gs),
                                                  C.byref(im), C.byref(pg), C.byref(ws), C.byref(loss), _stream()),
                       "ags_forward_batch_loss")
            return
        _lib.check(lib.ags_forward_batch(C.byref(cs), self.num_views if views is None else int(views), C.byref(gs),
                                         C.byref(im), C.byref(pg), C.byref(ws), _stream()), "ags_forward_batch")

    def forward(self, views: Optional[int] = None, touched: Optional[RowSet] = None, loss=None) -> None:
        """Render the first ``views`` poses currently held in ``viewmats`` / ``projmats`` (training
        loops stage them with one index_select); ``touched``: see ``forward``; ``loss``: ``FusedLoss.epilogue(...)`` -
        the forward also runs stage 1 of the loss head (``ags_forward_batch_loss``)."""
        self._enqueue_batched(views, touched, loss)

    def backward(self, views: int, d_rgb, d_normal, d_depth, grads: "GaussianGrads", touched: Optional[RowSet] = None,
                 adam_tick=None, defer_rows: bool = False) -> None:
        """Backward of the first ``views`` views of the last ``forward``: image-gradient batches
        ``(views,C,H,W)`` (None = zeros), gradients of all views SUMMED atomically into the pre-zeroed
        ``grads``.  ``defer_rows``: the blend backward only; ``backward_rows([(batch.cams[v], batch.states[v]) ...])``
        then does the per-Gaussian backward of all the views (and the optimiser step) in one launch."""
        lib = _lib.load()
        cs, gs = self.cam.c_struct(), self.g.c_struct()
        im, pg, ws = self._structs()
        dout = _lib.AgsImageGrads(ptr(d_rgb), ptr(d_normal), ptr(d_depth), None, None)
        din = _lib.AgsGaussianGrads(ptr(grads.means3D), ptr(grads.scales), ptr(grads.rotations), ptr(grads.opacities),
                                    ptr(grads.colors), ptr(grads.means2D), 2)
        if adam_tick is not None:
            clock, lrs, b1, b2 = adam_tick
            din.adam_clock = ptr(clock)
            for k in range(5):
                din.adam_lr[k] = float(lrs[k])
            din.adam_beta1, din.adam_beta2 = float(b1), float(b2)
        din.touched = _rowset_struct(touched)
        din.defer_rows = int(defer_rows)
        _lib.check(lib.ags_backward_batch(C.byref(cs), int(views), C.byref(gs), C.byref(im), C.byref(pg), C.byref(dout),
                                          C.byref(din), C.byref(ws), _stream()), "ags_backward_batch")

    def render(self, viewmats: torch.Tensor, projmats: torch.Tensor, use_graph: bool = True):
        """Render the ``V`` poses (``(V,4,4)`` view and view-projection matrices, row-vector
        convention as everywhere).  Asynchronous; returns the per-view ``ForwardState`` list."""
        self.viewmats.copy_(viewmats.reshape(self.num_views, 4, 4), non_blocking=True)
        self.projmats.copy_(projmats.reshape(self.num_views, 4, 4), non_blocking=True)
        if self.mode == "batched":
            self._enqueue_batched()
            return self.states
        if not use_graph:
            forward_many(self.cams, self.g, self.states, self.pool)
            return self.states
        if self._graph is None:
            forward_many(self.cams, self.g, self.states, self.pool)   # warm-up outside capture
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
                    forward_many(self.cams, self.g, self.states, self.pool)
            torch.cuda.current_stream().wait_stream(side)
            self._graph = graph
        self._graph.replay()
        return self.states

    def status_words(self, views: Optional[int] = None) -> torch.Tensor:
        """The first eight words of every view's status block as a (views, 8) int32 DEVICE tensor (a copy enqueued on the
        current stream; ``statuses`` reads it back)."""
        V = self.num_views if views is None else int(views)
        per = self._per
        return self.workspace[:self.num_views * per].view(self.num_views, per)[:V, :32].contiguous().view(torch.int32).view(V, 8)

    def statuses(self, views: Optional[int] = None) -> torch.Tensor:
        """Blocking: the first eight words of every view's status block (``AgsStatus``) in ONE transfer ->
        (views, 8) int64 on the host: [0] tile instances, [1] instances sorted, [2] overflow flag, [3] visible,
        the two sticky words since the workspaces were (re)bound - [4] peak of the needed capacity, [5] number of
        overflowed passes - and [6] longest tile list, [7] the capacity (``max_instances``) this view needs."""
        V = self.num_views if views is None else int(views)
        per = self._per
        words = self.workspace[:self.num_views * per].view(self.num_views, per)[:V, :32].contiguous().view(torch.int32).view(V, 8)
        return (words.cpu().to(torch.int64)) & 0xFFFFFFFF

    def overflowed(self, views: Optional[int] = None) -> bool:
        """Blocking: did any view need more tile instances than ``max_instances`` in its LAST pass?
        (``statuses()[:, 5]``: in any pass since ``bind``.)"""
        return bool(self.statuses(views)[:, 2].any())
