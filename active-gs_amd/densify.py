"""Map growth and pruning through the C ABI (``ags_smooth_depth``, ``ags_densify_candidates``,
``ags_voxel_select``, ``ags_prune_keep``, ``ags_compact_plan/rows``).

Host-side mirror of ``GaussianMap.add_gaussians`` / ``cal_mask`` / ``prune``
(/root/reference/mapping/gaussian_map.py:234-246,294-489), ``voxel_downsample`` and
``get_smooth_depth`` (/root/reference/utils/operations.py:161-169,603-625).  The map state is the
reference's: raw ``means (n,3)``, ``scales (n,3)``, ``rotations (n,4)``, ``opacities (n)``,
``harmonics (n,1,3)`` plus the non-learnable ``view_scores (n)``, ``view_supports (n)``,
``view_means (n,3)``.  Like the reference (boolean indexing / ``torch.cat``) each call learns the
new row count with one host read; everything else is stream-ordered device work.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Tuple

import torch

from . import _lib
from ._lib import ptr

STATE_KEYS = ("means", "scales", "rotations", "opacities", "harmonics", "view_scores", "view_supports", "view_means")
VOXEL_SIZE = 0.02                    # operations.py:603
MIN_OPACITY = 0.1                    # gaussian_map.py:235
NEW_Z_SCALE = -1e10                  # gaussian_map.py:373: surfels are flat


def _stream() -> int:
    return _lib.current_stream()


def _need_gpu(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on the GPU: map growth has no CPU path")
    return t.contiguous().float()


def smooth_depth(depth: torch.Tensor, d: int = 15, sigma_color: float = 0.5, sigma_space: float = 20.0) -> torch.Tensor:
    """``get_smooth_depth``: bilateral filter of a (1,H,W) / (H,W) depth image, invalid (<0) -> -1."""
    depth = _need_gpu(depth, "depth")
    h, w = depth.shape[-2:]
    out = torch.empty_like(depth)
    _lib.check(_lib.load().ags_smooth_depth(h, w, ptr(depth), ptr(out), d, sigma_color, sigma_space, _stream()),
               "ags_smooth_depth")
    return out


def candidates(frame: dict, depth_smooth: torch.Tensor, pred: Optional[dict], error_thres: float) -> dict:
    """Per-pixel candidate surfels of a keyframe and the ``select`` mask (before the voxel filter).
    ``frame``: rgb (3,H,W), depth (1,H,W), intrinsic (3,3 normalised), extrinsic (4,4 c2w);
    ``pred``: None (map not initialised) or dict(rgb (3,H,W), depth (H,W), opacity (H,W))."""
    rgb, depth = _need_gpu(frame["rgb"], "rgb"), _need_gpu(frame["depth"], "depth")
    dev = rgb.device
    h, w = rgb.shape[-2:]
    P = h * w
    # 3x3 inverse on the HOST (9 floats down, 9 up): on the GPU torch.linalg.inv is a rocSOLVER LU - a handful of tiny
    # launches (~0.9 ms per keyframe in the mapper loop) and, the first time in a process, the load of that library
    kinv = frame.get("intrinsic_inv")
    if kinv is None:
        kinv = torch.linalg.inv(frame["intrinsic"].detach().float().cpu())
    kinv = kinv.to(dev).float().contiguous()
    ext = frame["extrinsic"].to(dev).float().contiguous()
    f = _lib.AgsKeyframe(h, w, ptr(rgb), ptr(depth), ptr(kinv), ptr(ext))
    keep_alive = [rgb, depth, kinv, ext]
    if pred is not None:
        pr = [_need_gpu(pred[k], k) for k in ("rgb", "depth", "opacity")]
        keep_alive += pr
        pd = _lib.AgsDensifyPred(ptr(pr[0]), ptr(pr[1]), ptr(pr[2]))
    else:
        pd = _lib.AgsDensifyPred(None, None, None)
    out = dict(means=torch.empty(P, 3, device=dev), rotations=torch.empty(P, 4, device=dev),
               harmonics=torch.empty(P, 3, device=dev), select=torch.empty(P, device=dev, dtype=torch.int32))
    c = _lib.AgsCandidates(ptr(out["means"]), ptr(out["rotations"]), ptr(out["harmonics"]), ptr(out["select"]))
    ds = _need_gpu(depth_smooth, "depth_smooth")
    _lib.check(_lib.load().ags_densify_candidates(C.byref(f), ptr(ds), C.byref(pd), float(error_thres), C.byref(c),
                                                  _stream()), "ags_densify_candidates")
    return out


def voxel_select(points: torch.Tensor, select: torch.Tensor, voxel: float = VOXEL_SIZE) -> torch.Tensor:
    """In place: of the selected rows keep one per occupied voxel (``voxel_downsample``)."""
    lib = _lib.load()
    n = points.shape[0]
    ws = torch.empty(int(lib.ags_voxel_select_bytes(n)), device=points.device, dtype=torch.uint8)
    _lib.check(lib.ags_voxel_select(n, ptr(points), ptr(select), float(voxel), ptr(ws), ws.numel(), _stream()),
               "ags_voxel_select")
    return select


def compact_plan(keep: torch.Tensor, before_sync=None) -> Tuple[torch.Tensor, int]:
    """(dst_index, number of kept rows) - the one host read of the operation.  ``before_sync``: see add_gaussians
    (True -> (dst_index, None))."""
    lib = _lib.load()
    n = keep.shape[0]
    dst = torch.empty(max(n, 1), device=keep.device, dtype=torch.int32)
    total = torch.zeros(1, device=keep.device, dtype=torch.int32)
    scratch = torch.empty(int(lib.ags_compact_plan_bytes(n)), device=keep.device, dtype=torch.uint8)
    _lib.check(lib.ags_compact_plan(n, ptr(keep), ptr(dst), ptr(total), ptr(scratch), scratch.numel(), _stream()),
               "ags_compact_plan")
    if before_sync is not None and before_sync():
        return dst, None
    return dst, int(total.item())


def compact_rows(src: torch.Tensor, dst_index: torch.Tensor, dst: torch.Tensor) -> None:
    """dst[dst_index[i]] = src[i] for kept rows; ``dst`` is the first destination row (a view is fine)."""
    n = src.shape[0]
    if n == 0:
        return
    width = src.numel() // n
    assert src.is_contiguous() and dst.is_contiguous()
    _lib.check(_lib.load().ags_compact_rows(n, width, ptr(dst_index), ptr(src), ptr(dst), _stream()), "ags_compact_rows")


class MapArena:
    """The map's eight per-surfel arrays with room behind the last row.  The reference grows the map with ``torch.cat``
    and prunes it with boolean indexing (fresh tensors at every keyframe); here the arrays are views ``[:n]`` of buffers of
    ``cap`` rows - growing writes the new rows behind the old ones (one launch, ``ags_map_append``), pruning compacts into
    a second set of buffers (one launch, ``ags_map_compact``) and swaps.  Buffers double when they run out."""

    TAILS = dict(means=(3,), scales=(3,), rotations=(4,), opacities=(), harmonics=(1, 3), view_scores=(), view_supports=(),
                 view_means=(3,))

    def __init__(self, cap: int, device):
        self.cap, self.device = int(cap), device
        self.bufs = self._alloc(self.cap)
        self.alt = None                      # the buffers prune compacts into (made at the first prune)

    def _alloc(self, cap: int) -> Dict[str, torch.Tensor]:
        return {k: torch.empty((cap,) + t, device=self.device, dtype=torch.float32) for k, t in self.TAILS.items()}

    def views(self, n: int) -> Dict[str, torch.Tensor]:
        return {k: b[:n] for k, b in self.bufs.items()}

    def holds(self, state: Dict[str, torch.Tensor]) -> bool:
        """Are these tensors this arena's own leading rows?"""
        return all(state[k].data_ptr() == self.bufs[k].data_ptr() and state[k].shape[0] <= self.cap
                   and state[k].dtype == torch.float32 and state[k].is_contiguous() for k in STATE_KEYS)

    def adopt(self, state: Dict[str, torch.Tensor], room: int) -> None:
        """Make the arena hold ``state`` with at least ``room`` free rows behind it (copies only what it does not hold)."""
        n = state["means"].shape[0]
        held = self.holds(state)
        if n + room > self.cap:
            old = self.bufs if held else None
            self.cap = max(2 * (n + room), 1 << 16)
            self.bufs, self.alt = self._alloc(self.cap), None
            src = {k: old[k][:n] for k in STATE_KEYS} if held else state
            for k in STATE_KEYS:
                self.bufs[k][:n].copy_(src[k].reshape((n,) + self.TAILS[k]))
        elif not held:
            for k in STATE_KEYS:
                self.bufs[k][:n].copy_(state[k].reshape((n,) + self.TAILS[k]))

    def arrays(self, bufs: Dict[str, torch.Tensor], row: int) -> "_lib.AgsMapArrays":
        a = _lib.AgsMapArrays()
        for k in STATE_KEYS:
            width = 1
            for t in self.TAILS[k]:
                width *= t
            setattr(a, k, bufs[k].data_ptr() + 4 * width * row)
        return a


def add_gaussians(state: Dict[str, torch.Tensor], frame: dict, pred: Optional[dict], error_thres: float,
                  arena: Optional[MapArena] = None, before_sync=None):
    """``GaussianMap.add_gaussians`` (gaussian_map.py:294-462).  Returns (grown state, rows added).  With an ``arena`` the
    grown state is the arena's leading rows (no copy of the old map when it already lives there).
    ``before_sync`` (optional callable -> bool): called right before the one host read of the operation (the row count),
    when everything up to it is enqueued; if it returns True the map has changed under this call and nothing is appended
    (returns None: the caller starts over)."""
    ds = smooth_depth(frame["depth"])
    c = candidates(frame, ds, pred, error_thres)
    voxel_select(c["means"], c["select"])
    dst_index, k = compact_plan(c["select"], before_sync=before_sync)
    if k is None:
        return None
    n = state["means"].shape[0]
    dev = c["means"].device
    if arena is not None:
        arena.adopt(state, k)
        if k:
            cs = _lib.AgsCandidates(ptr(c["means"]), ptr(c["rotations"]), ptr(c["harmonics"]), ptr(c["select"]))
            first = arena.arrays(arena.bufs, n)
            _lib.check(_lib.load().ags_map_append(c["means"].shape[0], ptr(dst_index), C.byref(cs), NEW_Z_SCALE, C.byref(first),
                                                  _stream()), "ags_map_append")
        return arena.views(n + k), k
    out = {}
    for key in STATE_KEYS:
        old = state[key]
        new = torch.zeros((n + k,) + tuple(old.shape[1:]), device=dev, dtype=torch.float32)
        new[:n].copy_(old)
        out[key] = new
    if k:
        compact_rows(c["means"], dst_index, out["means"][n:])
        compact_rows(c["rotations"], dst_index, out["rotations"][n:])
        compact_rows(c["harmonics"], dst_index, out["harmonics"][n:])
        out["scales"][n:, 2] = NEW_Z_SCALE
    return out, k


def prune(state: Dict[str, torch.Tensor], prune_mask: Optional[torch.Tensor], arena: Optional[MapArena] = None
          ) -> Tuple[Dict[str, torch.Tensor], int]:
    """``GaussianMap.prune`` (gaussian_map.py:234-246).  Returns (pruned state, rows deleted)."""
    lib = _lib.load()
    opac = _need_gpu(state["opacities"], "opacities")
    n = opac.shape[0]
    dev = opac.device
    keep = torch.empty(max(n, 1), device=dev, dtype=torch.int32)
    pm = None if prune_mask is None else _need_gpu(prune_mask.to(dev).float(), "prune_mask")
    _lib.check(lib.ags_prune_keep(n, ptr(pm), ptr(opac), MIN_OPACITY, ptr(keep), _stream()), "ags_prune_keep")
    dst_index, k = compact_plan(keep[:n])
    if arena is not None and n > 0:
        arena.adopt(state, 0)
        if arena.alt is None:
            arena.alt = arena._alloc(arena.cap)
        src, dst = arena.arrays(arena.bufs, 0), arena.arrays(arena.alt, 0)
        _lib.check(lib.ags_map_compact(n, ptr(dst_index), C.byref(src), C.byref(dst), _stream()), "ags_map_compact")
        arena.bufs, arena.alt = arena.alt, arena.bufs
        return arena.views(k), n - k
    out = {}
    for key in STATE_KEYS:
        old = state[key].contiguous()
        new = torch.empty((k,) + tuple(old.shape[1:]), device=dev, dtype=torch.float32)
        if k:
            compact_rows(old, dst_index, new)
        out[key] = new
    return out, n - k
