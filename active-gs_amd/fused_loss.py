"""Host binding of the fused loss head (C ABI ``ags_loss_stage1/2``): facade post-processing +
depth->normal + the losses of /root/reference/mapping/gaussian_map.py:106-124, forward and
backward, on the GPU.  See csrc/loss.hip."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import torch

from . import _lib
from ._lib import ptr
from .raster_api import ForwardState


@dataclass
class LossBuffers:
    """Per-view scratch: the post-processed normal image and the three image gradients."""
    n_img: torch.Tensor
    d_rgb: torch.Tensor
    d_normal: torch.Tensor
    d_depth: torch.Tensor


class FusedLoss:
    def __init__(self, h: int, w: int, fov_x: float, fov_y: float, batch_total: int, max_views: int, device,
                 weights=(1.0, 0.8, 0.1, 0.1), sigma: float = 0.3):
        self.h, self.w = h, w
        stride = 4 + 2 * max_views
        self.cfg = _lib.AgsLossConfig(h, w, float(fov_x), float(fov_y), int(batch_total), *[float(x) for x in weights],
                                      float(sigma), stride, 0)
        self.msum = torch.zeros(h, w, dtype=torch.int32, device=device)
        # 64 accumulator rows: workgroups spread their atomics over them, readers sum the rows
        self.accum = torch.zeros(64, stride, dtype=torch.float32, device=device)
        self.device = device

    def set_batch_total(self, b: int) -> None:
        self.cfg.batch_total = int(b)

    def alloc_view(self) -> LossBuffers:
        f = dict(device=self.device, dtype=torch.float32)
        return LossBuffers(torch.empty(3, self.h, self.w, **f), torch.empty(3, self.h, self.w, **f),
                           torch.empty(3, self.h, self.w, **f), torch.empty(1, self.h, self.w, **f))

    def begin_step(self) -> None:
        self.accum.zero_()

    def stage1(self, st: ForwardState, gt_rgb, gt_depth, buf: LossBuffers, view: int, first_view, stream=None) -> None:
        """``first_view``: True / False for views processed one after another, -1 when views run
        concurrently on several streams (``msum`` must have been zeroed)."""
        img = st.images_struct()
        _lib.check(_lib.load().ags_loss_stage1(C.byref(self.cfg), C.byref(img), ptr(gt_rgb), ptr(gt_depth), ptr(buf.n_img),
                                               ptr(buf.d_rgb), ptr(buf.d_depth), ptr(self.msum), ptr(self.accum),
                                               int(view), int(first_view),
                                               _lib.current_stream() if stream is None else stream),
                   "ags_loss_stage1")

    def stage2(self, st: ForwardState, gt_depth, buf: LossBuffers, stream=None) -> None:
        img = st.images_struct()
        _lib.check(_lib.load().ags_loss_stage2(C.byref(self.cfg), C.byref(img), ptr(buf.n_img), ptr(gt_depth),
                                               ptr(self.msum), ptr(buf.d_normal), ptr(buf.d_depth), ptr(self.accum),
                                               _lib.current_stream() if stream is None else stream),
                   "ags_loss_stage2")

    # ---- all views of the batch in one launch each (blockIdx.y = view)
    def alloc_batch(self, views: int) -> LossBuffers:
        f = dict(device=self.device, dtype=torch.float32)
        return LossBuffers(torch.empty(views, 3, self.h, self.w, **f), torch.empty(views, 3, self.h, self.w, **f),
                           torch.empty(views, 3, self.h, self.w, **f), torch.empty(views, 1, self.h, self.w, **f))

    def stage1_batch(self, images: "_lib.AgsImages", gt_rgb, gt_depth, buf: LossBuffers, views: int, gt_index=None) -> None:
        """``images``: AgsImages of (views,C,H,W) batches; ``msum`` must be zero (atomic counting).  ``gt_index`` (int64 on
        the device): view v's ground truth is frame ``gt_index[v]`` of ``gt_rgb`` / ``gt_depth`` = the whole keyframe store."""
        self.cfg.num_views = int(views)
        self.cfg.gt_frame_index = ptr(gt_index)
        try:
            _lib.check(_lib.load().ags_loss_stage1(C.byref(self.cfg), C.byref(images), ptr(gt_rgb), ptr(gt_depth),
                                                   ptr(buf.n_img), ptr(buf.d_rgb), ptr(buf.d_depth), ptr(self.msum),
                                                   ptr(self.accum), 0, -1, _lib.current_stream()),
                       "ags_loss_stage1")
        finally:
            self.cfg.num_views = 0
            self.cfg.gt_frame_index = None

    def epilogue(self, gt_rgb, gt_depth, buf: LossBuffers, gt_index=None) -> "_lib.AgsLossEpilogue":
        """What ``ViewBatch.forward(views, loss=...)`` takes: stage 1 of the loss head as the epilogue of the forward blend
        kernel (``ags_forward_batch_loss`` = ``ags_forward_batch`` + ``stage1_batch`` in one set of launches; ``msum`` must be
        zero).  The struct points at this object's config: valid while ``self`` lives, reads ``batch_total`` at call time."""
        if not hasattr(self, "_epi_cfg"):
            self._epi_cfg = _lib.AgsLossConfig()
        C.memmove(C.byref(self._epi_cfg), C.byref(self.cfg), C.sizeof(_lib.AgsLossConfig))
        self._epi_cfg.gt_frame_index = ptr(gt_index)
        e = _lib.AgsLossEpilogue(C.pointer(self._epi_cfg), ptr(gt_rgb), ptr(gt_depth), ptr(buf.n_img), ptr(buf.d_rgb),
                                 ptr(buf.d_depth), ptr(self.msum), ptr(self.accum))
        e._keep = (self._epi_cfg, gt_rgb, gt_depth, gt_index)
        return e

    def stage2_batch(self, images: "_lib.AgsImages", gt_depth, buf: LossBuffers, views: int, gt_index=None) -> None:
        self.cfg.num_views = int(views)
        self.cfg.gt_frame_index = ptr(gt_index)
        try:
            _lib.check(_lib.load().ags_loss_stage2(C.byref(self.cfg), C.byref(images), ptr(buf.n_img), ptr(gt_depth),
                                                   ptr(self.msum), ptr(buf.d_normal), ptr(buf.d_depth), ptr(self.accum),
                                                   _lib.current_stream()), "ags_loss_stage2")
        finally:
            self.cfg.num_views = 0
            self.cfg.gt_frame_index = None

    def stage_frames(self, views: int, frame_index, all_view, all_proj, all_rgb, all_depth, dst_view, dst_proj,
                     dst_rgb, dst_depth) -> None:
        """Gather the sampled frames into the batch buffers and zero ``msum`` - one launch."""
        _lib.check(_lib.load().ags_stage_frames(int(views), self.h, self.w, ptr(frame_index), ptr(all_view), ptr(all_proj),
                                                ptr(all_rgb), ptr(all_depth), ptr(dst_view), ptr(dst_proj), ptr(dst_rgb),
                                                ptr(dst_depth), ptr(self.msum), _lib.current_stream()),
                   "ags_stage_frames")

    def finish(self, views: int, frame_index, frame_error, total_loss) -> None:
        """Per-frame errors -> ``frame_error[frame_index]``, total loss -> ``total_loss`` (a 0-d / 1-element
        tensor), accumulators zeroed - one launch instead of per_frame_errors + total_loss + begin_step."""
        _lib.check(_lib.load().ags_loss_finish(C.byref(self.cfg), ptr(self.accum), int(views), ptr(frame_index),
                                               ptr(frame_error), ptr(total_loss),
                                               _lib.current_stream()), "ags_loss_finish")

    def finish_next(self, views: int, frame_index, frame_error, total_loss, uniforms, n_weights: int, k: int,
                    first_random: int, all_view, all_proj, dst_view, dst_proj) -> None:
        """``finish`` + the NEXT iteration's draw (``uniforms``: n_weights numbers ~ U(0,1), None: the indices stay) +
        its ``stage_frames`` (matrices only, ``msum`` zeroed) - one launch (``ags_loss_finish_next``)."""
        nx = _lib.AgsNextIteration(ptr(uniforms), int(n_weights), int(k), int(first_random), int(views), ptr(all_view),
                                   ptr(all_proj), ptr(dst_view), ptr(dst_proj), ptr(self.msum))
        _lib.check(_lib.load().ags_loss_finish_next(C.byref(self.cfg), ptr(self.accum), int(views), ptr(frame_index),
                                                    ptr(frame_error), ptr(total_loss), C.byref(nx),
                                                    _lib.current_stream()), "ags_loss_finish_next")

    def total_loss(self) -> torch.Tensor:
        c, a, hw = self.cfg, self.accum.sum(0), float(self.h * self.w)
        b = float(c.batch_total)
        return (c.w_rgb * a[0] / (b * 3 * hw) + c.w_depth * a[1] / (b * hw) + c.w_cons * a[2] / (b * b * hw)
                + c.w_tv * a[3] / (b * 4 * hw))

    def per_frame_errors(self, n_views: int) -> torch.Tensor:
        hw = float(self.h * self.w)
        a = self.accum.sum(0)[4:4 + 2 * n_views].view(n_views, 2)
        return a[:, 0] / (3 * hw) + a[:, 1] / hw
