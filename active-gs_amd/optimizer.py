"""Fused Adam over the five Gaussian parameter tensors (C ABI ``ags_adam_step``).

Mirror of ``init_training`` + ``optimizer.step()`` at
/root/reference/mapping/gaussian_map.py:259-292,126: torch.optim.Adam semantics, eps 1e-15,
one learning rate per tensor (mean, scale, rotation, opacity, harmonic), state created at
construction (the reference rebuilds the optimizer at every keyframe)."""
from __future__ import annotations

import ctypes as C
from typing import Sequence

import torch

from . import _lib
from ._lib import ptr


class FusedAdam:
    def __init__(self, params: Sequence[torch.Tensor], lrs: Sequence[float], betas=(0.9, 0.999), eps: float = 1e-15,
                 zero: bool = True):
        """``zero=False``: the caller zeroes ``state_buffers()`` itself (``_lib.zero_many`` together with its other buffers)."""
        if len(params) != 5 or len(lrs) != 5:
            raise ValueError("FusedAdam takes the five map tensors (means, scales, rotations, opacities, harmonics)")
        for p in params:
            if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                raise RuntimeError("FusedAdam parameters must be contiguous float32 GPU tensors (no CPU fallback)")
        self.params = list(params)
        self.lrs = [float(x) for x in lrs]
        self.betas = betas
        self.eps = eps
        # The moments are private state (the reference's torch.optim.Adam keeps them per tensor and never looks at them,
        # gaussian_map.py:259-292), so they are laid out for the kernels: interleaved per surfel, (n, 28) floats
        # {exp_avg[14], exp_avg_sq[14]} (AgsAdamTensors.state_rows) - a member row of the row-set steps then costs one
        # 112-byte piece of state instead of ten 4..16-byte pieces of ten arrays.  exp_avg / exp_avg_sq stay available
        # as (strided) views of it.  Tensors that are not the five map tensors get the plain per-tensor layout.
        n = params[3].numel()
        self.state_rows = None
        if [p.numel() for p in params] == [3 * n, 3 * n, 4 * n, n, 3 * n] and n > 0:
            self.state_rows = (torch.zeros if zero else torch.empty)(n, 28, device=params[0].device, dtype=torch.float32)
            cols = ((0, 3), (3, 6), (6, 10), (10, 11), (11, 14))
            self.exp_avg = [self.state_rows[:, a:b].view(*p.shape) if p.dim() != 1 else self.state_rows[:, a]
                            for (a, b), p in zip(cols, params)]
            self.exp_avg_sq = [self.state_rows[:, 14 + a:14 + b].view(*p.shape) if p.dim() != 1 else self.state_rows[:, 14 + a]
                               for (a, b), p in zip(cols, params)]
        else:
            self.exp_avg = [torch.zeros_like(p) for p in params]
            self.exp_avg_sq = [torch.zeros_like(p) for p in params]
        self.step_count = 0
        # device-resident clock {int step; float step_size[5]; float inv_sqrt_bc2; int skipped; ...} for graph replay
        self.device_clock = (torch.zeros if zero else torch.empty)(16, device=params[0].device, dtype=torch.int32)
        self._clock_on_device = None   # where the ONE logical step counter currently lives (see use_clock)
        # optional raster_api.RowSet: update only the surfels this optimiser's views have shown
        # (exact: the others have zero gradient and zero moments, so the dense update is 0)
        self.touched = None
        # with ``touched``: gradient rows are zeroed as they are consumed (slabs the views add into atomically)
        self.zero_grad = False

    def state_buffers(self):
        """What a fresh optimiser needs zeroed (``zero=False``)."""
        return [self.device_clock] + ([self.state_rows] if self.state_rows is not None else self.exp_avg + self.exp_avg_sq)

    def use_clock(self, device: bool) -> None:
        """There is one logical step counter; it lives either in ``step_count`` (host, ``ags_adam_step``) or in
        ``device_clock`` (``ags_adam_step_device``, what a captured graph replays on).  Switching between the two
        carries the count over, so bias correction never restarts with warm moments (eager host-clock steps
        followed by ``capture()``, or the other way round).  Device -> host costs one 4-byte read-back."""
        if self._clock_on_device is device:
            return
        if device:
            if self.step_count:
                # the device clock keeps beta^step as running products (doubles at bytes 32..47; float betas as the kernels see them)
                b1, b2 = (float(torch.tensor(b, dtype=torch.float32).item()) for b in self.betas)
                pows = torch.tensor([b1 ** self.step_count, b2 ** self.step_count], dtype=torch.float64)
                self.device_clock[8:12] = pows.view(torch.int32).to(self.device_clock.device)
                self.device_clock[0] = self.step_count
        elif self._clock_on_device is not None:
            self.step_count = int(self.device_clock[0].item())
        self._clock_on_device = device

    def tick_args(self):
        """What ``raster_api.backward(adam_tick=...)`` needs to advance this optimizer's device clock."""
        return (self.device_clock, self.lrs, self.betas[0], self.betas[1])

    def tensors_struct(self, grads: Sequence[torch.Tensor], rows: "tuple | None" = None) -> "_lib.AgsAdamTensors":
        """The C-ABI view of this optimiser (also what ``raster_api.backward(fused_adam=...)`` takes).  ``rows`` =
        (begin, end): the view of the rows [begin, end) of the five map tensors only (pointers moved, element counts cut:
        the kernels see a smaller map) - the row chunks of a data-parallel rank's overlapped exchange."""
        t = _lib.AgsAdamTensors()
        if rows is not None:
            if self.state_rows is None or self.touched is not None:
                raise ValueError("a row range needs the five map tensors and no row set")
            a, b = int(rows[0]), int(rows[1])
            for k, width in enumerate((3, 3, 4, 1, 3)):
                gk = grads[k]
                if not gk.is_contiguous() or gk.numel() != self.params[k].numel():
                    raise ValueError("gradient arrays must be contiguous and shaped like the parameters")
                t.param[k] = ptr(self.params[k]) + 4 * a * width
                t.grad[k] = ptr(gk) + 4 * a * width
                t.numel[k] = (b - a) * width
                t.lr[k] = self.lrs[k]
            t.state_rows = ptr(self.state_rows) + 4 * 28 * a
            self._keep = grads
            return t
        for k in range(5):
            g = grads[k]
            if g.shape != self.params[k].shape and g.numel() == self.params[k].numel():
                g = g.reshape(self.params[k].shape)
            if not g.is_contiguous():
                g = g.contiguous()
            t.param[k] = ptr(self.params[k])
            t.grad[k] = ptr(g)
            if self.state_rows is None:
                t.exp_avg[k] = ptr(self.exp_avg[k])
                t.exp_avg_sq[k] = ptr(self.exp_avg_sq[k])
            t.numel[k] = self.params[k].numel()
            t.lr[k] = self.lrs[k]
        if self.touched is not None:
            t.touched = self.touched.c_struct()
            t.zero_grad = int(self.zero_grad)
        t.state_rows = ptr(self.state_rows)
        self._keep = grads
        return t

    def step(self, grads: Sequence[torch.Tensor], device_clock: bool = False, pre_ticked: bool = False) -> None:
        """One Adam update. With ``device_clock=True`` the step counter lives on the GPU
        (``ags_adam_step_device``), so the call can be captured in a hipGraph and replayed;
        ``pre_ticked`` says the step's last backward launch already advanced that clock."""
        lib = _lib.load()
        self.use_clock(device_clock)
        if not device_clock:
            self.step_count += 1
        t = self.tensors_struct(grads)
        stream = _lib.current_stream()
        if device_clock:
            _lib.check(lib.ags_adam_step_device(C.byref(t), self.betas[0], self.betas[1], self.eps,
                                                ptr(self.device_clock), int(pre_ticked), stream), "ags_adam_step_device")
        else:
            _lib.check(lib.ags_adam_step(C.byref(t), self.betas[0], self.betas[1], self.eps, self.step_count,
                                         stream), "ags_adam_step")

    def step_range(self, grads: Sequence[torch.Tensor], begin: int, end: int, device_clock: bool = True,
                   pre_ticked: bool = False, first: bool = True) -> None:
        """The Adam update of the rows [begin, end) only (same arithmetic per element as ``step``).  The chunks of one
        optimisation step share ONE tick of the clock: pass ``first=True`` for the chunk that comes first."""
        lib = _lib.load()
        self.use_clock(device_clock)
        if not device_clock and first:
            self.step_count += 1
        t = self.tensors_struct(grads, rows=(begin, end))
        stream = _lib.current_stream()
        if device_clock:
            _lib.check(lib.ags_adam_step_device(C.byref(t), self.betas[0], self.betas[1], self.eps, ptr(self.device_clock),
                                                int(pre_ticked or not first), stream), "ags_adam_step_device")
        else:
            _lib.check(lib.ags_adam_step(C.byref(t), self.betas[0], self.betas[1], self.eps, self.step_count, stream),
                       "ags_adam_step")

    def step_gathered(self, grads: Sequence[torch.Tensor], segments: torch.Tensor, world: int, capacity: int,
                      slot_table: torch.Tensor, pre_ticked: bool = False) -> None:
        """Adam over the ``touched`` rows with every row's gradient summed, in rank order, straight from
        the all-gathered row segments (``ags_rows_index`` filled ``slot_table``); device clock only."""
        t = self.tensors_struct(grads)
        _lib.check(_lib.load().ags_adam_step_gathered(C.byref(t), ptr(segments), int(world), int(capacity), ptr(slot_table),
                                                      self.betas[0], self.betas[1], self.eps, ptr(self.device_clock),
                                                      int(pre_ticked), _lib.current_stream()),
                   "ags_adam_step_gathered")

