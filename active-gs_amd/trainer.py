"""Host-side mirror of the optimisation inner loop of ``GaussianMap.train()``
(/root/reference/mapping/gaussian_map.py:66-127) over the C ABI:

    get_attr() activations (:529-581)      -> ags_activate
    render_view_all(require_grad=True)     -> ags_forward per view (operations.py:854)
    total_loss.backward() (:125)           -> ags_backward per view, accumulating in place,
                                              then ags_activate_backward
    optimizer.step() (:126, Adam :259-292) -> ags_adam_step on the raw parameters

Views are independent until the loss mean (:113-124), so with ``torch.distributed``
initialised each rank renders its own views and ONE collective precedes the replicated Adam step
(SURVEY.md §8e): an all-gather of the rows each rank's views have shown (``RowExchange``; 64 B per
row, ~1 MB per rank at 200 k surfels @1200x680) or, where that would not be smaller, an
all-reduce(sum) of the contiguous 14*N-float gradient slab (11 MB).
"""
from __future__ import annotations

import contextlib
import ctypes as C
from typing import Callable, Optional, Sequence

import torch

from . import _lib, raster_api as api
from .dist_util import all_reduce_
from ._lib import ptr

DEFAULT_LRS = dict(mean=5e-4, scale=1e-2, rotation=5e-4, opacity=1e-2, harmonic=1e-4)  # incremental.yaml:27-32


def _pad4(x: int) -> int:
    return (x + 3) & ~3


class GradSlab:
    """One contiguous buffer for the five gradients (segments start on 16-byte
    boundaries) so the data-parallel exchange is a single collective."""

    def __init__(self, n: int, device, zero: bool = True):
        sizes = [3 * n, 3 * n, 4 * n, n, 3 * n]
        offs, o = [], 0
        for s in sizes:
            offs.append(o)
            o += _pad4(s)
        self.flat = (torch.zeros if zero else torch.empty)(max(o, 4), device=device, dtype=torch.float32)
        v = [self.flat[a:a + s] for a, s in zip(offs, sizes)]
        self.grads = api.GaussianGrads(v[0].view(n, 3), v[1].view(n, 3), v[2].view(n, 4), v[3], v[4].view(n, 3))

    def as_list(self):
        g = self.grads
        return [g.means3D, g.scales, g.rotations, g.opacities, g.colors]


class RowExchange:
    """Sparse gradient exchange of the view-parallel step (include/ags_raster.h: ``ags_rows_pack`` or
    ``AgsGaussianGrads.pack_segment``, ``ags_rows_index`` + ``ags_adam_step_gathered``, or
    ``ags_rows_unpack``): every rank ships the gradient rows of its own sticky row set as a fixed
    size segment, ONE all-gather moves the segments, and every rank sums them in rank order -
    bit-identical sums on every rank - over the union of the rows, which is what the replicated
    Adam steps over.  ``capacity`` (rows per segment) is agreed
    once with ``agree()``; ``overflowed()`` says whether a rank has outgrown it since."""

    GROWTH, SLACK = 1.5, 1024      # capacity = GROWTH x the largest rank's row count + SLACK
    # after the all-gather: "indexed" = ags_rows_index + ags_adam_step_gathered (two launches for any number of
    # ranks, no slab round trip); "unpack" = one ags_rows_unpack per rank, then the row-set Adam over the slab
    TAIL = "indexed"

    def __init__(self, n: int, grads: Sequence[torch.Tensor], device, pg):
        self.n, self.pg, self.device = n, pg, device
        self.world = torch.distributed.get_world_size(pg)
        self.rank = torch.distributed.get_rank(pg)
        self.union = api.RowSet(n, device)
        self.capacity = None                      # None: not agreed yet; 0: the dense all-reduce is smaller
        self.agreements = 0                       # how often agree() ran (1 + the number of regrowths)
        self._grads = (C.c_void_p * 5)(*[ptr(g) for g in grads])
        self._keep = list(grads)
        self.send = self.recv = None

    def agree(self, local_rows: int, slab_floats: int) -> int:
        """Collective (host-synchronous; on the first step of an optimiser and again whenever a rank has
        outgrown the segment): capacity = GROWTH x the largest rank's row count + SLACK, never smaller
        than before; 0 if gathering the segments would not move clearly fewer bytes than all-reducing the
        dense slab."""
        t = torch.tensor([int(local_rows)], device=self.device, dtype=torch.int64)
        all_reduce_(t, torch.distributed.ReduceOp.MAX, self.pg)
        cap = min(self.n, max(int(self.GROWTH * int(t.item())) + self.SLACK, (self.capacity or 0) + 1))
        self.agreements += 1
        seg = int(_lib.load().ags_rows_segment_floats(cap))
        # bytes a rank receives: (world-1) segments against 2 (world-1)/world slabs of a ring all-reduce;
        # the row path also pays world unpack launches, so it has to win by a margin
        if seg * self.world >= 1.5 * slab_floats:
            self.capacity = 0
            return 0
        self.capacity = cap
        self.send = torch.zeros(seg, device=self.device, dtype=torch.float32)
        self.recv = torch.zeros(self.world, seg, device=self.device, dtype=torch.float32)
        self.slot_table = torch.zeros(self.n * self.world, device=self.device, dtype=torch.int32)   # kept zeroed by the gathered Adam
        return cap

    def pack(self, rows: "api.RowSet") -> None:
        r = rows.c_struct()
        _lib.check(_lib.load().ags_rows_pack(C.byref(r), C.byref(self._grads), self.capacity, ptr(self.send),
                                             _lib.current_stream()), "ags_rows_pack")

    def gather(self) -> None:
        if torch.distributed.get_backend(self.pg) == "nccl":     # RCCL: one all-gather over xGMI
            torch.distributed.all_gather_into_tensor(self.recv.view(-1), self.send, group=self.pg)
        else:
            # transports without a device all-gather (gloo, used by the tests): gathered on the host
            mine = self.send.cpu()
            parts = [torch.empty_like(mine) for _ in range(self.world)]
            torch.distributed.all_gather(parts, mine, group=self.pg)
            self.recv.copy_(torch.stack(parts))

    def unpack(self) -> None:
        lib, u = _lib.load(), self.union.c_struct()
        stream = _lib.current_stream()
        for r in range(self.world):                               # rank order: same sums everywhere
            _lib.check(lib.ags_rows_unpack(ptr(self.recv[r]), self.capacity, C.byref(self._grads), C.byref(u), stream),
                       "ags_rows_unpack")

    def index(self) -> None:
        u = self.union.c_struct()
        _lib.check(_lib.load().ags_rows_index(ptr(self.recv), self.world, self.capacity, ptr(self.slot_table), C.byref(u),
                                              _lib.current_stream()), "ags_rows_index")

    def overflowed(self) -> bool:
        """Host-synchronous: has any rank needed more rows than the agreed capacity in the last exchange?
        Every rank reads the same gathered headers, so every rank gets the same answer."""
        if not self.capacity:
            return False
        needed = self.recv[:, 1].view(torch.int32)
        return bool((needed > self.capacity).any().item())

    def needed_rows(self) -> int:
        """Host-synchronous: the largest row count any rank reported in the last exchange."""
        return int(self.recv[:, 1].view(torch.int32).max().item()) if self.capacity else 0

    def restore_own(self) -> None:
        """Undo this rank's pack: add its own segment back into the gradient arrays (the rows that did not fit
        never left them), so the arrays hold the rank's complete gradient again."""
        lib, u = _lib.load(), self.union.c_struct()
        _lib.check(lib.ags_rows_unpack(ptr(self.send), self.capacity, C.byref(self._grads), C.byref(u),
                                       _lib.current_stream()), "ags_rows_unpack")

    def reset(self) -> None:
        self.union.reset()


class SurfelTrainer:
    """Raw map parameters + fused train step. ``raw`` holds means (N,3), scales (N,3),
    rotations (N,4), opacities (N), harmonics (N,1,3), confidences (N) on the GPU."""

    def __init__(self, raw: dict, lrs: Optional[dict] = None, scale_factor: float = 0.01, max_scale: float = 0.05,
                 eps: float = 1e-15, process_group=None, binning_mode: int = api.BIN_DIRECT,
                 fused_activations: bool = True, sparse_rows: bool = True, tuning=None, view_streams: Optional[int] = None):
        from .optimizer import FusedAdam
        if view_streams is not None:
            self.VIEW_STREAMS = max(1, int(view_streams))
        self.tuning = tuning     # _lib.AgsTuning handed over with every workspace this trainer makes (None: process default)
        self._tuning_pinned = tuning is not None      # a caller's selection is not adapted (_adapt_kernels)
        lrs = {**DEFAULT_LRS, **(lrs or {})}
        self.raw = {k: v.contiguous() for k, v in raw.items()}
        dev = self.raw["means"].device
        if dev.type != "cuda":
            raise RuntimeError("SurfelTrainer needs GPU tensors: the rasterizer has no CPU fallback")
        self.n = self.raw["means"].shape[0]
        self.device = dev
        self.scale_factor, self.max_scale = scale_factor, max_scale
        self.act_scales = torch.empty_like(self.raw["scales"])
        self.act_rot = torch.empty_like(self.raw["rotations"])
        self.act_opac = torch.empty_like(self.raw["opacities"])
        self.slab = GradSlab(self.n, dev)
        self.params = [self.raw["means"], self.raw["scales"], self.raw["rotations"], self.raw["opacities"],
                       self.raw["harmonics"]]
        self.optim = FusedAdam(self.params, [lrs["mean"], lrs["scale"], lrs["rotation"], lrs["opacity"],
                                             lrs["harmonic"]], eps=eps)
        self.pg = process_group
        self.binning_mode = binning_mode
        # True: the per-Gaussian kernels apply the activations and their chain rule in registers
        # (AgsGaussians.raw_params); False: separate ags_activate / ags_activate_backward launches
        self.fused_activations = fused_activations
        # Sticky set of the surfels this optimiser's views have shown (api.RowSet): the per-Gaussian
        # backward and Adam launch work for those rows only.  Lossless (untouched rows have zero
        # gradient and zero moments -> a zero Adam update).  With data parallelism the ranks exchange
        # exactly those rows (RowExchange) and Adam steps over the union of all ranks' sets.
        self.rows = api.RowSet(self.n, dev) if sparse_rows else None
        self.exchange = None
        if self.rows is not None:
            self.slab.flat.zero_()
            if self._distributed():
                self.exchange = RowExchange(self.n, self.slab.as_list(), dev, self.pg)
                self.optim.touched, self.optim.zero_grad = self.exchange.union, True
            else:
                self.optim.touched = self.rows
        self._state = {}
        self._no_grads = api.GaussianGrads(None, None, None, None, None)
        # steps since the last look at the device-side overflow notes (refused exchange steps, workspace
        # overflow), with what is needed to repeat them: see check_overflow()
        self._pending = []
        self.exchange_regrowths = 0
        # software-pipelined steps (step(..., next_cam=...)): (ForwardState, Camera) whose per-Gaussian stage the
        # previous step's last launch has already run, or None
        self._prepared = None
        self._lanes = []
        self._rows_hint = 0      # members of the row set when last looked at (+ slack): sizes the pipelined launch

    CHECK_EVERY = 16     # optimisation steps between two reads of the overflow notes (one 64-byte D2H each)
    # Views of a multi-view step can run on several streams (their launches overlap each other's ramps and tails: config 4's
    # four 1200x680 views 1.82 -> 1.53 ms per step on 4 streams; 2: 1.57, 3: 1.54).  OPT-IN (``view_streams=`` / this
    # attribute): the caller's ``image_grads(v, st)`` then runs under the view's stream, so it must not
    # share scratch buffers between views and must consume what it allocates on that stream - a callback written for the
    # one-stream step (the default) need not know any of this.
    VIEW_STREAMS = 1
    MAX_ROW_VIEWS = 16       # AGS_MAX_ROW_VIEWS (include/ags_raster.h): views one ags_backward_rows launch joins
    MULTI_VIEW_ROWS = True   # several views per step: one per-Gaussian backward launch for all of them (ags_backward_rows)
    # Data-parallel ranks that exchange the DENSE gradient slab (row sets covering most of the map: configuration 4): the
    # per-Gaussian backward, the all-reduce and the Adam update are cut into this many row chunks; chunk k's all-reduce
    # runs on a communication stream under chunk k + 1's chain rule and the Adam update of the chunks that have arrived.
    # Every element is summed over the ranks exactly as in one all-reduce, so the replicas stay bit-identical; 1 = one
    # all-reduce of the whole slab behind the whole backward.
    DENSE_CHUNKS = 4
    DP_FORCE = False             # tests / measurements: take the data-parallel step in a one-rank process group too
    GRAPH_COLLECTIVES = True     # False: never record collectives inside the step graph
    CULL_ADAPT = True            # False: never switch the per-Gaussian forward kernel from what the views show
    DENSE_CHUNK_MIN_ROWS = 1 << 14

    def reset_optimizer(self) -> None:
        """What the reference does at the start of every ``train()`` call (``init_training``,
        gaussian_map.py:259-292): a fresh Adam.  Moments, step counter, gradient slab and the sticky
        row set are cleared together (the row set's losslessness rests on exactly that)."""
        self.check_overflow()              # settle the previous optimiser's steps first
        self._drop_prepared()              # a prepared pass has inserted its surfels into the row set that is cleared below
        for t in self.optim.exp_avg + self.optim.exp_avg_sq:
            t.zero_()
        self.optim.step_count = 0
        self.optim.device_clock.zero_()
        self.slab.flat.zero_()
        if self.rows is not None:
            self.rows.reset()
        if self.exchange is not None:
            self.exchange.reset()

    # -- pieces --------------------------------------------------------------------------
    def _act_struct(self) -> _lib.AgsActivation:
        return _lib.AgsActivation(self.n, self.scale_factor, self.max_scale, ptr(self.raw["scales"]),
                                  ptr(self.raw["rotations"]), ptr(self.raw["opacities"]))

    def activate(self) -> api.Gaussians:
        a = self._act_struct()
        _lib.check(_lib.load().ags_activate(C.byref(a), ptr(self.act_scales), ptr(self.act_rot), ptr(self.act_opac),
                                            _lib.current_stream()), "ags_activate")
        return api.Gaussians(self.raw["means"], self.act_scales, self.act_rot, self.act_opac,
                             self.raw["harmonics"].view(self.n, 3), self.raw["confidences"])

    def activate_backward(self) -> None:
        a = self._act_struct()
        g = self.slab.grads
        _lib.check(_lib.load().ags_activate_backward(C.byref(a), ptr(g.scales), ptr(g.rotations), ptr(g.opacities),
                                                     _lib.current_stream()), "ags_activate_backward")

    def state_for(self, h: int, w: int, max_instances: int, slot: int = 0) -> api.ForwardState:
        key = (h, w, slot)
        st = self._state.get(key)
        if st is None or st.max_instances < max_instances or st.radii.shape[0] != self.n:
            st = api.alloc_state(self.n, h, w, max_instances, self.device, self.binning_mode, tuning=self.tuning)
            self._state[key] = st
        return st

    # -- one optimisation step -----------------------------------------------------------
    def _distributed(self) -> bool:
        if not (torch.distributed.is_available() and torch.distributed.is_initialized()):
            return False
        # DP_FORCE (tests): take the data-parallel path in a one-rank group too, which is how the RCCL
        # collectives and their capture into the step graph are exercised on a single-GPU box
        return torch.distributed.get_world_size(self.pg) > 1 or self.DP_FORCE

    def gaussians(self) -> api.Gaussians:
        if self.fused_activations:
            return api.Gaussians(self.raw["means"], self.raw["scales"], self.raw["rotations"], self.raw["opacities"],
                                 self.raw["harmonics"].view(self.n, 3), self.raw["confidences"], raw_params=True,
                                 scale_factor=self.scale_factor, max_scale=self.max_scale)
        return self.activate()

    def _drop_prepared(self) -> None:
        """Leave the pipeline: the prepared pass has taken key slots in its workspace - clear them (the status block's
        sticky overflow notes stay: ``check_overflow`` has not looked at the last steps yet)."""
        if self._prepared is not None:
            st, _ = self._prepared
            api.discard_pass(st, self.n, st.rgb.shape[-2], st.rgb.shape[-1])
            self._prepared = None

    def _view_streams(self, views: int):
        """Streams the views of a multi-view step are spread over (VIEW_STREAMS > 1): a view's four launches depend
        on nothing another view of the step produces until the per-Gaussian backward joins them."""
        k = min(int(self.VIEW_STREAMS), views)
        if k <= 1:
            return []
        while len(self._lanes) < k:
            self._lanes.append(torch.cuda.Stream())
        return self._lanes[:k]

    def _local_pass(self, cams, image_grads, max_instances, tick: bool = False, fuse_adam: bool = False,
                    next_cam: Optional[api.Camera] = None) -> bool:
        """Forward+backward of this rank's views; with ``tick`` the last backward also advances
        the Adam device clock (returns whether it did); with ``fuse_adam`` that last backward
        performs the optimiser step itself (``self.adam_fused`` says whether it did).  ``next_cam`` (with a fused
        optimiser step and one-pass binning): that launch also runs the per-Gaussian stage of the NEXT step's first
        view, which then starts at its tile sort (software pipelining across steps)."""
        g = self.gaussians()
        ticked = False
        self.adam_fused = False
        fuse_adam = fuse_adam and tick and self.rows is not None and self.fused_activations
        pipeline = next_cam is not None and fuse_adam and self.binning_mode == api.BIN_DIRECT and len(cams) > 0
        # data-parallel step with an agreed segment size: the rank's last backward writes the exchange segment itself
        x = self.exchange
        pack = (x.send, x.capacity) if (x is not None and x.capacity and self.fused_activations and len(cams) > 0) else None
        self._packed = pack is not None
        # several views per step on the row-set path: every view keeps its gradient records in its own workspace and ONE
        # launch (ags_backward_rows) turns them all into parameter gradients - the member list is walked once, a row's
        # inputs are loaded once, the views' gradients meet in registers (no read-modify-write of the gradient rows per
        # view), and the optimiser step / exchange segment is its tail
        if len(cams) > 1 and self.rows is not None and self.MULTI_VIEW_ROWS:
            self._drop_prepared()
            done = []
            lanes = self._view_streams(len(cams))
            main = torch.cuda.current_stream()
            # (every view's workspace exists BEFORE the streams fork: a workspace made inside the loop is initialised by a
            # launch on the main stream that a view stream, already told to wait for the main stream's EARLIER position,
            # would not wait for - its forward could then meet tile counters that are not zeroed yet)
            states = [self.state_for(cam.image_height, cam.image_width, max_instances, slot=v) for v, cam in enumerate(cams)]
            for s in lanes:
                s.wait_stream(main)
            for v, cam in enumerate(cams):
                st = states[v]
                with torch.cuda.stream(lanes[v % len(lanes)]) if lanes else contextlib.nullcontext():
                    api.forward(cam, g, st, touched=self.rows)
                    d = image_grads(v, st)
                    last = tick and v == len(cams) - 1
                    api.backward(cam, g, st, *d, adam_tick=self.optim.tick_args() if last else None, defer_rows=True)
                done.append((cam, st))
                ticked |= last
            for s in lanes:
                main.wait_stream(s)
            fused = (self.optim.tensors_struct(self.slab.as_list()), self.optim.eps) if (ticked and fuse_adam) else None
            # (one launch joins at most MAX_ROW_VIEWS views: a rank with more - all 32 views of configuration 4 on one GPU -
            # sums the earlier groups into the gradient rows and gives the tail to the last group)
            groups = [done[k:k + self.MAX_ROW_VIEWS] for k in range(0, len(done), self.MAX_ROW_VIEWS)]
            for j, grp in enumerate(groups[:-1]):
                api.backward_rows(grp, g, self.slab.grads, self.rows, accumulate=(j > 0))
            many = len(groups) > 1
            api.backward_rows(groups[-1], g, self._no_grads if (fused is not None and not many) else self.slab.grads, self.rows,
                              accumulate=many, adam_clock=self.optim.tick_args() if fused is not None else None,
                              fused_adam=fused, pack=pack)
            self.adam_fused = fused is not None
            if not self.fused_activations:
                self.activate_backward()
            return ticked
        for v, cam in enumerate(cams):
            st = self.state_for(cam.image_height, cam.image_width, max_instances)
            resume = self._prepared is not None and self._prepared[0] is st and self._prepared[1] is cam and v == 0
            if not resume:
                self._drop_prepared()
            self._prepared = None
            api.forward(cam, g, st, touched=self.rows, resume=resume)
            d = image_grads(v, st)
            final = v == len(cams) - 1
            last = tick and final
            fused = (self.optim.tensors_struct(self.slab.as_list()), self.optim.eps) if (last and fuse_adam) else None
            # a fused step over ONE view: the gradient never leaves the registers (no slab rows written at all)
            grads = self._no_grads if (fused is not None and v == 0) else self.slab.grads
            nxt = None
            if pipeline and fused is not None:
                nxt = (next_cam, self.state_for(next_cam.image_height, next_cam.image_width, max_instances), self._rows_hint)
            api.backward(cam, g, st, *d, grads=grads, accumulate=(v > 0),
                         adam_tick=self.optim.tick_args() if last else None, touched=self.rows, fused_adam=fused,
                         pack=pack if final else None, next_view=nxt)
            if nxt is not None:
                self._prepared = (nxt[1], nxt[0])
            ticked |= last
            self.adam_fused |= fused is not None
        if len(cams) == 0:
            self.slab.flat.zero_()
        if not self.fused_activations:
            self.activate_backward()
        return ticked

    def step(self, cams: Sequence[api.Camera], image_grads: Callable, max_instances: int,
             device_clock: bool = True, next_cam: Optional[api.Camera] = None) -> None:
        """``cams``: this rank's views. ``image_grads(view_index, state)`` returns the five
        image gradients (d_rgb, d_normal, d_depth, d_opacity, d_confidence; None = zero)
        for that view, already divided by the GLOBAL number of views where the loss is a
        batch mean.  With ``view_streams`` > 1 (opt-in) and several views per step it is called under the stream that
        view runs on: what it enqueues for different views may overlap, so it must not share scratch buffers between
        views.  Asynchronous except for the collective and, every ``CHECK_EVERY`` steps, one small
        read-back (``check_overflow``).  ``device_clock`` (default): the Adam step counter lives on the GPU,
        the same clock ``capture()`` replays on; False = the host-side counter of ``ags_adam_step``.
        ``next_cam`` (single rank): the first view of the NEXT step, its matrices already in place - this step's
        last launch then also runs that view's per-Gaussian stage (``ags_backward_fused_next``) and the next
        ``step`` whose first camera is that object starts at its tile sort."""
        self.optim.use_clock(device_clock)
        self._step_once(cams, image_grads, max_instances, device_clock, next_cam)
        self._pending.append((cams, image_grads, max_instances, device_clock))
        if len(self._pending) >= self.CHECK_EVERY:
            self.check_overflow()

    def _dense_chunked(self, cams) -> bool:
        return (self._distributed() and self.rows is None and self.fused_activations and self.MULTI_VIEW_ROWS
                and self.DENSE_CHUNKS >= 1)

    def _dense_step(self, cams, image_grads, max_instances, device_clock: bool) -> None:
        """The data-parallel step with the dense slab: blend passes of this rank's views, then per ROW CHUNK the
        per-Gaussian backward of all views (``ags_backward_rows`` without a row set), the chunk's all-reduce on the
        communication stream, and its Adam update once the sum has arrived."""
        done, ticked = self._dense_blend(cams, image_grads, max_instances, device_clock)
        self._dense_tail(done, ticked, device_clock)

    def _dense_blend(self, cams, image_grads, max_instances, device_clock: bool):
        """forward + blend backward of this rank's views; every view keeps its gradient records in its own workspace"""
        g = self.gaussians()
        self._drop_prepared()
        self.adam_fused, self._packed = False, False
        done, ticked = [], False
        main = torch.cuda.current_stream()
        lanes = self._view_streams(len(cams))
        states = [self.state_for(cam.image_height, cam.image_width, max_instances, slot=v) for v, cam in enumerate(cams)]   # before the fork (see _local_pass)
        for s_ in lanes:
            s_.wait_stream(main)
        for v, cam in enumerate(cams):
            st = states[v]
            with torch.cuda.stream(lanes[v % len(lanes)]) if lanes else contextlib.nullcontext():
                api.forward(cam, g, st)
                d = image_grads(v, st)
                last = device_clock and v == len(cams) - 1
                api.backward(cam, g, st, *d, adam_tick=self.optim.tick_args() if last else None, defer_rows=True)
            done.append((cam, st))
            ticked |= last
        for s_ in lanes:
            main.wait_stream(s_)
        return done, ticked

    def _dense_tail(self, done, ticked: bool, device_clock: bool) -> None:
        """per row chunk: chain rule of all views -> all-reduce (communication stream) -> Adam"""
        g = self.gaussians()
        main = torch.cuda.current_stream()
        n = self.n
        K = max(1, min(int(self.DENSE_CHUNKS), n // self.DENSE_CHUNK_MIN_ROWS))
        bounds = [(n * c // K) & ~63 if 0 < c < K else (0 if c == 0 else n) for c in range(K + 1)]
        segs = [t.view(-1) for t in self.slab.as_list()]
        if getattr(self, "_comm", None) is None:
            self._comm = torch.cuda.Stream()
        # ``tail_probe`` (a list, set by a measuring caller on EAGER steps): timing events of this tail are appended -
        # how long the chain rule, every chunk's all-reduce and the Adam updates took and how long the main stream
        # waited for sums that had not arrived (``tail_timeline`` turns them into milliseconds)
        probe = getattr(self, "tail_probe", None)
        if probe is not None and torch.cuda.is_current_stream_capturing():
            probe = None

        def stamp(stream):
            e = torch.cuda.Event(enable_timing=True)
            e.record(stream)
            return e
        rec = dict(start=stamp(main), rows=[], ar=[], adam=[], chunks=K, bounds=bounds) if probe is not None else None
        groups = [done[k:k + self.MAX_ROW_VIEWS] for k in range(0, len(done), self.MAX_ROW_VIEWS)]
        arrived = []
        for c in range(K):
            a, b = bounds[c], bounds[c + 1]
            for j, grp in enumerate(groups):      # (more than MAX_ROW_VIEWS views on a rank: later groups add to the chunk)
                api.backward_rows(grp, g, self.slab.grads, None, row_range=(a, b), accumulate=(j > 0))
            if not done and c == 0:
                self.slab.flat.zero_()            # a rank without views this step contributes zeros
            if rec is not None:
                rec["rows"].append(stamp(main))
            if K == 1:
                t0 = stamp(main) if rec is not None else None
                all_reduce_(self.slab.flat, group=self.pg)
                if rec is not None:
                    rec["ar"].append((t0, stamp(main)))
                continue
            ready = torch.cuda.Event()
            ready.record(main)
            with torch.cuda.stream(self._comm):
                self._comm.wait_event(ready)
                t0 = stamp(self._comm) if rec is not None else None
                for seg, width in zip(segs, (3, 3, 4, 1, 3)):
                    all_reduce_(seg[a * width:b * width], group=self.pg)
                ev = torch.cuda.Event()
                ev.record(self._comm)
                if rec is not None:
                    rec["ar"].append((t0, stamp(self._comm)))
            arrived.append(ev)
        for c in range(K):
            if K > 1:
                main.wait_event(arrived[c])
            t0 = stamp(main) if rec is not None else None
            self.optim.step_range(self.slab.as_list(), bounds[c], bounds[c + 1], device_clock=device_clock,
                                  pre_ticked=ticked, first=(c == 0))
            if rec is not None:
                rec["adam"].append((t0, stamp(main)))
        if rec is not None:
            probe.append(rec)

    @staticmethod
    def tail_timeline(rec: dict) -> dict:
        """Host-synchronous: one ``tail_probe`` record in milliseconds.  ``exposed_ms`` = what the main stream spent
        between the last chunk's chain rule and the last Adam update that was NOT an Adam update: the wait for sums
        still on the wire (with one chunk: the whole all-reduce)."""
        torch.cuda.synchronize()
        rows_ms = rec["start"].elapsed_time(rec["rows"][-1])
        ar = [a.elapsed_time(b) for a, b in rec["ar"]]
        adam = [a.elapsed_time(b) for a, b in rec["adam"]]
        tail_ms = rec["start"].elapsed_time(rec["adam"][-1][1])
        return dict(chunks=rec["chunks"], rows_ms=rows_ms, all_reduce_ms=ar, all_reduce_sum_ms=sum(ar), adam_ms=sum(adam),
                    tail_ms=tail_ms, exposed_ms=max(0.0, tail_ms - rows_ms - sum(adam)))

    def _step_once(self, cams, image_grads, max_instances, device_clock, next_cam=None) -> None:
        dist_on = self._distributed()
        if dist_on and self._dense_chunked(cams):
            self._dense_step(cams, image_grads, max_instances, device_clock)
            return
        ticked = self._local_pass(cams, image_grads, max_instances, tick=device_clock, fuse_adam=not dist_on,
                                  next_cam=None if dist_on else next_cam)
        if dist_on:
            # ``exchange_probe`` (a list, set by a measuring caller on eager steps): an event pair around the exchange -
            # in this form of the step the collective sits on the main stream, all of it is exposed
            probe = getattr(self, "exchange_probe", None)
            if probe is not None and not torch.cuda.is_current_stream_capturing():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                self._exchange_gradients(device_clock)
                e1.record()
                probe.append((e0, e1))
            else:
                self._exchange_gradients(device_clock)
        self._optimizer_step(device_clock, ticked)

    # -- overflow: noted on the device, looked at every few steps, repaired without losing a gradient row ----
    def refused_steps(self) -> int:
        """Host-synchronous: optimisation steps the gathered Adam refused since the counter was last cleared
        (a rank's row set had outgrown the agreed exchange segment; ``ags_adam_step_gathered``)."""
        return int(self.optim.device_clock[7].item())

    def workspace_overflows(self) -> dict:
        """Host-synchronous: per view size, (overflowed passes, peak tile instances) since the workspace was made."""
        out = {}
        vis, views = 0, 0
        for key, st in self._state.items():
            info = api.read_status(st)
            if info["overflow_passes"]:
                out[key] = (info["overflow_passes"], info["peak_instances"])
            vis, views = vis + info["num_visible"], views + 1
        if views:
            self._adapt_kernels(vis / views)
        return out

    # Which per-Gaussian forward kernel: the cull-first form (cull on the means alone, project the survivors on full waves:
    # ags_k_preprocess_cull) is the library's choice from 2^20 rows up; below that the plain kernel's single pass is the
    # shorter chain - unless a view shows little of the map: at C2 (200 k rows in random order, 7 % visible) nearly every
    # wave of the plain kernel still has a visible lane and runs the ~900-instruction projection at a tenth of its lanes,
    # and the cull-first kernel is 1.1-3.9 us faster (step -3.5 %, profiles/r05_h_preprocess_c2.md); a mapper-grown map
    # (neighbours in a wave: the visible rows fill few waves well) is 3.5-5 % SLOWER with it although a view shows only
    # 14-18 % of it (profiles/r05_i_mapper_cull.txt, r05_ag_mapper_cull_sizes.txt) - what decides is how the visible rows are
    # spread over the waves, which the host cannot see; the visible FRACTION is the proxy it has.  The trainer reads its
    # views' status blocks every CHECK_EVERY steps anyway: it asks for the cull-first kernel while the views show less than
    # CULL_FIRST_BELOW of the rows - a tenth: below what a map grown by the mapper shows from inside, above C2's 7 %.
    # Hysteresis: switched ON below CULL_FIRST_ON of the rows, OFF again only above CULL_FIRST_OFF - a trainer whose views sit
    # at the threshold does not flip kernels at every look (both kernels give the same bits; a captured graph keeps the one
    # it was recorded with, `cull_first_kernel()` says which one the next launch takes).
    CULL_FIRST_BELOW, CULL_FIRST_MIN_ROWS = 0.10, 1 << 16
    CULL_FIRST_ON, CULL_FIRST_OFF = 0.08, 0.12

    def cull_first_kernel(self) -> bool:
        """Does the next per-Gaussian forward launch of this trainer take the cull-first kernel?"""
        t = self.tuning if self.tuning is not None else _lib.default_tuning()
        thr = int(t.cull_first_min_n)
        return self.fused_activations and (thr == 1 or (thr == 0 and self.n >= (1 << 20)) or (thr > 1 and self.n >= thr))

    def _adapt_kernels(self, visible_per_view: float) -> None:
        if self.n < self.CULL_FIRST_MIN_ROWS or not self.fused_activations or getattr(self, "_tuning_pinned", False):
            return
        if _lib.cull_choice_pinned or not self.CULL_ADAPT:
            return                                    # the process's explicit choice stands
        now = 1 if (self.tuning is not None and self.tuning.cull_first_min_n == 1) else 0
        frac = visible_per_view / max(self.n, 1)
        want = 1 if frac < self.CULL_FIRST_ON else (0 if frac > self.CULL_FIRST_OFF else now)   # AgsTuning.cull_first_min_n: 1 = always, 0 = default
        if self.tuning is None:
            if want == 0:
                return
            self.tuning = _lib.make_tuning()          # this trainer's own copy of the process default
            for st in self._state.values():
                st.tuning = self.tuning
        if self.tuning.cull_first_min_n in (0, 1):    # (a caller's explicit threshold is left alone)
            self.tuning.cull_first_min_n = want

    def regrow_exchange(self) -> int:
        """Collective.  If the gathered Adam has refused steps, agree on a larger segment (or fall back to the
        dense all-reduce) and clear the counter.  Returns the number of refused steps: the caller repeats
        them (``check_overflow`` does; a captured replay has to be re-captured first, its segment size is
        part of the graph)."""
        k = self.refused_steps() if self.exchange is not None and self.exchange.capacity else 0
        if k:
            self.exchange_regrowths += 1
            self.optim.device_clock[7] = 0
            self.slab.flat.zero_()       # rows that did not fit the segment kept their totals here (ags_rows_pack contract)
            if self.exchange.agree(int(self.rows.count.item()), self.slab.flat.numel()) == 0:
                self._go_dense()
        return k

    def _go_dense(self) -> None:
        # from the next pass on everything is dense: dense per-Gaussian backward (overwrites the whole slab),
        # all-reduce of the slab, dense Adam (rows outside the old union have zero moments: a zero update)
        self.exchange, self.rows = None, None
        self.optim.touched, self.optim.zero_grad = None, False

    def check_overflow(self) -> int:
        """Settle the steps taken since the last call (host-synchronous, collective when ranks are present):
        * a view that needed more tile instances than ``max_instances`` in any of them raises (its tile lists
          were truncated: the gradients of those steps were wrong);
        * steps the row exchange refused because a rank had outgrown the agreed segment are repeated after
          agreeing on a larger one - no gradient row is dropped, the replicas never diverge.
        Returns the number of repeated steps."""
        pending, self._pending = self._pending, []
        if self.rows is not None:
            self._rows_hint = int(int(self.rows.count.item()) * 1.25) + 1024
        bad = self.workspace_overflows()
        if bad:
            raise RuntimeError(f"a view outgrew its rasterizer workspace {bad} (view size -> overflowed passes, "
                               "instances needed): raise max_instances; the last steps used truncated tile lists")
        redone = 0
        while True:
            k = self.regrow_exchange()
            if k == 0:
                return redone
            if k > len(pending):
                raise RuntimeError(f"{k} refused steps but only {len(pending)} remembered: call check_overflow() "
                                   "at least every CHECK_EVERY steps")
            for args in pending[-k:]:
                self._step_once(*args)
            redone += k

    def _optimizer_step(self, device_clock: bool, ticked: bool) -> None:
        if self.adam_fused:
            return
        x = self.exchange
        if x is not None and x.capacity and x.TAIL == "indexed" and device_clock:
            self.optim.step_gathered(self.slab.as_list(), x.recv, x.world, x.capacity, x.slot_table, pre_ticked=ticked)
        else:
            self.optim.step(self.slab.as_list(), device_clock=device_clock, pre_ticked=ticked)

    def _row_exchange_on(self) -> bool:
        """Agree on the segment size the first time (host-synchronous); False = dense all-reduce."""
        x = self.exchange
        if x is None:
            return False
        if x.capacity is None:
            if x.agree(int(self.rows.count.item()), self.slab.flat.numel()) == 0:
                # not worth it: from the next pass on everything is dense again (this pass's slab is
                # zero outside the listed rows, so its dense all-reduce is still right)
                self._go_dense()
                return False
        return True

    def _exchange_gradients(self, device_clock: bool = True) -> None:
        if self._row_exchange_on():
            if not getattr(self, "_packed", False):      # first step (segment size not agreed yet) or no views on this rank
                self.exchange.pack(self.rows)
            self.exchange.gather()
            x = self.exchange
            if not (x.TAIL == "indexed" and device_clock) and not torch.cuda.is_current_stream_capturing():
                # host-driven tail (ags_rows_unpack + row-set Adam): it cannot refuse a step on the device, so
                # the headers are read right here (every rank sees the same ones) and an outgrown segment is
                # repaired inside the step: own rows back into the slab, larger segment, pack and gather again
                while x.overflowed():
                    self.exchange_regrowths += 1
                    x.restore_own()
                    if x.agree(int(self.rows.count.item()), self.slab.flat.numel()) == 0:
                        # dense from here on; this step: the slab holds the rank's whole gradient again
                        x.union.reset()
                        self._go_dense()
                        all_reduce_(self.slab.flat, group=self.pg)
                        return
                    x.pack(self.rows)
                    x.gather()
            self._exchange_tail(device_clock)
        else:
            all_reduce_(self.slab.flat, group=self.pg)

    def _exchange_tail(self, device_clock: bool = True) -> None:
        """What follows the all-gather, up to (not including) the optimiser step."""
        if self.exchange.TAIL == "indexed" and device_clock:
            self.exchange.index()
        else:
            self.exchange.unpack()

    def _collectives_capturable(self) -> bool:
        """Can this process group's collectives be recorded into a hipGraph?  Only RCCL's can (they
        are kernels on the stream; gloo's run on the host).  Probed once with a 1-element all-reduce
        captured and replayed twice; any error or wrong sum means "no" and the step is captured as
        graph | collective | graph instead."""
        if getattr(self, "_capturable", None) is not None:
            return self._capturable
        ok = False
        if self.GRAPH_COLLECTIVES and torch.distributed.get_backend(self.pg) == "nccl":
            try:
                world = torch.distributed.get_world_size(self.pg)
                t = torch.ones(1, device=self.device)
                torch.distributed.all_reduce(t, group=self.pg)       # communicator exists before capture
                t.fill_(1.0)
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                g = torch.cuda.CUDAGraph()
                with torch.cuda.stream(side):
                    with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                        torch.distributed.all_reduce(t, group=self.pg)
                torch.cuda.current_stream().wait_stream(side)
                g.replay()
                g.replay()
                torch.cuda.synchronize()
                ok = abs(float(t.item()) - float(world) ** 2) < 0.5
            except Exception:
                ok = False
                torch.cuda.synchronize()
        # every rank must take the same branch
        flag = torch.tensor([1 if ok else 0], device=self.device, dtype=torch.int32)
        all_reduce_(flag, torch.distributed.ReduceOp.MIN, self.pg)
        self._capturable = bool(flag.item())
        return self._capturable

    def capture(self, cams: Sequence[api.Camera], image_grads: Callable, max_instances: int, repeat: int = 1,
                pipeline: bool = False) -> Callable:
        """Capture one optimisation step into hipGraphs and return a ``replay()`` callable.

        The library never allocates or synchronises and every per-view input that changes
        between iterations sits behind a device pointer (camera matrices, image gradients,
        parameters), so a step is a fixed launch sequence: new views are rendered by copying
        their matrices into the captured ``Camera`` tensors before ``replay()``.  With more
        than one rank the gradient exchange is recorded into the same graph when the transport's
        collectives are stream operations (RCCL); otherwise it stays outside (graph | collective |
        graph).  Call after at least one eager ``step`` so every buffer exists.  ``repeat`` > 1
        records that many consecutive optimisation steps in ONE graph, so a replay pays the
        graph-launch latency once per ``repeat`` steps; ``replay.steps`` says how many steps a call
        performs.  ``pipeline`` (single rank, one-pass binning): every recorded step's last launch also runs the
        per-Gaussian stage of the step that follows it (the same views again: ``cams[0]``), so a step is FOUR launches
        (tile sort, blend, blend backward, [Adam of this step + per-Gaussian stage of the next]); one eager pipelined
        step is taken here to prime the first replay, and ``cams[0]``'s matrices must not change between replays
        (or call ``step`` without ``next_cam`` once to leave the pipeline)."""
        self.check_overflow()              # settle the eager steps first: a capture bakes the segment size in
        self.optim.use_clock(True)
        dist_on = self._distributed()
        # (the pipelined kernel continues from the fused Adam update on raw parameters: _local_pass's own conditions)
        if pipeline and len(cams) > 1 and self.rows is not None and self.MULTI_VIEW_ROWS:
            import warnings
            warnings.warn("SurfelTrainer.capture(pipeline=True) with several cameras: a multi-view step joins its views in one "
                          "per-Gaussian launch (ags_backward_rows) and does not software-pipeline; recording the un-pipelined step")
            pipeline = False
        pipeline = (pipeline and not dist_on and self.binning_mode == api.BIN_DIRECT and len(cams) > 0
                    and self.rows is not None and self.fused_activations)
        if pipeline:
            self.step(cams, image_grads, max_instances, next_cam=cams[0])      # primes: cams[0] is prepared from here on
            self.check_overflow()
        rows_x = dist_on and self._row_exchange_on()
        in_graph = dist_on and self._collectives_capturable()
        repeat = max(1, int(repeat)) if (not dist_on or in_graph) else 1
        if dist_on and not in_graph and self._dense_chunked(cams):
            # host-driven collectives (gloo) between the row chunks: nothing to record, the step is replayed eagerly
            def replay_eager():
                self._dense_step(cams, image_grads, max_instances, True)
            replay_eager.collective_in_graph, replay_eager.pipelined, replay_eager.steps = False, False, 1
            return replay_eager
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        g_local, g_opt = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            if in_graph:
                # RCCL collectives are stream operations: the whole step, exchange included, is one graph
                try:
                    with torch.cuda.graph(g_local, stream=side, capture_error_mode="thread_local"):
                        for _ in range(repeat):
                            if self._dense_chunked(cams):       # chunked chain rule | all-reduce | Adam, all in the graph
                                self._dense_step(cams, image_grads, max_instances, True)
                                continue
                            ticked = self._local_pass(cams, image_grads, max_instances, tick=True)
                            self._exchange_gradients()
                            self._optimizer_step(True, ticked)
                except Exception:          # deterministic across ranks (same software): all fall back together
                    torch.cuda.synchronize()
                    self._capturable = in_graph = False
                    repeat = 1
                    g_local = torch.cuda.CUDAGraph()
            if dist_on and not in_graph:
                with torch.cuda.graph(g_local, stream=side, capture_error_mode="thread_local"):
                    ticked = self._local_pass(cams, image_grads, max_instances, tick=True)
                    if rows_x and not self._packed:
                        self.exchange.pack(self.rows)
                with torch.cuda.graph(g_opt, stream=side, capture_error_mode="thread_local"):
                    if rows_x:
                        self._exchange_tail()
                    self._optimizer_step(True, ticked)
            elif not dist_on:
                with torch.cuda.graph(g_local, stream=side, capture_error_mode="thread_local"):
                    for _ in range(repeat):
                        ticked = self._local_pass(cams, image_grads, max_instances, tick=True, fuse_adam=True,
                                                  next_cam=cams[0] if pipeline else None)
                        if not self.adam_fused:
                            self.optim.step(self.slab.as_list(), device_clock=True, pre_ticked=ticked)
        torch.cuda.current_stream().wait_stream(side)

        prepared_for = (self.state_for(cams[0].image_height, cams[0].image_width, max_instances), cams[0]) if pipeline else None

        def replay():
            if pipeline:
                # a pipelined graph starts at the tile sort of a pass whose per-Gaussian stage the previous step ran
                if self._prepared is None or self._prepared[0] is not prepared_for[0] or self._prepared[1] is not prepared_for[1]:
                    raise RuntimeError("pipelined replay: the prepared pass is gone (an un-pipelined step or reset_optimizer() "
                                       "ran in between) - capture again")
            else:
                self._drop_prepared()      # (no-op unless a pipelined step ran in between)
            g_local.replay()
            if dist_on and not in_graph:
                if rows_x:
                    self.exchange.gather()
                else:
                    all_reduce_(self.slab.flat, group=self.pg)
                g_opt.replay()

        replay.collective_in_graph = in_graph
        replay.pipelined = pipeline
        self._graphs = (g_local, g_opt)
        replay.steps = repeat
        # After some replays: ``trainer.regrow_exchange()`` (collective, host-synchronous) returns the number of
        # steps the exchange refused since the last look; if it is not 0, capture again (the new segment size is
        # part of the graph) and replay that many steps more.  ``trainer.workspace_overflows()`` likewise.
        return replay
