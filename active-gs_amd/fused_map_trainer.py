"""``GaussianMap.train()`` (/root/reference/mapping/gaussian_map.py:66-139) with every per-view
stage in the C ABI: activations inside the per-Gaussian kernels, forward, fused loss head
(stage 1 for all views -> visibility-count exchange -> stage 2 + backward per view, gradients
accumulated in place), one collective, fused Adam.  Same sampler, per-frame error tracking,
post-processing and prune as ``GaussianMapTrainer`` (which it extends); no torch autograd and
no per-view host synchronisation in the loop.
"""
from __future__ import annotations


from typing import List, Optional

import torch

from . import densify
from . import raster_api as api
from .camera import camera_matrices
from .dist_util import all_reduce_
from .fused_loss import FusedLoss
from .map_trainer import GaussianMapTrainer, make_frame_sampler
from .optimizer import FusedAdam
from .trainer import GradSlab


def weighted_choice_without_replacement(weights: torch.Tensor, k: int) -> torch.Tensor:
    """``k`` distinct indices drawn like ``np.random.choice(len(w), k, replace=False, p=w/sum(w))``
    (mapping/utils.py:190-228 draws its older frames that way) without leaving the device of
    ``weights``: successive sampling without replacement is the same as taking the k largest
    ``log(u_i) / w_i`` with u_i ~ U(0,1) (Efraimidis & Spirakis 2006).  No host synchronisation."""
    keys = torch.log(torch.rand_like(weights, dtype=torch.float32)) / weights.float().clamp(min=1e-30)
    return torch.topk(keys, k).indices


def weighted_choice_into(weights: torch.Tensor, k: int, out: torch.Tensor) -> None:
    """The same draw written into ``out`` (k int64 on the device) with two launches - torch.rand_like and
    ``ags_weighted_topk`` - instead of six; the same uniforms give the same indices as the function above."""
    n = weights.shape[0]
    if not weights.is_cuda or n > 8192 or weights.dtype != torch.float32 or not weights.is_contiguous() or not out.is_contiguous():
        out.copy_(weighted_choice_without_replacement(weights, k))
        return
    from . import _lib
    u = torch.rand_like(weights)
    _lib.check(_lib.load().ags_weighted_topk(_lib.ptr(u), _lib.ptr(weights), n, int(k), _lib.ptr(out),
                                             _lib.current_stream()), "ags_weighted_topk")


class FusedMapTrainer(GaussianMapTrainer):
    # the losses of the last train() call: kept on the device by the batched loop and read when somebody asks
    @property
    def last_losses(self):
        self.settle()
        if self._losses_dev is not None:
            self._losses_host, self._losses_dev = [float(x) for x in self._losses_dev.cpu()], None
        return self._losses_host

    @last_losses.setter
    def last_losses(self, value):
        self._losses_host, self._losses_dev = value, None

    _losses_host, _losses_dev = [], None

    def __init__(self, raw: dict, frames: List[dict], cfg: Optional[dict] = None, process_group=None,
                 binning_mode: int = api.BIN_DIRECT, use_graph: bool = True, num_streams: int = 4,
                 batched: bool = True):
        super().__init__(raw, frames, cfg, process_group=process_group)
        # single rank, frames of one size and field of view: all views of an iteration go through ONE
        # set of launches (ags_forward_batch / ags_backward_batch, loss stages with blockIdx.y = view)
        self.batched = batched
        self._batched_cache = None
        self.graph_min_steps = 50   # train_batched replays a hipGraph only for calls at least this long
        if self.device.type != "cuda":
            raise RuntimeError("FusedMapTrainer needs GPU tensors: there is no CPU fallback")
        self.binning_mode = binning_mode
        self.tuning = None       # _lib.AgsTuning handed over with every workspace this trainer makes (None: process default)
        # hipGraph replay of the iteration (single rank): the launch sequence of an iteration is
        # fixed for a given batch shape; what changes (which frames) is staged into static
        # per-slot camera / ground-truth buffers by four index_select launches before the replay.
        self.use_graph = use_graph
        # the views of an iteration are independent until the gradient sum: with num_streams > 1
        # they run concurrently on that many HIP streams (a 512x512 view alone leaves most SIMDs
        # idle), gradients and the visibility count are accumulated atomically, and the streams
        # join before the collective / Adam.
        self.num_streams = max(1, int(num_streams))
        self._streams = None
        self._cams = {}          # frame index -> (api.Camera, fov_x, fov_y)
        self._store = None       # _frame_store()
        self._uniform = None     # _uniform_frames()
        self._count_batch = None # _render_counts(): the chunk of views the prune pass renders at a time
        self._states = {}        # view slot -> ForwardState
        self._arena = None       # _map_arena(): the buffers the map's arrays are views of
        self._perf_buf = None    # _append_performance()
        self._snap_bufs = None   # _snapshot()
        self._state_bufs = {}    # _state(): view slot -> the buffers its ForwardState is a view of
        self._loss = None
        self._loss_bufs = []
        self._cap = 0
        self._last_need, self._last_need_n = None, 0   # most tile instances a view of the last train() call needed, and the map size then
        self._pending_check = None   # settle(): a train() call whose workspace check has not been looked at yet
        self.is_init = len(frames) > 0 and self.means.shape[0] > 0   # gaussian_map.py:35,130

    # ---- the keyframes' images and matrices as ONE growing set of arrays the sampled batch is gathered from on the
    # device.  Stacking all K frames at every train() call (as round 2 did) is K x 4 MB of copies and a fresh
    # allocation a size larger than the last per call: 63 ms per call at 300 keyframes against 9 ms at 50.
    def _frame_store(self):
        """(all_view (K,4,4), all_proj (K,4,4), all_rgb (K,3,H,W), all_depth (K,1,H,W)) of self.frames, appended to
        as frames arrive (capacity doubles); the frames must share one image size (_uniform_frames)."""
        K = len(self.frames)
        st = self._store
        h, w = self.frames[0]["rgb"].shape[-2:]
        ids = list(self.frames)                          # (the frame objects themselves: identity, not id() - ids are reused)
        if st is not None and (st["hw"] != (h, w) or st["n"] > K or any(a is not b for a, b in zip(st["ids"], ids))):
            st = None                                   # another image size, or the list was edited: rebuild
        if st is None or st["cap"] < K:
            cap = max(16, 2 * K)
            f = dict(device=self.device, dtype=torch.float32)
            new = dict(hw=(h, w), cap=cap, n=0, ids=[], view=torch.empty(cap, 4, 4, **f), proj=torch.empty(cap, 4, 4, **f),
                       rgb=torch.empty(cap, 3, h, w, **f), depth=torch.empty(cap, 1, h, w, **f))
            if st is not None:
                for k in ("view", "proj", "rgb", "depth"):
                    new[k][:st["n"]] = st[k][:st["n"]]
                new["n"], new["ids"] = st["n"], st["ids"]
            st = self._store = new
        for i in range(st["n"], K):
            cam = self._camera(i)[0]
            st["view"][i] = cam.viewmatrix; st["proj"][i] = cam.projmatrix
            st["rgb"][i] = self.frames[i]["rgb"]; st["depth"][i] = self.frames[i]["depth"].reshape(1, h, w)
        st["n"], st["ids"] = K, ids
        return st["view"][:K], st["proj"][:K], st["rgb"][:K], st["depth"][:K]

    # ---- small host -> device uploads (a keyframe's matrices) through page-locked memory: a copy from pageable memory
    # makes the host wait for everything the stream still has to do
    def _upload(self, array) -> torch.Tensor:
        import numpy as np
        a = np.ascontiguousarray(array, dtype=np.float32).reshape(-1)
        ring = getattr(self, "_ring", None)
        if ring is None:
            ring = self._ring = dict(buf=torch.empty(128, 64, dtype=torch.float32).pin_memory(), k=0, ev=[None] * 128)
        if a.size > 64:
            return torch.from_numpy(a).to(self.device)
        j = ring["k"] % 128
        slot = ring["buf"][j]
        ring["k"] += 1
        if ring["ev"][j] is not None:
            ring["ev"][j].synchronize()        # the copy that last read this slot (128 uploads ago) has executed: no wait in practice
        slot[:a.size] = torch.from_numpy(a)
        out = slot[:a.size].to(self.device, non_blocking=True)
        ev = ring["ev"][j] = ring["ev"][j] or torch.cuda.Event()
        ev.record()
        return out

    # ---- cached per-frame camera (the intrinsics -> fov step needs host scalars once per frame)
    def _camera(self, idx: int):
        c = self._cams.get(idx)
        if c is None:
            c = self._make_camera(self.frames[idx])
            self._cams[idx] = c
        return c

    def _make_camera(self, f: dict):
        """(api.Camera, fov_x, fov_y) of a frame.  The 4x4 / 3x3 algebra runs on the HOST (one small
        read of the pose, ONE small upload): on the GPU it is a dozen tiny launches and an inverse."""
        h, w = f["rgb"].shape[-2:]
        # camera.camera_matrices for one frame in float64 (rounded to float32 at the end) - the same algebra as ~40 small torch
        # ops (0.34 ms of every keyframe), and with scalar arithmetic where numpy's per-call overhead dominated (the four
        # unit rays and the 3x3 inverse: 75 -> 30 us; the GPU idles while this runs)
        import math

        import numpy as np
        dr = f.get("depth_range")
        hp = f.get("_pose_host")
        if hp is not None:
            # the HOST-POSE form: the caller still has the pose on the host (the simulator made it
            # there, mapper.py:94) - no read-back, so the render and the candidate kernels of this keyframe are enqueued
            # while the GPU is still running the previous call's iterations
            host = np.concatenate([np.asarray(x, dtype=np.float64).reshape(-1) for x in hp if x is not None])
            if hp[2] is not None:
                f["_far_host"] = float(host[25 + 1])
        else:
            # (pose, intrinsics and the far bound in ONE read-back when the frame lives on the device)
            parts = [f["extrinsic"].detach().reshape(-1), f["intrinsic"].detach().reshape(-1)]
            if torch.is_tensor(dr):
                parts.append(dr.detach().reshape(-1).to(parts[0].device))
            dt = torch.float64 if any(p_.dtype == torch.float64 for p_ in parts) else torch.float32   # (no conversion launches for float32 frames)
            host = torch.cat([p_.to(dt) for p_ in parts]).cpu().numpy().astype(np.float64)
            if torch.is_tensor(dr):
                f["_far_host"] = float(host[25 + 1])
        host32 = host.astype(np.float32)
        k = host[16:25].tolist()
        det = (k[0] * (k[4] * k[8] - k[5] * k[7]) - k[1] * (k[3] * k[8] - k[5] * k[6]) + k[2] * (k[3] * k[7] - k[4] * k[6]))
        Ki = [(k[4] * k[8] - k[5] * k[7]) / det, (k[2] * k[7] - k[1] * k[8]) / det, (k[1] * k[5] - k[2] * k[4]) / det,
              (k[5] * k[6] - k[3] * k[8]) / det, (k[0] * k[8] - k[2] * k[6]) / det, (k[2] * k[3] - k[0] * k[5]) / det,
              (k[3] * k[7] - k[4] * k[6]) / det, (k[1] * k[6] - k[0] * k[7]) / det, (k[0] * k[4] - k[1] * k[3]) / det]

        def ray(u, v):
            x, y, z = Ki[0] * u + Ki[1] * v + Ki[2], Ki[3] * u + Ki[4] * v + Ki[5], Ki[6] * u + Ki[7] * v + Ki[8]
            n = math.sqrt(x * x + y * y + z * z)
            return x / n, y / n, z / n

        def angle(a, b):
            return math.acos(min(1.0, max(-1.0, a[0] * b[0] + a[1] * b[1] + a[2] * b[2])))

        fov_x, fov_y = angle(ray(0.0, 0.5), ray(1.0, 0.5)), angle(ray(0.5, 0.0), ray(0.5, 1.0))
        near, far = (float(x) for x in self.cfg["bound"])
        tx, ty = math.tan(0.5 * fov_x), math.tan(0.5 * fov_y)
        P = np.zeros((4, 4))
        P[0, 0], P[1, 1], P[3, 2] = 1.0 / tx, 1.0 / ty, 1.0
        P[2, 2], P[2, 3] = far / (far - near), -(far * near) / (far - near)
        view = np.linalg.inv(host[:16].reshape(4, 4)).T
        proj = view @ P.T
        up = np.empty(41, dtype=np.float32)
        up[:16], up[16:32] = view.reshape(-1), proj.reshape(-1)
        need_inv = "intrinsic_inv" not in f
        if need_inv:     # what densify.candidates needs of the intrinsics (a float32 inverse like torch.linalg.inv's): no second read-back
            up[32:] = np.linalg.inv(host32[16:25].reshape(3, 3)).reshape(-1)
        dev_up = self._upload(up if need_inv else up[:32])
        if need_inv:
            f["intrinsic_inv"] = dev_up[32:41].reshape(3, 3)
        mats = dev_up[:32].reshape(2, 4, 4)
        tanx, tany = float(np.float32(tx)), float(np.float32(ty))
        cam = api.Camera(h, w, tanx, tany, mats[0], mats[1], self.background)
        return (cam, 2.0 * math.atan(tanx), 2.0 * math.atan(tany))

    def _state(self, slot: int, n: int, h: int, w: int) -> api.ForwardState:
        st = self._states.get(slot)
        if (st is None or st.max_instances < self._cap or st.radii.shape[0] != n or st.rgb.shape[-2:] != (h, w)
                or st.binning_mode != self.binning_mode):
            # the map changes size at every keyframe: the slot's buffers are allocated for twice the map and the state is a
            # view of their leading rows, laid out again (one memset) - not nine allocations per slot and keyframe
            full = self._state_bufs.get(slot)
            if (full is None or full.max_instances < self._cap or full.radii.shape[0] < n or full.rgb.shape[-2:] != (h, w)
                    or full.binning_mode != self.binning_mode or full.tuning is not self.tuning):
                full = api.alloc_state(max(2 * n, 1 << 16), h, w, self._cap, self.device, self.binning_mode, tuning=self.tuning)
                self._state_bufs[slot] = full
            st = api.ForwardState(full.rgb, full.normal, full.depth, full.opacity, full.confidence, full.importance[:n],
                                  full.count[:n], full.radii[:n], full.workspace, full.max_instances, full.binning_mode,
                                  full.tuning)
            api.init_workspace(st, n, h, w)
            self._states[slot] = st
        return st

    def _gaussians(self) -> api.Gaussians:
        n = self.means.shape[0]
        return api.Gaussians(self.means, self.scales, self.rotations, self.opacities, self.harmonics.view(n, 3),
                             self._confidences().contiguous(), raw_params=True, scale_factor=self.cfg["scale_factor"],
                             max_scale=0.05)

    # One-pass binning needs tiles x the LONGEST tile list of key slots; with badly skewed lists (a distant or
    # zoomed-out camera: most surfels in a few tiles) that is far above the instance total, which is all the scan-based
    # binning needs - same images.  Past these thresholds the trainer goes on with AGS_BIN_TILE_SORT.
    SKEW_FACTOR, DIRECT_BUDGET_BYTES = 8, 1 << 30
    COUNT_CHUNK = 16           # views per batched launch of the prune pass's count render (_render_counts)
    MULTI_VIEW_ROWS = True     # the batched iteration's per-Gaussian backward + Adam as ONE launch (ags_backward_rows)
    _no_grads = api.GaussianGrads(None, None, None, None, None)

    def _grow_cap(self, need: int, instances: int = 0) -> None:
        """A view needed ``need`` key slots (``AgsStatus.needed_instances``; ``instances`` = its instance total when
        known): raise the capacity new workspaces are made with - or leave one-pass binning when its need is skew."""
        if (self.binning_mode == api.BIN_DIRECT and instances > 0 and need > self.SKEW_FACTOR * max(instances, 1 << 16)
                and need * 24 > self.DIRECT_BUDGET_BYTES):
            self.binning_mode = api.BIN_TILE_SORT
            self.mode_switches = getattr(self, "mode_switches", 0) + 1
            self._states.clear()
            if getattr(self, "_batched_cache", None):
                self._batched_cache["batch"] = None
            need = instances
            self._cap = 0
        self._cap = min(max(int(need * 1.5) + 4096, self._cap + 1), 0xFFFFFFFF)

    def _check_capacity(self, slots) -> bool:
        infos = [api.read_status(self._states[s]) for s in slots]
        need = max((i["needed"] for i in infos), default=0)
        if need > self._cap:
            self._grow_cap(need, max((i["num_instances"] for i in infos), default=0))
            return False
        return True

    def _check_capacity_sticky(self, slots) -> bool:
        """After a loop: did ANY pass on these workspaces since they were initialised need more tile instances
        than they hold (``AgsStatus.peak_instances / overflow_passes``)?  Raises ``_cap`` if so.  With several
        ranks the answer is agreed (MAX) so that all of them repeat the call together."""
        peak, bad, inst = 0, 0, 0
        for s in slots:
            info = api.read_status(self._states[s])
            peak, bad, inst = max(peak, info["peak_instances"]), bad + info["overflow_passes"], max(inst, info["num_instances"])
        if self.world > 1:
            t = torch.tensor([peak, bad, inst], device=self.device, dtype=torch.int64)
            all_reduce_(t, torch.distributed.ReduceOp.MAX, self.pg)
            peak, bad, inst = int(t[0]), int(t[1]), int(t[2])
        if bad:
            self._grow_cap(peak, inst)      # (inst: the LAST pass's total - a proxy for the pass that peaked)
            return False
        return True

    # ---- a train() call is all-or-nothing: a view that outgrows its workspace in ANY iteration truncates its
    # tile lists (wrong gradients, no crash), so the loops below only note it (sticky status words, read once
    # after the loop) and the call is then repeated from a snapshot with larger workspaces
    def _snapshot(self) -> dict:
        import numpy as np
        self.settle()      # a pending check's snapshot lives in the buffers overwritten below: look at it first
        # into buffers kept across calls, as ONE multi-tensor copy (six clones are six allocations and six launches with the
        # GPU idle behind them)
        # (the view statistics too: a call whose check is deferred has its post-processing enqueued before anybody knows
        # whether the call stands - settle())
        keys = ("means", "scales", "rotations", "opacities", "harmonics", "training_performance", "view_supports", "view_means",
                "view_scores")
        for k in keys[6:]:
            setattr(self, k, getattr(self, k).float().contiguous())
        src = [getattr(self, k) for k in keys]
        bufs = self._snap_bufs
        if (bufs is None or len(bufs) != len(src) or any(b.shape[0] < t.shape[0] or b.shape[1:] != t.shape[1:] or b.dtype != t.dtype or b.device != t.device
                                for b, t in zip(bufs, src))):
            bufs = self._snap_bufs = [torch.empty((max(2 * t.shape[0], 64),) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
                                      for t in src]
        dst = [b[:t.shape[0]] for b, t in zip(bufs, src)]
        torch._foreach_copy_(dst, src)
        snap = dict(zip(keys, dst))
        # the random streams a repeated call has to draw from again: numpy's global one (the reference's weighted sampler)
        # only when the host sampler is in use - reading it costs 0.8 ms inside this loop, as much as the rest of the
        # call's set-up - torch's CPU generator (sampler_type "uniform": randperm), the device generator (cfg["sampler"]
        # = "device")
        snap["np_rng"] = None if self._device_sampler() else np.random.get_state()
        snap["torch_rng"] = torch.get_rng_state()
        snap["cuda_rng"] = torch.cuda.get_rng_state(self.device)
        return snap

    def _device_sampler(self) -> bool:
        return self.cfg.get("sampler", "host") == "device" and self.cfg.get("sampler_type", "weighted") == "weighted"

    def _restore(self, snap: dict) -> None:
        import numpy as np
        for k in ("means", "scales", "rotations", "opacities", "harmonics", "training_performance", "view_supports", "view_means",
                  "view_scores"):
            getattr(self, k).copy_(snap[k])
        if snap["np_rng"] is not None:
            np.random.set_state(snap["np_rng"])
        torch.set_rng_state(snap["torch_rng"])
        torch.cuda.set_rng_state(snap["cuda_rng"], self.device)

    def _all_or_nothing(self, run, steps) -> None:
        self.settle()
        snap = self._snapshot()
        for _ in range(6):
            if run(steps):
                self.post_processing()
                return
            self.overflow_retries = getattr(self, "overflow_retries", 0) + 1
            self._restore(snap)
        raise RuntimeError("train(): the rasterizer workspace kept overflowing after six enlargements")

    def train(self, steps: Optional[int] = None):
        self.settle()
        if self.batched and self._uniform_frames():
            return self.train_batched(steps)
        self._all_or_nothing(self._train_views, steps)

    def _train_views(self, steps: Optional[int] = None) -> bool:
        dist = torch.distributed
        lrs = self.cfg["lrs"]
        for name in ("means", "scales", "rotations", "opacities", "harmonics"):
            setattr(self, name, getattr(self, name).contiguous())
        params = [self.means, self.scales, self.rotations, self.opacities, self.harmonics]
        n = self.means.shape[0]
        optim = FusedAdam(params, [lrs["mean"], lrs["scale"], lrs["rotation"], lrs["opacity"], lrs["harmonic"]], eps=1e-15)
        slab = GradSlab(n, self.device)
        # sticky set of the surfels this call's views have shown (single rank; see api.RowSet)
        rows = api.RowSet(n, self.device) if self.world == 1 else None
        optim.touched = rows
        slab.flat.zero_()
        sampler = make_frame_sampler(self.cfg, self.frames)
        self.last_losses = []
        self._cap = max(self._cap, 1 << 16, 2 * n)
        sizes = {tuple(f["rgb"].shape[-2:]) for f in self.frames}
        if len(sizes) > 1:
            # the reference stacks the sampled frames (torch.stack, /root/reference/mapping/utils.py:220-221,253-254): frames of
            # different sizes cannot be trained together there either
            raise ValueError(f"keyframes of different image sizes cannot be trained together: {sorted(sizes)}")
        for it in range(self.cfg["optimization_steps"] if steps is None else steps):
            ids = sampler.next_ids(self.training_performance)
            B = len(ids)
            mine = list(range(self.rank, B, self.world))
            h, w = self.frames[int(ids[0])]["rgb"].shape[-2:]
            if self._loss is None or (self._loss.h, self._loss.w) != (h, w):
                _, fx, fy = self._camera(int(ids[0]))
                self._loss = FusedLoss(h, w, fx, fy, B, self.cfg["batch_size"], self.device)
                self._loss_bufs = []
            self._loss.set_batch_total(B)
            while len(self._loss_bufs) < len(mine):
                self._loss_bufs.append(self._loss.alloc_view())
            g = self._gaussians()
            S = min(self.num_streams, max(len(mine), 1))
            if S > 1 and self._streams is None:
                self._streams = [torch.cuda.Stream() for _ in range(self.num_streams)]
            main = torch.cuda.current_stream()

            main_h = main.cuda_stream
            side_h = [s_.cuda_stream for s_ in self._streams] if S > 1 else []

            def fan_out(fn):
                """Run fn(slot, b, stream_handle) for this rank's views, round-robin over S streams that
                first wait for the main stream and that the main stream then waits for.  The kernels
                are enqueued through the C ABI with the raw stream handle (no torch stream switch)."""
                if S == 1:
                    for slot, b in enumerate(mine):
                        fn(slot, b, main_h)
                    return
                for k in range(S):
                    self._streams[k].wait_stream(main)
                for slot, b in enumerate(mine):
                    fn(slot, b, side_h[slot % S])
                for k in range(S):
                    main.wait_stream(self._streams[k])

            def fwd_view(slot, b, sh):
                cam, _, _ = self._camera(int(ids[b]))
                st = self._state(slot, n, h, w)
                api.forward(cam, g, st, stream=sh, checked=True, touched=rows)
                f = self.frames[int(ids[b])]
                self._loss.stage1(st, f["rgb"], f["depth"], self._loss_bufs[slot], b, -1 if S > 1 else slot == 0, sh)

            def bwd_view(slot, b, sh):
                cam, fxv, fyv = self._camera(int(ids[b]))
                st, buf = self._states[slot], self._loss_bufs[slot]
                self._loss.cfg.fov_x, self._loss.cfg.fov_y = float(fxv), float(fyv)   # depth -> normal with THIS view's intrinsics
                self._loss.stage2(st, self.frames[int(ids[b])]["depth"], buf, sh)
                api.backward(cam, g, st, buf.d_rgb, buf.d_normal, buf.d_depth, None, None, grads=slab.grads,
                             accumulate=2 if S > 1 else (slot > 0), stream=sh, touched=rows)

            while True:  # forward every local view; re-run the batch once if a workspace was too small
                self._loss.begin_step()
                if S > 1:
                    self._loss.msum.zero_()
                    for slot in range(len(mine)):   # allocate outside the side streams
                        self._state(slot, n, h, w)
                fan_out(fwd_view)
                # first iteration: cheap early sizing (nothing has been stepped yet).  Later iterations are
                # covered by the sticky status words read once after the loop.  (world > 1: no per-rank
                # re-run here, the ranks' collectives must stay aligned.)
                if it > 0 or self.world > 1 or self._check_capacity(range(len(mine))):
                    break
                for slot in range(len(mine)):       # re-allocated (fresh sticky words) by _state() above on the re-run
                    self._states.pop(slot, None)
            if not mine:
                self._loss.msum.zero_()
            if self.world > 1:
                all_reduce_(self._loss.msum, group=self.pg)
            if S > 1 or not mine:
                slab.flat.zero_()
            fan_out(bwd_view)
            if self.world > 1:
                all_reduce_(slab.flat, group=self.pg)
                all_reduce_(self._loss.accum, group=self.pg)  # 64 x (4+2B) floats
            self.training_performance[torch.as_tensor(ids, device=self.device)] = self._loss.per_frame_errors(B)
            optim.step(slab.as_list())
            self.last_losses.append(self._loss.total_loss())
        used = [s for s in self._states if isinstance(s, int)]
        if not self._check_capacity_sticky(used):
            for s in used:
                self._states.pop(s)
            return False
        self.last_losses = [float(x) for x in self.last_losses]
        return True

    # ---- post_processing (gaussian_map.py:141-232) and get_confidences (:552-565) without their ~30 torch ops
    def confidences(self):
        """(public readers see a settled map; the loop's own launches use ``_confidences`` - the view statistics do not
        change inside a train() call, and add_gaussians settles at its own wait)"""
        self.settle()
        return self._confidences()

    def _confidences(self):
        n = self.means.shape[0]
        if n == 0 or not self.means.is_cuda:
            return super().confidences()
        from . import _lib
        from ._lib import ptr
        for k in ("view_supports", "view_means", "view_scores"):
            setattr(self, k, getattr(self, k).float().contiguous())
        out = torch.empty(n, device=self.device, dtype=torch.float32)
        _lib.check(_lib.load().ags_confidences(n, ptr(self.view_supports), ptr(self.view_means), ptr(self.view_scores),
                                               int(bool(self.cfg["use_view_distribution"])), ptr(out),
                                               _lib.current_stream()), "ags_confidences")
        return out

    def post_processing(self):
        self.settle()
        return self._post_processing()

    def _post_processing(self):
        """The reference's rule (count render of the newest keyframe, or of ALL keyframes every prune_interval-th frame;
        supports / view means / view scores of the surfels the newest frame sees; prune what no keyframe sees) with the
        per-surfel bookkeeping as one launch (``ags_view_stats_update``) and the ground-truth depths taken from the
        frame store instead of a stack of all keyframes."""
        if self.world > 1 or not self.means.is_cuda or self.means.shape[0] == 0 or not self._uniform_frames():
            return super().post_processing()
        self._post_processing_end(self._post_processing_begin())

    def _post_processing_begin(self):
        """Enqueue the count render of the newest keyframe WITHOUT waiting for its status (``_post_processing_end`` looks at
        it); the prune pass over all keyframes (every prune_interval-th frame) is left to ``_post_processing_end`` whole."""
        if self.world > 1 or not self.means.is_cuda or self.means.shape[0] == 0 or not self._uniform_frames():
            return None
        k = len(self.frames)
        if k % self.cfg["prune_interval"] == 0:
            return None
        depth_gt = self._frame_store()[3][k - 1]
        h, w = depth_gt.shape[-2:]
        n = self.means.shape[0]
        self._cap = max(self._cap, 1 << 16, 2 * n)
        g = self._gaussians()
        cam0, _, _ = self._camera(k - 1)
        cam = api.Camera(h, w, cam0.tanfovx, cam0.tanfovy, cam0.viewmatrix, cam0.projmatrix, self.background,
                         want_stats=api.STATS_SEEN, front_only=True, render_mask=(depth_gt > 0.0).float().contiguous())
        st = self._state("count", n, h, w)
        api.forward(cam, g, st)
        return dict(cam=cam, g=g, n=n, hw=(h, w))

    def _post_processing_end(self, pending, check: bool = True):
        """``check`` False (a call whose workspace check is deferred, settle()): the count render's status is not read here."""
        if self.world > 1 or not self.means.is_cuda or self.means.shape[0] == 0 or not self._uniform_frames():
            return super().post_processing()
        from . import _lib
        from ._lib import ptr
        k = len(self.frames)
        prune_now = k % self.cfg["prune_interval"] == 0
        n = self.means.shape[0]
        if pending is not None and not prune_now and pending["n"] == n:
            while check and not self._check_capacity(["count"]):       # (the GPU has drained by now: this read does not wait)
                st = self._state("count", n, *pending["hw"])
                api.forward(pending["cam"], pending["g"], st)
            newest = self._states["count"].count
            counts = None
        else:
            use = list(range(k)) if prune_now else [k - 1]
            depth_all = self._frame_store()[3]
            depth_gt = depth_all if prune_now else depth_all[k - 1:k]
            h, w = depth_gt.shape[-2:]
            params = [self.means, self.scales, self.rotations, self.opacities, self.harmonics]
            counts = self._render_counts(use, None, None, depth_gt, params, (h, w))
            newest = counts[-1].contiguous()
        last = self.frames[-1]
        far = last.get("_far_host")           # (read back with the frame's pose: _make_camera)
        if far is None:
            far = last["depth_range"][1]
            far = float(far.item()) if torch.is_tensor(far) else float(far)
        campos = last["extrinsic"][:3, 3].float().contiguous()
        for key in ("view_supports", "view_means", "view_scores", "means", "rotations"):
            setattr(self, key, getattr(self, key).float().contiguous())
        _lib.check(_lib.load().ags_view_stats_update(n, ptr(self.means), ptr(self.rotations), ptr(campos), far, ptr(newest),
                                                     int(bool(self.cfg["use_view_distribution"])), ptr(self.view_supports),
                                                     ptr(self.view_means), ptr(self.view_scores),
                                                     _lib.current_stream()), "ags_view_stats_update")
        if prune_now:
            self.prune(~(counts.sum(0) >= 1))

    def _render_counts(self, frame_ids, extr, intr, depth_gt, params, hw):
        """Count render of post_processing straight through the C ABI (forward only, importance /
        count enabled, front_only, render mask = valid ground-truth depth): no module, no autograd."""
        n = self.means.shape[0]
        h, w = hw
        self._cap = max(self._cap, 1 << 16, 2 * n)
        g = self._gaussians()
        if len(frame_ids) > 1 and self._uniform_frames():
            # the prune pass over all keyframes: batched launches over COUNT_CHUNK views at a time, one status read per
            # chunk, in ONE set of buffers kept across calls (a batch of all K keyframes is K workspaces: 30 GB of
            # fresh allocations per pass at 300 keyframes)
            cam0, _, _ = self._camera(int(frame_ids[0]))
            V, CH = len(frame_ids), self.COUNT_CHUNK
            vm = torch.stack([self._camera(int(f))[0].viewmatrix for f in frame_ids])
            pm = torch.stack([self._camera(int(f))[0].projmatrix for f in frame_ids])
            masks = (depth_gt > 0.0).float().reshape(V, h, w)
            out = torch.empty(V, n, device=self.device, dtype=torch.int32)
            v0 = 0
            while v0 < V:
                cnt = min(CH, V - v0)
                batch = self._count_batch
                key = (h, w, round(cam0.tanfovx, 7), round(cam0.tanfovy, 7), self.binning_mode)
                if (batch is None or batch._count_key != key or batch.capacity_n < n or batch.max_instances < self._cap):
                    self._count_batch = None            # release before the larger one is made
                    batch = api.ViewBatch(g, CH, h, w, cam0.tanfovx, cam0.tanfovy, self.background, self._cap,
                                          want_stats=api.STATS_SEEN, front_only=True, render_masks=torch.zeros(CH, h, w, device=self.device),
                                          binning_mode=self.binning_mode, capacity_n=max(2 * n, 1 << 16), tuning=self.tuning)
                    batch._count_key = key
                    self._count_batch = batch
                elif batch.g is not g:
                    batch.bind(g)
                batch.masks[:cnt] = masks[v0:v0 + cnt]
                batch.viewmats[:cnt] = vm[v0:v0 + cnt]
                batch.projmats[:cnt] = pm[v0:v0 + cnt]
                batch.forward(cnt)
                stw = batch.statuses(cnt)
                need = int(stw[:, 7].max())
                if need > self._cap:
                    self._grow_cap(need, int(stw[:, 0].max()))
                    self._count_batch = None            # rebuilt with the larger capacity (or the other binning mode)
                    continue
                out[v0:v0 + cnt] = batch.count[:cnt]
                v0 += cnt
            return out
        out = []
        for k, fid in enumerate(frame_ids):
            cam0, _, _ = self._camera(int(fid))
            cam = api.Camera(h, w, cam0.tanfovx, cam0.tanfovy, cam0.viewmatrix, cam0.projmatrix, self.background,
                             want_stats=api.STATS_SEEN, front_only=True, render_mask=(depth_gt[k] > 0.0).float().contiguous())
            while True:
                st = self._state("count", n, h, w)
                api.forward(cam, g, st)
                if self._check_capacity(["count"]):
                    break
            out.append(st.count.clone())
        return torch.stack(out)

    # ------------------------------------------------------------------ map growth / pruning
    def _map_state(self) -> dict:
        return {k: getattr(self, k) for k in densify.STATE_KEYS}

    def _set_map_state(self, state: dict) -> None:
        for k in densify.STATE_KEYS:
            setattr(self, k, state[k])
        self._states.clear()          # per-view workspaces are sized by the number of surfels

    # Where a keyframe's time goes, measured in ONE run: a caller may set ``phase_hook`` (callable(label)); the loop calls it at
    # its phase boundaries - "grow" (add_gaussians), "set-up" (a train() call up to its first iteration), "iterations",
    # "post" (count render, the wait of the call, view statistics / prune), "between".  synthetic.run_mapper_loop(phases=True)
    # records a HIP event and the host clock there.
    phase_hook = None

    def _phase(self, label: str) -> None:
        if self.phase_hook is not None:
            self.phase_hook(label)

    def add_gaussians(self, frame: dict) -> int:
        """``GaussianMap.add_gaussians`` (gaussian_map.py:294-468): spawn surfels from a new RGB-D
        keyframe where the map's own render is wrong / empty / occluding, one per 2 cm voxel, then
        register the frame.  All per-pixel work and the compaction run in densify.hip.  Returns
        the number of surfels added."""
        self._phase("grow")
        try:
            return self._add_gaussians(frame)
        finally:
            self._phase("between")

    @staticmethod
    def _host_pose(frame: dict):
        """(extrinsic, intrinsic, depth_range) as host arrays when the caller still has them there: CPU tensors / numpy arrays
        under the usual keys, or under ``extrinsic_host`` / ``intrinsic_host`` / ``depth_range_host`` next to device copies."""
        import numpy as np
        out = []
        for k in ("extrinsic", "intrinsic", "depth_range"):
            v = frame.get(k + "_host", frame.get(k))
            if torch.is_tensor(v):
                if v.is_cuda:
                    return None if k != "depth_range" else tuple(out) + (None,)
                v = v.detach().numpy()
            out.append(None if v is None else np.asarray(v))
        return tuple(out) if out[0] is not None and out[1] is not None else None

    def _add_gaussians(self, frame: dict) -> int:
        pose_host = self._host_pose(frame)
        frame = {k: (v.to(self.device) if torch.is_tensor(v) else v) for k, v in frame.items() if not k.endswith("_host")}
        if pose_host is not None:
            frame["_pose_host"] = pose_host
        if self.frames and tuple(frame["rgb"].shape[-2:]) != tuple(self.frames[0]["rgb"].shape[-2:]):
            # (the reference stacks the sampled frames, /root/reference/mapping/utils.py:220-221,253-254: it cannot train such a set either)
            raise ValueError(f"keyframes of different image sizes cannot be trained together: {tuple(frame['rgb'].shape[-2:])} "
                             f"after {tuple(self.frames[0]['rgb'].shape[-2:])}")
        pred = None
        if self.is_init and self.means.shape[0] > 0:
            h, w = frame["rgb"].shape[-2:]
            n = self.means.shape[0]
            made = self._make_camera(frame)
            self._cams[len(self.frames)] = made          # the frame is registered below under this index
            cam = made[0]
            self._cap = max(self._cap, 1 << 16, 2 * n)
            g = self._gaussians()
            while True:
                # the render's capacity check rides on the read-back densify needs anyway (its row count): one wait for
                # the GPU per keyframe here instead of two; an outgrown workspace (rare) repeats both.  The previous train()
                # call's pending check (settle()) is looked at at that same wait: everything up to here - the render, the
                # smoothing, the candidates, the voxel filter - was enqueued behind that call's iterations without waiting.
                st = self._state("densify", n, h, w)
                api.forward(cam, g, st)
                pred = dict(rgb=st.rgb, depth=st.depth[0], opacity=st.opacity[0])
                grown = densify.add_gaussians(self._map_state(), frame, pred, self.cfg["error_thres"], arena=self._map_arena(),
                                              before_sync=self.settle)
                if grown is None:            # the pending call was repeated: the map is not what this render showed
                    g = self._gaussians()
                    continue
                state, added = grown
                if self._check_capacity(["densify"]):
                    break
        else:
            self.settle()
            state, added = densify.add_gaussians(self._map_state(), frame, pred, self.cfg["error_thres"], arena=self._map_arena())
        self._set_map_state(state)
        self.frames.append(frame)
        self._append_performance(10.0)
        return added

    def _map_arena(self) -> "densify.MapArena":
        """The buffers the map's arrays are the leading rows of (densify.MapArena): made on first use, with room for a map
        twice the current size."""
        if self._arena is None:
            self._arena = densify.MapArena(max(2 * self.means.shape[0], 1 << 18), self.device)
        return self._arena

    def _append_performance(self, value: float) -> None:
        """training_performance = cat(training_performance, [value]) (gaussian_map.py:466-468) as one fill of the next
        element of a buffer with room (the cat is an allocation, a copy and an upload from pageable memory per keyframe)."""
        perf = self.training_performance
        k = perf.shape[0]
        buf = self._perf_buf
        if buf is None or buf.data_ptr() != perf.data_ptr() or k + 1 > buf.shape[0] or perf.dtype != torch.float32:
            buf = torch.empty(max(64, 2 * (k + 1)), device=self.device, dtype=torch.float32)
            buf[:k].copy_(perf)
            self._perf_buf = buf
        buf[k:k + 1].fill_(float(value))
        self.training_performance = buf[:k + 1]

    def update(self, frame: dict, steps: Optional[int] = None) -> None:
        """``GaussianMap.update`` (gaussian_map.py:62-64): grow the map from the keyframe, then train."""
        self.add_gaussians(frame)
        self.train(steps)
        self.is_init = True

    def prune(self, mask):
        self.settle()
        state, deleted = densify.prune(self._map_state(), mask, arena=self._map_arena())
        self._set_map_state(state)
        return deleted

    # ------------------------------------------------------------------ batched iteration
    def _uniform_frames(self) -> bool:
        """One image size and one field of view for all keyframes (one AgsFrame for the whole batch)?  Looked at frame by
        frame as the list grows (the answer for the first K frames is kept; a list that was edited is looked at again)."""
        if self.world > 1 or len(self.frames) == 0 or self.means.shape[0] == 0:
            return False
        K = len(self.frames)
        c = self._uniform
        if c is None or c["n"] > K or c["first"] is not self.frames[0] or (c["n"] and c["last"] is not self.frames[c["n"] - 1]):
            c = self._uniform = dict(n=0, key=None, ok=True, first=self.frames[0], last=None)
        for i in range(c["n"], K):
            cam = self._camera(i)[0]
            key = (tuple(self.frames[i]["rgb"].shape), round(cam.tanfovx, 7), round(cam.tanfovy, 7))
            if c["key"] is None:
                c["key"] = key
            c["ok"] = c["ok"] and key == c["key"]
        c["n"], c["last"] = K, self.frames[K - 1]
        return c["ok"]

    def train_batched(self, steps: Optional[int] = None):
        """All-or-nothing like every train() here, with ONE wait at its end: the loop's iterations and the count render of
        post_processing are enqueued back to back, then the sticky status words of the loop's views and the count
        render's status are read together (round 3 waited for the loop, then again for the count render)."""
        self.settle()
        self._phase("set-up")
        snap = self._snapshot()
        self._train_batched_checked(steps, snap, 6, defer_ok=True)

    # The wait of a train() call - did any pass of any iteration, or the count render, outgrow its workspace? - keeps the
    # GPU idle for as long as the host then needs to enqueue what follows it: the view statistics, and the next keyframe's
    # render / candidates up to ITS read-back.  On keyframes that do not prune the call therefore returns with the check
    # PENDING: the status words are copied to page-locked memory behind the count render, the view-statistics update is
    # enqueued at once, and ``settle()`` looks at the words at the next point that waits for the GPU anyway (the row count
    # of the next add_gaussians) - or when anybody reads the map (GaussianMap's attributes, last_losses, save).  A call
    # that did overflow (rare: workspaces carry head-room) is then repeated from its snapshot - parameters, per-frame
    # errors, view statistics, random streams - exactly as the immediate check would have repeated it.
    DEFER_SETTLE = True
    # batched iterations: loss stage 1 as the epilogue of the forward blend kernel (ags_forward_batch_loss).  Built, bit-identical,
    # MEASURED SLOWER on the mapper loop (0.3300 s against 0.3266 s for 500 iterations, profiles/r06_mapper_ab.md: the
    # stand-alone launch streams four images at 29 us per eleven views, the epilogue adds four atomics per wave and the
    # ground-truth reads to a kernel bound by vector-instruction issue): off by default
    FUSE_LOSS_STAGE1 = False

    def _train_batched_checked(self, steps, snap, attempts: int, defer_ok: bool = False) -> None:
        for attempt in range(attempts):
            settle = self._train_batched(steps, defer=True)
            self._phase("post")
            pending = self._post_processing_begin()
            if defer_ok and attempt == 0 and self.DEFER_SETTLE and pending is not None and settle.batch is not None:
                words_b = self._words_async(settle.batch.status_words())
                count_state = self._states["count"]
                words_c = self._words_async(count_state.workspace[:32].view(torch.int32).view(1, 8))
                self._post_processing_end(pending, check=False)          # the view statistics, enqueued unchecked
                ev = torch.cuda.Event()
                ev.record()
                self._pending_check = dict(event=ev, words_b=words_b, words_c=words_c, count_cap=count_state.max_instances,
                                           snap=snap, steps=steps, settle=settle)
                self._phase("between")
                return
            if settle():
                self._post_processing_end(pending)
                self._phase("between")
                return
            self.overflow_retries = getattr(self, "overflow_retries", 0) + 1
            self._restore(snap)
        raise RuntimeError("train(): the rasterizer workspace kept overflowing after six enlargements")

    def _words_async(self, words_dev: torch.Tensor) -> torch.Tensor:
        """(V, 8) int32 status words -> page-locked host tensor, copy enqueued on the current stream (no wait).  The host
        buffers are kept (page-locking memory costs more than the wait it saves): two per shape, used in turn - a pending
        check is always settled before the next train() call makes another."""
        key = tuple(words_dev.shape)
        pool = self.__dict__.setdefault("_words_pool", {})
        slot = pool.setdefault(key, [[torch.empty(key, dtype=torch.int32).pin_memory() for _ in range(2)], 0])
        host = slot[0][slot[1] & 1]
        slot[1] += 1
        host.copy_(words_dev, non_blocking=True)
        return host

    def settle(self) -> bool:
        """Look at the workspace check a train() call left pending (see DEFER_SETTLE); repeats the call if it overflowed.
        Returns True when a call was repeated (the map's parameters are then not what they were a moment ago).  Cheap when
        nothing is pending; waits for the GPU otherwise.
        A repeated call starts from its snapshot INCLUDING the random streams (numpy's global one when the host sampler is in
        use, torch's CPU and device generators): they are rewound to where the call began, so draws a caller made from those
        GLOBAL streams between train() returning and this look (it may run inside a later attribute read) are drawn again.
        The reference seeds nothing (main.py, mapper.py), so no caller can depend on those streams; one that does should call
        ``settle()`` right after ``train()`` / ``update()`` or set ``DEFER_SETTLE = False``."""
        p = self._pending_check
        if p is None:
            return False
        self._pending_check = None
        p["event"].synchronize()
        ok = p["settle"].evaluate((p["words_b"].to(torch.int64)) & 0xFFFFFFFF)
        wc = (p["words_c"].to(torch.int64)) & 0xFFFFFFFF
        if int(wc[0, 7]) > p["count_cap"]:               # the count render's lists were truncated
            self._grow_cap(int(wc[0, 7]), int(wc[0, 0]))
            ok = False
        if ok:
            return False
        self.overflow_retries = getattr(self, "overflow_retries", 0) + 1
        self._restore(p["snap"])
        self._train_batched_checked(p["steps"], p["snap"], 5, defer_ok=False)
        return True

    def _train_batched(self, steps: Optional[int] = None, defer: bool = False):
        """``train`` with the B views of an iteration in ONE set of launches per ITERATION instead of per view (a 512x512
        view is 1024 tiles - a quarter of what the GPU holds): per-Gaussian stage, tile sort, blend, two loss stages, blend
        backward, ONE per-Gaussian backward + Adam over the rows the views showed, and a last launch that writes the frame
        errors, draws the next iteration's frames and stages their matrices (eight launches).  The sampled frames' images
        are read where they are, in the keyframe store."""
        lrs = self.cfg["lrs"]
        for name in ("means", "scales", "rotations", "opacities", "harmonics"):
            setattr(self, name, getattr(self, name).contiguous())
        params = [self.means, self.scales, self.rotations, self.opacities, self.harmonics]
        n, dev = self.means.shape[0], self.device
        # (the call's buffers are zeroed together further down: one launch instead of seven)
        optim = FusedAdam(params, [lrs["mean"], lrs["scale"], lrs["rotation"], lrs["opacity"], lrs["harmonic"]], eps=1e-15,
                          zero=False)
        slab = GradSlab(n, dev, zero=False)
        rows = api.RowSet(n, dev, zero=False)
        optim.touched = rows
        sampler = make_frame_sampler(self.cfg, self.frames)
        K = len(self.frames)
        h, w = self.frames[0]["rgb"].shape[-2:]
        cam0, fx, fy = self._camera(0)
        all_view, all_proj, all_rgb, all_depth = self._frame_store()
        Bmax = self.cfg["batch_size"] + self.cfg["active_size"]
        self._cap = max(self._cap, 1 << 16, 2 * n)
        g = self._gaussians()
        # buffers that do not depend on the map size live across train() calls; the ViewBatch is
        # allocated with head-room for a growing map and re-bound (bind) while the map fits
        key = (h, w, Bmax, round(fx, 7), round(fy, 7))
        keep = self._batched_cache if self._batched_cache and self._batched_cache["key"] == key else None
        if keep is None:
            loss = FusedLoss(h, w, fx, fy, Bmax, Bmax, dev)
            # (gathered copies of the sampled frames' images: only where the loss stages cannot read them in place)
            gathered = (h * w) % 4 != 0
            keep = dict(key=key, loss=loss, gt_rgb=torch.empty(Bmax, 3, h, w, device=dev) if gathered else None,
                        gt_depth=torch.empty(Bmax, 1, h, w, device=dev) if gathered else None, bufs=loss.alloc_batch(Bmax),
                        batch=None)
            self._batched_cache = keep
        self._loss, gt_rgb, gt_depth, bufs = keep["loss"], keep["gt_rgb"], keep["gt_depth"], keep["bufs"]
        self._loss_bufs = []
        total = self.cfg["optimization_steps"] if steps is None else steps
        losses = torch.empty(max(total, 1), device=dev)
        loss_now = torch.empty((), device=dev)
        from . import _lib
        _lib.zero_many(optim.state_buffers() + [slab.flat, rows.buf, losses, loss_now, self._loss.accum])
        state = dict(batch=None, idx=None, B=0)
        cached = keep["batch"]
        if (cached is not None and cached.capacity_n >= n and cached.max_instances >= self._cap
                and cached.binning_mode == self.binning_mode):
            cached.bind(g)
            state["batch"] = cached
            self._cap = cached.max_instances

        optim.zero_grad = True          # the row-set Adam leaves the slab clean for the next iteration
        fast_stage = (h * w) % 4 == 0

        def iteration(loss_out, staged=False, next_uniforms=False):
            """everything of one optimisation step that runs on the GPU (9-10 launches); inputs:
            state['idx'] (device), output: per-frame errors and the loss value.  ``staged``: the previous iteration's last
            launch has already drawn this one's frames and staged their matrices; ``next_uniforms`` (a row of uniforms,
            None, or False = no): this iteration's last launch does that for the next one."""
            batch, idx, B = state["batch"], state["idx"], state["B"]
            if staged:
                pass
            elif fast_stage:
                # the sampled frames' matrices into the batch; their images stay where they are (the loss stages read view
                # v's ground truth at frame idx[v] of the keyframe store: no 4 MB copy per view and iteration)
                self._loss.stage_frames(B, idx, all_view, all_proj, None, None, batch.viewmats, batch.projmats, None, None)
            else:
                torch.index_select(all_view, 0, idx, out=batch.viewmats[:B])
                torch.index_select(all_proj, 0, idx, out=batch.projmats[:B])
                torch.index_select(all_rgb, 0, idx, out=gt_rgb[:B])
                torch.index_select(all_depth, 0, idx, out=gt_depth[:B])
                self._loss.msum.zero_()
            # stage 1 of the loss head in the forward blend kernel's epilogue (ags_forward_batch_loss): one launch and one
            # re-read of four images less per iteration; bit-identical n_img / d_rgb / d_depth / msum
            fuse = fast_stage and self.FUSE_LOSS_STAGE1
            batch.forward(B, touched=rows, loss=self._loss.epilogue(all_rgb, all_depth, bufs, gt_index=idx) if fuse else None)
            images = batch._structs()[0]
            if fast_stage:
                if not fuse:
                    self._loss.stage1_batch(images, all_rgb, all_depth, bufs, B, gt_index=idx)
                self._loss.stage2_batch(images, all_depth, bufs, B, gt_index=idx)
            else:
                self._loss.stage1_batch(images, gt_rgb, gt_depth, bufs, B)
                self._loss.stage2_batch(images, gt_depth, bufs, B)
            if self.MULTI_VIEW_ROWS and B <= 16:
                # the blend backward of all views, then ONE per-Gaussian launch over the member rows: the views' chain rules
                # meet in registers and the Adam update follows in the same lane (no gradient slab, no atomics into it, no
                # separate optimiser kernel)
                batch.backward(B, bufs.d_rgb, bufs.d_normal, bufs.d_depth, self._no_grads, adam_tick=optim.tick_args(),
                               defer_rows=True)
                api.backward_rows((batch.view_refs(), B), batch.g, self._no_grads, rows,
                                  adam_clock=optim.tick_args(), fused_adam=(optim.tensors_struct(slab.as_list()), optim.eps))
            else:
                batch.backward(B, bufs.d_rgb, bufs.d_normal, bufs.d_depth, slab.grads, touched=rows,
                               adam_tick=optim.tick_args())
                optim.step(slab.as_list(), device_clock=True, pre_ticked=True)
            if next_uniforms is False:
                self._loss.finish(B, idx, self.training_performance, loss_out)
            else:
                self._loss.finish_next(B, idx, self.training_performance, loss_out, next_uniforms, n_old, n_random, n_active,
                                       all_view, all_proj, batch.viewmats, batch.projmats)

        def fits() -> bool:
            """forward-only probe of the staged batch: were the per-view workspaces large enough?"""
            batch, idx, B = state["batch"], state["idx"], state["B"]
            torch.index_select(all_view, 0, idx, out=batch.viewmats[:B])
            torch.index_select(all_proj, 0, idx, out=batch.projmats[:B])
            batch.forward(B)
            stw = batch.statuses(B)
            need = int(stw[:, 7].max())
            if need <= self._cap:
                return True
            self._grow_cap(need, int(stw[:, 0].max()))
            if state["batch"] is not None and state["batch"].binning_mode != self.binning_mode:
                state["batch"] = None          # one-pass binning was left: the batch is rebuilt in the other mode
            return False

        graph = None                       # (accum was zeroed above; from here on ags_loss_finish leaves it zeroed)
        # (the device draw is the weighted one; the uniform sampler keeps torch.randperm's host stream)
        device_sampler = self._device_sampler()
        n_active, n_random = len(sampler.active_ids), sampler.num_random
        n_old = len(sampler.older_ids)
        will_graph = self.use_graph and total >= self.graph_min_steps
        # the iterations of a call chained on the device: the last launch of one (ags_loss_finish_next) writes the frame
        # errors, draws the next iteration's frames from them and stages their matrices; the uniforms of all the call's
        # draws come from ONE torch.rand.  (A recorded graph replays a fixed launch sequence with fixed pointers: it keeps
        # the separate draw in front of every replay.)
        perf = self.training_performance
        chained = (device_sampler and fast_stage and not will_graph and n_old <= 8192 and perf.dtype == torch.float32
                   and perf.is_contiguous())
        uniforms = torch.rand(total, max(n_old, 1), device=dev) if chained and n_random > 0 else None
        self._phase("iterations")
        for it in range(total):
            staged = chained and it > 0
            if staged:
                B = state["B"]
            elif device_sampler:
                # the same draw as np.random.choice(older, n_random, replace=False, p = error / sum) - successive
                # sampling without replacement == the n_random largest log(u_i) / w_i (Efraimidis-Spirakis) -
                # from torch's device generator, so the host never reads the errors back
                B = n_active + n_random
                if state["idx"] is None or B != state["B"]:
                    state["idx"], state["B"], graph = torch.empty(B, device=dev, dtype=torch.long), B, None
                    act = [int(x) for x in sampler.active_ids]
                    if act and act == list(range(act[0], act[0] + len(act))):
                        # (the newest frames: written on the device - an upload from pageable memory waits for the stream)
                        torch.arange(act[0], act[0] + len(act), out=state["idx"][:n_active])
                    else:
                        state["idx"][:n_active] = torch.as_tensor(sampler.active_ids, dtype=torch.long)
                if n_random > 0 and uniforms is not None:
                    from . import _lib
                    _lib.check(_lib.load().ags_weighted_topk(_lib.ptr(uniforms[0]), _lib.ptr(perf), n_old, n_random,
                                                             _lib.ptr(state["idx"][n_active:]), _lib.current_stream()),
                               "ags_weighted_topk")
                elif n_random > 0:
                    weighted_choice_into(self.training_performance[:n_old], n_random, state["idx"][n_active:])
            else:
                ids = sampler.next_ids(self.training_performance)   # host read of the errors
                B = len(ids)
                if state["idx"] is None or B != state["B"]:
                    state["idx"], state["B"], graph = torch.empty(B, device=dev, dtype=torch.long), B, None
                state["idx"].copy_(torch.as_tensor(ids, dtype=torch.long))
            self._loss.set_batch_total(B)
            if graph is not None:
                graph.replay()
            else:
                # the workspace probe (a forward + a status read-back) is skipped when the previous train() call's
                # views needed well under what the workspace holds - an overflow is still caught after the loop
                known_fit = self._last_need is not None and 1.5 * self._last_need * (n / max(self._last_need_n, 1)) <= self._cap
                while state["batch"] is None or (it == 0 and not known_fit and not fits()):
                    keep["batch"] = None               # release the old buffers before the larger ones are made
                    # head-room for a map that doubles and for tile lists four times the average (one-pass binning
                    # needs tiles x the LONGEST list): a re-allocation is ~4 ms of hipMalloc + workspace set-up, and
                    # with 30 % head-room the mapper loop paid it on two of three keyframes (memory is not the
                    # constraint on a 288 GB part: ~0.1 GB per view at 512x512 and 260 k surfels)
                    ncap = max(2 * n + 4096, 1 << 18)
                    self._cap = max(self._cap, 4 * ncap)
                    state["batch"] = keep["batch"] = api.ViewBatch(
                        g, Bmax, h, w, cam0.tanfovx, cam0.tanfovy, self.background, self._cap,
                        binning_mode=self.binning_mode, capacity_n=ncap, tuning=self.tuning)
                nxt = False
                if chained and it + 1 < total:
                    nxt = None if uniforms is None else uniforms[it + 1]
                iteration(losses[it:it + 1], staged=staged, next_uniforms=nxt)
                if will_graph and it + 1 < total:
                    # for a given batch size the iteration is a fixed launch sequence: record it once
                    # (capture costs a few ms: it pays for long train() calls, not for the mapper's 10)
                    torch.cuda.synchronize()
                    side = torch.cuda.Stream()
                    side.wait_stream(torch.cuda.current_stream())
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.stream(side):
                        with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
                            iteration(loss_now)
                    torch.cuda.current_stream().wait_stream(side)
                continue
            losses[it].copy_(loss_now)
        batch = state["batch"]
        self._losses_host, self._losses_dev = [], losses[:total]      # (read when somebody asks: last_losses)

        def evaluate(status) -> bool:
            """status: (views, 8) words of every slot of the batch (sticky: any pass of any iteration)"""
            self._last_need, self._last_need_n = int(status[:, 4].max()), n
            if bool(status[:, 5].any()):
                self._grow_cap(self._last_need, int(status[:, 0].max()))
                self._last_need = None
                keep["batch"] = None                      # too small: the repeat allocates a larger one
                return False
            return True

        def settle() -> bool:
            """the wait of the call: did any pass of any iteration outgrow its workspace?"""
            return evaluate(batch.statuses()) if batch is not None else True
        settle.batch, settle.evaluate = batch, evaluate
        return settle if defer else settle()

    # ------------------------------------------------------------------ hipGraph iteration
    def _graph_ok(self) -> bool:
        if not self.use_graph:
            return False
        return self._uniform_frames()

    def train_graph(self, steps: Optional[int] = None):
        """Same iteration as ``train`` replayed from a hipGraph (single rank, frames of one shape
        and one field of view).  Falls back to ``train`` when those conditions do not hold."""
        self.settle()
        if not self._graph_ok():
            return self.train(steps)
        self._all_or_nothing(self._train_graph, steps)

    def _train_graph(self, steps: Optional[int] = None) -> bool:
        lrs = self.cfg["lrs"]
        for name in ("means", "scales", "rotations", "opacities", "harmonics"):
            setattr(self, name, getattr(self, name).contiguous())
        params = [self.means, self.scales, self.rotations, self.opacities, self.harmonics]
        n = self.means.shape[0]
        dev = self.device
        optim = FusedAdam(params, [lrs["mean"], lrs["scale"], lrs["rotation"], lrs["opacity"], lrs["harmonic"]], eps=1e-15)
        slab = GradSlab(n, dev)
        sampler = make_frame_sampler(self.cfg, self.frames)
        K = len(self.frames)
        h, w = self.frames[0]["rgb"].shape[-2:]
        cam0, fx, fy = self._camera(0)
        all_view, all_proj, all_rgb, all_depth = self._frame_store()
        B = sampler.num_random + len(sampler.active_ids)
        st_view = torch.empty(B, 4, 4, device=dev); st_proj = torch.empty(B, 4, 4, device=dev)
        st_rgb = torch.empty(B, 3, h, w, device=dev); st_depth = torch.empty(B, 1, h, w, device=dev)
        cams = [api.Camera(h, w, cam0.tanfovx, cam0.tanfovy, st_view[b], st_proj[b], self.background) for b in range(B)]
        self._loss = FusedLoss(h, w, fx, fy, B, self.cfg["batch_size"], dev)
        bufs = [self._loss.alloc_view() for _ in range(B)]
        self._cap = max(self._cap, 1 << 16, 2 * n)
        conf = self._confidences().contiguous()     # constant during train(): view stats change in post_processing
        g = api.Gaussians(self.means, self.scales, self.rotations, self.opacities, self.harmonics.view(n, 3), conf,
                          raw_params=True, scale_factor=self.cfg["scale_factor"], max_scale=0.05)

        def stage(ids):
            idx = torch.as_tensor(ids, device=dev, dtype=torch.long)
            torch.index_select(all_view, 0, idx, out=st_view); torch.index_select(all_proj, 0, idx, out=st_proj)
            torch.index_select(all_rgb, 0, idx, out=st_rgb); torch.index_select(all_depth, 0, idx, out=st_depth)

        def iteration(tick: bool):
            self._loss.begin_step()
            for b in range(B):
                st = self._state(b, n, h, w)
                api.forward(cams[b], g, st)
                self._loss.stage1(st, st_rgb[b], st_depth[b], bufs[b], b, b == 0)
            for b in range(B):
                st = self._states[b]
                self._loss.stage2(st, st_depth[b], bufs[b])
                api.backward(cams[b], g, st, bufs[b].d_rgb, bufs[b].d_normal, bufs[b].d_depth, None, None,
                             grads=slab.grads, accumulate=(b > 0),
                             adam_tick=optim.tick_args() if (tick and b == B - 1) else None)
            optim.step(slab.as_list(), device_clock=True, pre_ticked=tick)

        total = self.cfg["optimization_steps"] if steps is None else steps
        self.last_losses = []
        graph = None
        for it in range(total):
            ids = sampler.next_ids(self.training_performance)
            stage(ids)
            if it == 0:
                iteration(tick=True)   # eager first iteration: creates every buffer (an overflow shows in the sticky status)
            else:
                if graph is None:
                    side = torch.cuda.Stream()
                    side.wait_stream(torch.cuda.current_stream())
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.stream(side):
                        with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
                            iteration(tick=True)
                    torch.cuda.current_stream().wait_stream(side)
                graph.replay()
            self.training_performance[torch.as_tensor(ids, device=dev)] = self._loss.per_frame_errors(B)
            self.last_losses.append(self._loss.total_loss())
        if not self._check_capacity_sticky(range(B)):
            for b in range(B):
                self._states.pop(b)
            return False
        self.last_losses = [float(x) for x in self.last_losses]
        self._graph = graph
        return True
