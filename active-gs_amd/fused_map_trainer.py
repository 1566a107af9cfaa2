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

from . import raster_api as api
from .camera import camera_matrices
from .fused_loss import FusedLoss
from .map_trainer import GaussianMapTrainer, WeightedFrameSampler
from .optimizer import FusedAdam
from .trainer import GradSlab


class FusedMapTrainer(GaussianMapTrainer):
    def __init__(self, raw: dict, frames: List[dict], cfg: Optional[dict] = None, process_group=None,
                 binning_mode: int = api.BIN_TILE_SORT):
        super().__init__(raw, frames, cfg, process_group=process_group)
        if self.device.type != "cuda":
            raise RuntimeError("FusedMapTrainer needs GPU tensors: there is no CPU fallback")
        self.binning_mode = binning_mode
        self._cams = {}          # frame index -> (api.Camera, fov_x, fov_y)
        self._states = {}        # view slot -> ForwardState
        self._loss = None
        self._loss_bufs = []
        self._cap = 0

    # ---- cached per-frame camera (the intrinsics -> fov step needs host scalars once per frame)
    def _camera(self, idx: int):
        c = self._cams.get(idx)
        if c is None:
            f = self.frames[idx]
            h, w = f["rgb"].shape[-2:]
            cm = camera_matrices(f["extrinsic"][None].float(), f["intrinsic"][None].float(), *self.cfg["bound"])
            tan = cm["tanfov"][0].cpu()
            cam = api.Camera(h, w, float(tan[0]), float(tan[1]), cm["viewmatrix"][0].contiguous(),
                             cm["projmatrix"][0].contiguous(), self.background)
            fov = (2.0 * torch.atan(tan)).tolist()
            c = (cam, fov[0], fov[1])
            self._cams[idx] = c
        return c

    def _state(self, slot: int, n: int, h: int, w: int) -> api.ForwardState:
        st = self._states.get(slot)
        if st is None or st.max_instances < self._cap or st.radii.shape[0] != n or st.rgb.shape[-2:] != (h, w):
            st = api.alloc_state(n, h, w, self._cap, self.device, self.binning_mode)
            self._states[slot] = st
        return st

    def _gaussians(self) -> api.Gaussians:
        n = self.means.shape[0]
        return api.Gaussians(self.means, self.scales, self.rotations, self.opacities, self.harmonics.view(n, 3),
                             self.confidences().contiguous(), raw_params=True, scale_factor=self.cfg["scale_factor"],
                             max_scale=0.05)

    def _check_capacity(self, slots) -> bool:
        need = max((api.read_status(self._states[s])["num_instances"] for s in slots), default=0)
        if need > self._cap:
            self._cap = int(need * 1.5) + 4096
            return False
        return True

    def train(self, steps: Optional[int] = None):
        dist = torch.distributed
        lrs = self.cfg["lrs"]
        for name in ("means", "scales", "rotations", "opacities", "harmonics"):
            setattr(self, name, getattr(self, name).contiguous())
        params = [self.means, self.scales, self.rotations, self.opacities, self.harmonics]
        n = self.means.shape[0]
        optim = FusedAdam(params, [lrs["mean"], lrs["scale"], lrs["rotation"], lrs["opacity"], lrs["harmonic"]], eps=1e-15)
        slab = GradSlab(n, self.device)
        sampler = WeightedFrameSampler(self.frames, self.cfg["batch_size"], self.cfg["active_size"])
        self.last_losses = []
        self._cap = max(self._cap, 1 << 16, 2 * n)
        for it in range(self.cfg["optimization_steps"] if steps is None else steps):
            _, _, _, _, ids = sampler.next_frames(self.training_performance)
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
            while True:  # forward every local view; re-run the batch once if a workspace was too small
                self._loss.begin_step()
                for slot, b in enumerate(mine):
                    cam, _, _ = self._camera(int(ids[b]))
                    st = self._state(slot, n, h, w)
                    api.forward(cam, g, st)
                    f = self.frames[int(ids[b])]
                    self._loss.stage1(st, f["rgb"], f["depth"], self._loss_bufs[slot], b, slot == 0)
                if it > 0 or self._check_capacity(range(len(mine))):
                    break
            if not mine:
                self._loss.msum.zero_()
            if self.world > 1:
                dist.all_reduce(self._loss.msum, group=self.pg)
            for slot, b in enumerate(mine):
                cam, _, _ = self._camera(int(ids[b]))
                st, buf = self._states[slot], self._loss_bufs[slot]
                self._loss.stage2(st, self.frames[int(ids[b])]["depth"], buf)
                api.backward(cam, g, st, buf.d_rgb, buf.d_normal, buf.d_depth, None, None, grads=slab.grads,
                             accumulate=(slot > 0))
            if not mine:
                slab.flat.zero_()
            if self.world > 1:
                dist.all_reduce(slab.flat, group=self.pg)
                dist.all_reduce(self._loss.accum, group=self.pg)
            self.training_performance[torch.as_tensor(ids, device=self.device)] = self._loss.per_frame_errors(B)
            optim.step(slab.as_list())
            self.last_losses.append(self._loss.total_loss())
        if not self._check_capacity(self._states.keys()):
            raise RuntimeError("a view outgrew the rasterizer workspace during train(); call train() again "
                               "(the capacity has been raised)")
        self.last_losses = [float(x) for x in self.last_losses]
        self.post_processing()
