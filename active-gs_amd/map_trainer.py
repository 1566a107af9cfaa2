"""Reference-compatible train loop: mirror of ``GaussianMap.train()`` /
``post_processing()`` / ``prune()`` / ``get_attr()`` (/root/reference/mapping/gaussian_map.py:
66-139, 141-246, 529-581) and of ``WeightedSampler`` (/root/reference/mapping/utils.py:190-228)
on top of this repository's rasterizer.  Autograd carries the gradients from the torch loss
head through the rasterizer's HIP backward into the raw parameters; the optimizer is the
fused Adam.  ``rasterizer_module`` / ``optimizer_factory`` exist so the CPU test-suite can
drive this host logic against the golden ``train()`` capture; the defaults are the HIP path
and raise without a GPU.
"""
from __future__ import annotations

from typing import Callable, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

from .dist_util import all_reduce_
from .facade import SurfelRenderer, training_losses

DEFAULT_CFG = dict(bound=(0.001, 10.0), scale_factor=0.01, optimization_steps=10, prune_interval=5, error_thres=0.25,
                   background=(0.0, 0.0, 0.0, 0.0), batch_size=8, active_size=3, use_view_distribution=True,
                   # "host": the reference's np.random.choice on the host (reads the per-frame errors back every
                   # iteration); "device" (FusedMapTrainer.train_batched): the same distribution drawn on the GPU
                   sampler="host", sampler_type="weighted",
                   lrs=dict(mean=5e-4, scale=1e-2, rotation=5e-4, opacity=1e-2, harmonic=1e-4))


class WeightedFrameSampler:
    """Newest ``active_size`` frames + up to ``batch_size-active_size`` error-weighted older
    frames drawn with ``np.random.choice(..., replace=False)`` (global numpy RNG, as the reference)."""

    def __init__(self, frames: list, batch_size: int, active_size: int):
        active = min(active_size, len(frames))
        ids = list(range(len(frames)))
        self.frames = frames
        self.active_ids = np.array(ids[-active:])
        self.older_ids = np.array(ids[:-active])
        self.num_random = min(len(self.older_ids), batch_size - active)

    def next_ids(self, weight: torch.Tensor):
        """the frame indices of the next batch (what the fused trainers need: they gather the frames on the device)"""
        sel = self.active_ids.copy()
        if self.num_random > 0:
            w = weight[self.older_ids]
            w = w / torch.sum(w)
            picked = np.random.choice(self.older_ids, size=self.num_random, p=w.cpu().numpy(), replace=False)
            sel = np.append(sel, self.older_ids[picked])  # the reference indexes by the drawn values
        return sel

    def next_frames(self, weight: torch.Tensor):
        sel = self.next_ids(weight)
        st = lambda k: torch.stack([self.frames[i][k] for i in sel])
        return st("rgb"), st("depth"), st("extrinsic"), st("intrinsic"), sel


class UniformFrameSampler:
    """Mirror of ``UniformSampler`` (/root/reference/mapping/utils.py:231-261): the newest ``active_size`` frames plus
    ``batch_size - active_size`` older ones drawn with ``torch.randperm`` (torch's global stream).  ``frames`` is
    a dict keyed by frame id, as there (a list is taken as ``{0: f0, 1: f1, ...}``).  The reference's ``train()``
    calls ``next_frames(weight)`` on whichever sampler the config names (gaussian_map.py:81), which this class's
    original does not accept - ``sampler_type: uniform`` is dead there; here the weight is accepted and ignored so
    that ``cfg["sampler_type"] = "uniform"`` works in the train loops."""

    def __init__(self, frames, batch_size: int, active_size: int):
        self.frames = frames if isinstance(frames, dict) else dict(enumerate(frames))
        ids = list(self.frames.keys())
        assert len(ids) >= active_size
        self.active_ids = np.array(ids[-active_size:])
        self.older_ids = np.array(ids[:-active_size])
        self.num_random = min(len(self.older_ids), batch_size - active_size)
        self.v = len(self.active_ids) + self.num_random

    def next_ids(self, weight: Optional[torch.Tensor] = None):
        sel = self.active_ids.copy()
        if self.num_random > 0:
            picked = torch.randperm(len(self.older_ids))[: self.num_random]
            sel = np.append(sel, self.older_ids[picked.numpy()])
        return sel

    def next_frames(self, weight: Optional[torch.Tensor] = None):
        sel = self.next_ids(weight)
        st = lambda k: torch.stack([self.frames[i][k] for i in sel])
        return st("rgb"), st("depth"), st("extrinsic"), st("intrinsic"), sel


def make_frame_sampler(cfg: dict, frames):
    """``cfg["sampler_type"]``: "weighted" (the reference's default, incremental.yaml:23) or "uniform"
    (gaussian_map.py:253-256)."""
    kind = cfg.get("sampler_type", "weighted")
    if kind == "uniform":
        return UniformFrameSampler(frames, cfg["batch_size"], min(cfg["active_size"], len(frames)))
    if kind != "weighted":
        raise ValueError(f"unknown sampler_type {kind!r}")
    return WeightedFrameSampler(frames, cfg["batch_size"], cfg["active_size"])


def _default_optimizer(params, lrs):
    from .optimizer import FusedAdam
    return FusedAdam(params, lrs, eps=1e-15)


class GaussianMapTrainer:
    def __init__(self, raw: dict, frames: List[dict], cfg: Optional[dict] = None, rasterizer_module=None,
                 optimizer_factory: Optional[Callable] = None, process_group=None):
        self.cfg = {**DEFAULT_CFG, **(cfg or {})}
        self.means = raw["means"].clone()
        self.scales = raw["scales"].clone()
        self.rotations = raw["rotations"].clone()
        self.opacities = raw["opacities"].clone()
        self.harmonics = raw["harmonics"].clone()
        n = self.means.shape[0]
        dev = self.means.device
        self.device = dev
        self.view_scores = raw.get("view_scores", torch.zeros(n, device=dev)).clone()
        self.view_supports = raw.get("view_supports", torch.zeros(n, device=dev)).clone()
        self.view_means = raw.get("view_means", torch.zeros(n, 3, device=dev)).clone()
        self.frames = frames
        self.training_performance = torch.full((len(frames),), 10.0, device=dev)
        self.module = rasterizer_module
        self.optimizer_factory = optimizer_factory or _default_optimizer
        self.background = torch.tensor(self.cfg["background"], dtype=torch.float32, device=dev)
        self.last_losses: List[float] = []
        # view-parallel data parallelism: parameters replicated, the views of
        # an iteration sharded rank::world, ONE all-reduce of the gradients before Adam.
        self.pg = process_group
        dist = torch.distributed
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if self.world > 1 else 0

    # ---- activations (gaussian_map.py:529-581)
    def confidences(self):
        if self.cfg["use_view_distribution"]:
            var = self.view_means.norm(dim=-1)
            var = torch.where(torch.isnan(var), torch.ones_like(var), var)
            return torch.clamp(torch.exp(1 - var) * self.view_scores, min=0, max=1)
        return torch.clamp(1 - 1 / torch.exp(self.view_supports), min=0, max=1)

    def attr(self, params):
        means, scales, rotations, opacities, harmonics = params
        return (means, harmonics, torch.sigmoid(opacities), self.confidences(),
                torch.clamp(self.cfg["scale_factor"] * torch.exp(scales), min=0, max=0.05),
                F.normalize(rotations))

    def _renderer(self, extr, intr, params, hw, masks=None):
        return SurfelRenderer(extr, intr, self.attr(params), self.background, self.cfg["bound"], hw, self.device,
                              render_masks=masks, rasterizer_module=self.module)

    # ---- train() (gaussian_map.py:66-130)
    def train(self, steps: Optional[int] = None):
        params = [self.means, self.scales, self.rotations, self.opacities, self.harmonics]
        lrs = self.cfg["lrs"]
        optim = self.optimizer_factory(params, [lrs["mean"], lrs["scale"], lrs["rotation"], lrs["opacity"],
                                                lrs["harmonic"]])  # fresh state per train(), :259-292
        sampler = make_frame_sampler(self.cfg, self.frames)
        self.last_losses = []
        dist = torch.distributed
        for _ in range(self.cfg["optimization_steps"] if steps is None else steps):
            rgb_gt, depth_gt, extr, intr, ids = sampler.next_frames(self.training_performance)
            B = rgb_gt.shape[0]
            h, w = rgb_gt.shape[-2:]
            mine = list(range(self.rank, B, self.world))          # this rank's views
            leaves = [p.detach().requires_grad_(True) for p in params]
            if mine:
                rgb, depth, normal, opacity, d2n, *_ = self._renderer(extr[mine], intr[mine], leaves, (h, w)).render_view_all(
                    require_grad=True)
                msum = (opacity.detach() > 1e-3).long().sum(0)
            else:
                msum = torch.zeros(1, h, w, dtype=torch.long, device=self.device)
            if self.world > 1:
                all_reduce_(msum, group=self.pg)
            per_frame_all = torch.zeros(B, device=self.device)
            if mine:
                total, per_frame = training_losses(rgb, depth, normal, opacity, d2n, rgb_gt[mine], depth_gt[mine],
                                                   batch_total=B, mask_vis_sum=msum)
                per_frame_all[mine] = per_frame
                total.backward()
                grads = [l.grad if l.grad is not None else torch.zeros_like(l) for l in leaves]
                loss_val = total.detach().reshape(1).clone()
            else:
                grads = [torch.zeros_like(l) for l in leaves]
                loss_val = torch.zeros(1, device=self.device)
            if self.world > 1:
                flat = torch.cat([g.reshape(-1) for g in grads] + [per_frame_all, loss_val])
                all_reduce_(flat, group=self.pg)            # one collective: 14N grads + B errors + loss
                o = 0
                for i, g in enumerate(grads):
                    grads[i] = flat[o:o + g.numel()].view_as(g)
                    o += g.numel()
                per_frame_all, loss_val = flat[o:o + B], flat[o + B:o + B + 1]
            self.training_performance[torch.as_tensor(ids, device=self.device)] = per_frame_all
            optim.step(grads)
            self.last_losses.append(float(loss_val))
        self.post_processing()

    # ---- post_processing() (gaussian_map.py:141-232)
    def post_processing(self):
        k = len(self.frames)
        prune_now = k % self.cfg["prune_interval"] == 0
        use = list(range(k)) if prune_now else [k - 1]
        st = lambda key: torch.stack([self.frames[i][key] for i in use])
        extr, intr, depth_gt = st("extrinsic"), st("intrinsic"), st("depth")
        h, w = depth_gt.shape[-2:]
        params = [self.means, self.scales, self.rotations, self.opacities, self.harmonics]
        with torch.no_grad():
            mine = list(range(self.rank, len(use), self.world))
            n = self.means.shape[0]
            if mine:
                counts = self._render_counts([use[m] for m in mine], extr[mine], intr[mine], depth_gt[mine], params, (h, w))
            else:
                counts = torch.zeros(0, n, dtype=torch.int32, device=self.device)
            counts_sum = counts.sum(0).to(torch.int32)
            newest_here = bool(mine) and mine[-1] == len(use) - 1
            seen_i = (counts[-1] >= 1.0).to(torch.int32) if newest_here else torch.zeros(n, dtype=torch.int32, device=self.device)
            if self.world > 1:
                both = torch.stack([counts_sum, seen_i])
                all_reduce_(both, group=self.pg)  # counts over all views + newest view's row
                counts_sum, seen_i = both[0], both[1]
            seen = seen_i > 0
            self.view_supports += seen.float()
            if self.cfg["use_view_distribution"]:
                normals = F.normalize(_quat_third_column(F.normalize(self.rotations)))
                to_cam = extr[-1:, :3, 3] - self.means
                dist = torch.linalg.norm(to_cam, dim=1)
                to_cam = to_cam / dist.unsqueeze(-1)
                # masked updates written with where(): boolean indexing would cost a host sync (nonzero) each
                step = (to_cam - self.view_means) / self.view_supports.clamp(min=1.0).unsqueeze(-1)
                self.view_means = torch.where(seen.unsqueeze(-1), self.view_means + step, self.view_means)
                cos = torch.clamp(torch.sum(normals * to_cam, 1), min=0, max=1)
                far = self.frames[-1]["depth_range"][1]
                gain = (1 - torch.clamp(dist / far, min=0, max=1)) * cos
                self.view_scores = self.view_scores + torch.where(seen, gain, torch.zeros_like(gain))
            if prune_now:
                self.prune(~(counts_sum >= 1.0))

    def _render_counts(self, frame_ids, extr, intr, depth_gt, params, hw):
        """(len(frame_ids), N) int32: per view, pixels (with valid ground-truth depth) in which each
        front-facing surfel has blend weight > weight_thres (gaussian_map.py:173-192)."""
        return self._renderer(extr, intr, params, hw, masks=(depth_gt > 0.0).float()).render_view_all(
            require_importance=True, front_only=True)[7]

    def prune(self, mask):
        mask = mask | (torch.sigmoid(self.opacities) < 0.1)
        keep = ~mask
        for name in ("means", "scales", "rotations", "opacities", "harmonics", "view_scores", "view_supports",
                     "view_means"):
            setattr(self, name, getattr(self, name)[keep])
        return int(mask.sum())


def _quat_third_column(q):
    r, x, y, z = q.unbind(-1)
    return torch.stack([2 * (x * z + r * y), 2 * (y * z - r * x), 1 - 2 * (x * x + y * y)], -1)
