"""Host-side mirror of the renderer facade and loss head around the rasterizer.

  SurfelRenderer        <-> GaussianRenderer            (/root/reference/utils/operations.py:723-904)
  render_core           <-> render_cuda_core            (operations.py:645-720)
  depth_to_normal       <-> depth2normal                (operations.py:172-219)
  training_losses       <-> the loss head of train()    (/root/reference/mapping/gaussian_map.py:106-124,
                                                         /root/reference/mapping/utils.py:14-62,120-121)

Same names of arguments, same 9-tuple order, same quirks (noted inline).  The rasterizer is
the drop-in module of this repository; a different module object can be injected (the
CPU tests inject an adapter over the oracle) — the default never falls back to anything.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from .camera import camera_matrices


def _default_module():
    import diff_gaussian_rasterization_2d as m
    return m


class _FacadePost(torch.autograd.Function):
    """normalize(normal) * mask and depth2normal on the GPU: one launch forward, one backward
    (``ags_facade_post[_backward]``, csrc/loss.hip) instead of ~35 torch ops and their autograd nodes per view."""

    @staticmethod
    def forward(ctx, normal_raw, depth, opacity, tanx, tany):
        import ctypes as C
        from . import _lib
        from ._lib import ptr
        _, H, W = depth.shape
        depth_c, op_c = depth.detach().float().contiguous(), opacity.detach().float().contiguous()
        nr_c = None if normal_raw is None else normal_raw.detach().float().contiguous()
        d2n = torch.empty(3, H, W, device=depth.device, dtype=torch.float32)
        n_out = None if nr_c is None else torch.empty_like(d2n)
        _lib.check(_lib.load().ags_facade_post(H, W, float(tanx), float(tany), ptr(nr_c), ptr(depth_c), ptr(op_c), ptr(n_out),
                                               ptr(d2n), _lib.current_stream()), "ags_facade_post")
        ctx.save_for_backward(*([depth_c, op_c] + ([nr_c] if nr_c is not None else [])))
        ctx.geom = (H, W, float(tanx), float(tany), nr_c is not None)
        if n_out is None:
            return d2n
        return n_out, d2n

    @staticmethod
    def backward(ctx, *grads):
        from . import _lib
        from ._lib import ptr
        H, W, tanx, tany, has_normal = ctx.geom
        saved = ctx.saved_tensors
        depth_c, op_c = saved[0], saved[1]
        nr_c = saved[2] if has_normal else None
        g_n, g_d = (grads[0], grads[1]) if has_normal else (None, grads[0])
        g_n = None if g_n is None else g_n.float().contiguous()
        g_d = None if g_d is None else g_d.float().contiguous()
        d_nr = torch.empty_like(nr_c) if (has_normal and ctx.needs_input_grad[0]) else None
        d_depth = torch.zeros(1, H, W, device=depth_c.device, dtype=torch.float32) if ctx.needs_input_grad[1] else None
        _lib.check(_lib.load().ags_facade_post_backward(H, W, tanx, tany, ptr(nr_c), ptr(depth_c), ptr(op_c), ptr(g_n),
                                                        ptr(g_d if d_depth is not None else None), ptr(d_nr), ptr(d_depth),
                                                        _lib.current_stream()), "ags_facade_post_backward")
        return d_nr, d_depth, None, None, None


def depth_to_normal(depth: torch.Tensor, mask: torch.Tensor, fov) -> torch.Tensor:
    """(1,H,W) depth, (1,H,W) bool mask -> (3,H,W) unit normals from the four neighbouring
    back-projected points.  Quirk kept from the reference: the x focal is derived from
    fov[0] and the image HEIGHT, the y focal from fov[1] and the WIDTH.  Device tensors take the one-launch
    kernel (and its one-launch backward); host tensors (the CPU tests) the torch statement below."""
    if depth.is_cuda:
        return _FacadePost.apply(None, depth, mask.to(torch.float32), math.tan(float(fov[0]) / 2.0),
                                 math.tan(float(fov[1]) / 2.0))
    return _depth_to_normal_torch(depth, mask, fov)


def _depth_to_normal_torch(depth: torch.Tensor, mask: torch.Tensor, fov) -> torch.Tensor:
    _, H, W = depth.shape
    d = depth[0]
    m = mask[0].to(d.dtype)
    fx = H / (2.0 * math.tan(float(fov[0]) / 2.0))
    fy = W / (2.0 * math.tan(float(fov[1]) / 2.0))
    ys, xs = torch.meshgrid(torch.arange(H, device=d.device, dtype=d.dtype),
                            torch.arange(W, device=d.device, dtype=d.dtype), indexing="ij")
    pts = torch.stack([(xs - 0.5 * W) * d / fx, (ys - 0.5 * H) * d / fy, d], 0)  # (3,H,W)
    pp = F.pad(pts[None], (1, 1, 1, 1), mode="replicate")[0]
    mp = F.pad(m[None, None], (1, 1, 1, 1), mode="replicate")[0, 0] > 0
    c = pts * mp[1:-1, 1:-1]
    up = (pp[:, :-2, 1:-1] - c) * mp[:-2, 1:-1]
    left = (pp[:, 1:-1, :-2] - c) * mp[1:-1, :-2]
    down = (pp[:, 2:, 1:-1] - c) * mp[2:, 1:-1]
    right = (pp[:, 1:-1, 2:] - c) * mp[1:-1, 2:]
    cr = lambda a, b: torch.linalg.cross(a, b, dim=0)
    n = cr(up, left) + cr(right, up) + cr(down, right) + cr(left, down)
    n = F.normalize(n, dim=0)
    return n * mask


def render_core(module, cam_pos, fov, view_matrix, projection_matrix, render_mask, image_shape, background_color,
                means, harmonics, opacities, confidences, scales, rotations, front_only=False,
                require_importance=False, weight_thres=0.03, tan_fov_host=None, config=None):
    """One view through the rasterizer + the facade's post-processing; returns the 9-tuple
    (rgb, depth, normal, opacity, d2n, confidence, importance, count, radii).
    ``tan_fov_host`` (two Python floats) / ``config`` (the 5-float device tensor): what the reference derives per call
    with two ``.item()`` read-backs and one host-to-device copy (operations.py:682-699), handed in by a caller that
    has them already (SurfelRenderer: once per batch)."""
    device = means.device
    if tan_fov_host is None:
        tan_fov = (0.5 * fov).tan()
        tan_fov_host = (tan_fov[0].item(), tan_fov[1].item())
    if config is None:
        config = torch.tensor([1.0, 1.0, 1.0, 1.0 if require_importance else 0.0, 1.0 if front_only else 0.0]).to(device)
    means_2d = torch.zeros_like(means, requires_grad=True)
    settings = module.GaussianRasterizationSettings(
        image_height=image_shape[0], image_width=image_shape[1], tanfovx=tan_fov_host[0],
        tanfovy=tan_fov_host[1], bg=background_color, scale_modifier=1.0, viewmatrix=view_matrix,
        projmatrix=projection_matrix, sh_degree=0, campos=cam_pos, prefiltered=False, render_mask=render_mask,
        weight_thres=weight_thres, debug=False, config=config)
    rgb, normal, depth, opacity, confidence, importance, count, radii = module.GaussianRasterizer(settings)(
        means3D=means, means2D=means_2d, opacities=opacities[..., None], confidences=confidences, shs=None,
        colors_precomp=harmonics[:, 0, :], scales=scales, rotations=rotations, cov3D_precomp=None)
    if depth.is_cuda:      # (the same two statements as one launch, one more for their backward)
        normal, d2n = _FacadePost.apply(normal, depth, opacity, settings.tanfovx, settings.tanfovy)
    else:
        mask = opacity.detach() > 1e-2
        normal = F.normalize(normal, dim=0) * mask
        d2n = depth_to_normal(depth, mask, fov)
    return rgb, depth, normal, opacity, d2n, confidence, importance, count, radii


# ---- the forward-only views of a renderer as ONE set of launches.  The planners build one renderer for ~100 candidate
# poses and call render_view(i) per candidate under no_grad (/root/reference/planning/confidence.py:24-46,
# exploration.py:24-44); evaluation, mesh extraction and the prune pass loop over views the same way.  One 128x128 view
# is 64 tiles - it cannot fill 256 CUs, and a launch set per view is host-bound besides (100 views one by one: 22 ms;
# as one batch: 1.5 ms).  The workspaces of a batch are the big allocation; they are kept (module level, a few entries)
# and re-bound to the next renderer's map, the OUTPUT images belong to the renderer that asked for them.
_BATCH_POOL = {}
_BATCH_POOL_MAX = 3
BATCH_BYTES_BUDGET = 12 << 30        # views per batched launch set are limited so that their workspaces stay below this


def release_batches() -> None:
    """Drop the pooled batch workspaces (at most _BATCH_POOL_MAX sets of up to BATCH_BYTES_BUDGET bytes are kept between
    renderers); the next forward-only renderer allocates again."""
    _BATCH_POOL.clear()


def _pooled_view_batch(g, key, V, h, w, tanx, tany, bg, cap, want_stats, front_only, has_mask, mode):
    from . import raster_api as api
    vb = _BATCH_POOL.get(key)
    if vb is not None and (vb.capacity_n < g.n or vb.max_instances < cap or vb.binning_mode != mode):
        _BATCH_POOL.pop(key)
        vb = None                       # release before the larger one is made
    if vb is None:
        while len(_BATCH_POOL) >= _BATCH_POOL_MAX:
            _BATCH_POOL.pop(next(iter(_BATCH_POOL)))
        masks = torch.zeros(V, h, w, device=g.means3D.device) if has_mask else None
        vb = api.ViewBatch(g, V, h, w, tanx, tany, bg, cap, want_stats=want_stats, front_only=front_only,
                           render_masks=masks, binning_mode=mode, capacity_n=max(int(1.5 * g.n) + 4096, 1 << 16))
        _BATCH_POOL[key] = vb
    else:
        _BATCH_POOL[key] = _BATCH_POOL.pop(key)       # most recently used last
        vb.bind(g)
        vb.cam.bg = bg
    return vb


class SurfelRenderer:
    """``GaussianRenderer`` (/root/reference/utils/operations.py:723-904): same constructor, same ``render_view`` /
    ``render_view_all`` / ``update_attr``, same 9-tuples.  Two things differ underneath, neither visible in the results:

    * passes WITHOUT grad (planners, evaluation, mesh, GUI, the count render of post-processing) of a renderer whose views
      share one field of view are rendered as one batch the first time one of them is asked for (``ViewBatch``:
      blockIdx.y = view) and served from it afterwards; ``update_attr`` forgets the batch;
    * passes WITH grad go view by view through the drop-in module with its workspace checks deferred to ONE wait per
      batch of views (``rasterizer.deferred_status``); a view that outgrew its workspace makes the loop run again with
      the size just learnt - the caller never sees truncated tile lists or an exception.
    A module injected by the caller (``rasterizer_module``: the CPU tests' adapter over the oracle) takes neither path."""

    def __init__(self, extrinsics, intrinsics, gaussians_attr, background_color, near_far, resolution, device,
                 render_masks=None, rasterizer_module=None):
        self._own_module = rasterizer_module is None
        self.module = rasterizer_module if rasterizer_module is not None else _default_module()
        self.device = device
        (self.gaussian_means, self.gaussian_harmonics, self.gaussian_opacities, self.gaussian_confidences,
         self.gaussian_scales, self.gaussian_rotations) = gaussians_attr
        self.background_color = background_color
        self.h, self.w = resolution
        self.batch_size = extrinsics.shape[0]
        self._intrinsic0 = intrinsics[0]
        cm = camera_matrices(extrinsics, intrinsics, near_far[0], near_far[1])
        self.cam_pos = cm["campos"]
        self.view_matrices = cm["viewmatrix"]
        self.projection_matrices = cm["projmatrix"]
        self.fovs = 2.0 * torch.atan(cm["tanfov"])
        self._tan_host = [(float(a), float(b)) for a, b in (0.5 * self.fovs).tan().cpu().tolist()]   # one read-back per batch
        self._configs = {}
        self._batched = {}           # (require_importance, front_only) -> batched outputs of all views
        self._has_masks = render_masks is not None
        if render_masks is None:
            self.render_masks = [torch.tensor([], device=device) for _ in range(self.batch_size)]
        else:
            self.render_masks = render_masks

    @property
    def raydir_map(self):
        """(3,H,W) unit ray directions of view 0 in camera space (operations.py:764-772 builds it in the constructor; the
        reference's only use of it is a dead ``visible_mask``, :716) - made when somebody asks."""
        if getattr(self, "_raydir", None) is None:
            h, w = self.h, self.w
            dev = self.view_matrices.device
            ys, xs = torch.meshgrid((torch.arange(h, device=dev, dtype=torch.float32) + 0.5) / h,
                                    (torch.arange(w, device=dev, dtype=torch.float32) + 0.5) / w, indexing="ij")
            pix = torch.stack([xs, ys, torch.ones_like(xs)], -1).reshape(-1, 3)
            d = pix @ torch.linalg.inv(self._intrinsic0.float().to(dev)).T
            self._raydir = F.normalize(d, dim=-1).reshape(h, w, 3).permute(2, 0, 1).contiguous()
        return self._raydir

    def update_attr(self, gaussians_attr):
        (self.gaussian_means, self.gaussian_harmonics, self.gaussian_opacities, self.gaussian_confidences,
         self.gaussian_scales, self.gaussian_rotations) = gaussians_attr
        self._batched = {}

    def _core(self, i, front_only, require_importance):
        key = (bool(require_importance), bool(front_only))
        if key not in self._configs:
            self._configs[key] = torch.tensor([1.0, 1.0, 1.0, 1.0 if require_importance else 0.0,
                                               1.0 if front_only else 0.0]).to(self.gaussian_means.device)
        return render_core(self.module, self.cam_pos[i], self.fovs[i], self.view_matrices[i],
                           self.projection_matrices[i], self.render_masks[i], (self.h, self.w),
                           self.background_color, self.gaussian_means, self.gaussian_harmonics,
                           self.gaussian_opacities, self.gaussian_confidences, self.gaussian_scales,
                           self.gaussian_rotations, front_only=front_only, require_importance=require_importance,
                           tan_fov_host=self._tan_host[i], config=self._configs[key])

    # ---- forward-only views as one batch
    def _batchable(self) -> bool:
        if not self._own_module or self.batch_size < 2 or not self.gaussian_means.is_cuda or len(self.gaussian_means) == 0:
            return False
        t0 = self._tan_host[0]
        return all(abs(t[0] - t0[0]) <= 1e-7 * abs(t0[0]) and abs(t[1] - t0[1]) <= 1e-7 * abs(t0[1]) for t in self._tan_host)

    def _render_batch(self, require_importance, front_only):
        """All views of this renderer, forward only, in chunks of as many views as BATCH_BYTES_BUDGET allows; returns
        the batched outputs (rgb, normal_raw, depth, opacity, confidence, importance, count, radii) of ALL views, owned by
        this renderer.  One wait per chunk (the views' status blocks): a chunk whose views outgrew their workspaces is
        rendered again with the capacity just learnt."""
        from . import raster_api as api
        dev = self.gaussian_means.device
        n, V, h, w = len(self.gaussian_means), self.batch_size, self.h, self.w
        f32 = lambda t: t.detach().float().contiguous()
        g = api.Gaussians(f32(self.gaussian_means), f32(self.gaussian_scales), f32(self.gaussian_rotations),
                          f32(self.gaussian_opacities).reshape(-1), f32(self.gaussian_harmonics[:, 0, :]),
                          f32(self.gaussian_confidences).reshape(-1))
        tanx, tany = self._tan_host[0]
        bg = f32(self.background_color.to(dev))
        masks = None
        if self._has_masks:
            masks = f32(torch.stack([m.to(dev).reshape(h, w) for m in self.render_masks]) if not torch.is_tensor(self.render_masks)
                        else self.render_masks.to(dev).reshape(V, h, w))
        st = getattr(SurfelRenderer, "_batch_caps", None)
        if st is None:
            st = SurfelRenderer._batch_caps = {}
        ckey = (dev.index, h, w)
        cap, mode = st.get(ckey, (max(1 << 16, 2 * n), api.BIN_DIRECT))
        cap = max(cap, 1 << 16, 2 * n)
        o = dict(device=dev, dtype=torch.float32)
        # (without statistics every view's importance / count are zeros, like the extension's: ONE row, shown V times -
        # an ``expand``ed view: READ-ONLY for the caller (torch refuses an in-place write on the batch; a write through a
        # per-view slice would show in every view).  The reference never writes to them: operations.py:845-851 stacks
        # and returns them, gaussian_map.py:195,229-232 only compares)
        stat_rows = V if require_importance else 1
        out = dict(rgb=torch.empty(V, 3, h, w, **o), normal=torch.empty(V, 3, h, w, **o), depth=torch.empty(V, 1, h, w, **o),
                   opacity=torch.empty(V, 1, h, w, **o), confidence=torch.empty(V, 1, h, w, **o),
                   importance=torch.zeros(stat_rows, n, **o).expand(V, n),
                   count=torch.zeros(stat_rows, n, device=dev, dtype=torch.int32).expand(V, n),
                   radii=torch.empty(V, n, device=dev, dtype=torch.int32))
        v0 = 0
        while v0 < V:
            per = api.workspace_bytes(max(int(1.5 * n) + 4096, 1 << 16), h, w, cap)
            CH = int(max(1, min(V - v0, BATCH_BYTES_BUDGET // max(per, 1))))
            CH = min(CH, 65535)
            key = (dev.index, CH, h, w, round(tanx, 7), round(tany, 7), bool(require_importance), bool(front_only),
                   masks is not None)
            vb = _pooled_view_batch(g, key, CH, h, w, tanx, tany, bg, cap, bool(require_importance), bool(front_only),
                                    masks is not None, mode)
            cnt = min(CH, V - v0)
            vb.viewmats[:cnt] = self.view_matrices[v0:v0 + cnt]
            vb.projmats[:cnt] = self.projection_matrices[v0:v0 + cnt]
            if masks is not None:
                vb.masks[:cnt] = masks[v0:v0 + cnt]
            vb.forward(cnt)
            stw = vb.statuses(cnt)                         # the one wait of the chunk
            need, inst = int(stw[:, 7].max()), int(stw[:, 0].max())
            if need > vb.max_instances:
                # one-pass binning needs tiles x the LONGEST tile list; badly skewed lists go on with the scan-based
                # binning, which needs the instance total (same images) - the rule of FusedMapTrainer._grow_cap
                if mode == api.BIN_DIRECT and need > 8 * max(inst, 1 << 16) and need * 24 > (1 << 30):
                    mode, need = api.BIN_TILE_SORT, inst
                cap = min(max(int(need * 1.5) + 4096, cap + 1), 0xFFFFFFFF)
                st[ckey] = (cap, mode)
                _BATCH_POOL.pop(key, None)
                continue
            st[ckey] = (max(cap, vb.max_instances), mode)
            for name in ("rgb", "normal", "depth", "opacity", "confidence"):
                out[name][v0:v0 + cnt] = getattr(vb, name)[:cnt]
            out["radii"][v0:v0 + cnt] = vb.radii[:cnt]
            if require_importance:
                out["importance"][v0:v0 + cnt] = vb.importance[:cnt]
                out["count"][v0:v0 + cnt] = vb.count[:cnt]
            v0 += cnt
        return out

    def _batched_view(self, i, require_importance, front_only):
        key = (bool(require_importance), bool(front_only))
        b = self._batched.get(key)
        if b is None:
            from . import _lib
            from ._lib import ptr
            r = self._render_batch(*key)
            # the facade's post-processing of ALL the views in one launch (ags_facade_post_batch), the in-frustum masks in one op
            V, h, w = self.batch_size, self.h, self.w
            normal, d2n = torch.empty_like(r["normal"]), torch.empty_like(r["normal"])
            tanx, tany = self._tan_host[0]
            _lib.check(_lib.load().ags_facade_post_batch(V, h, w, float(tanx), float(tany), ptr(r["normal"]), ptr(r["depth"]),
                                                         ptr(r["opacity"]), ptr(normal), ptr(d2n),
                                                         _lib.current_stream()), "ags_facade_post_batch")
            r["normal_post"], r["d2n"], r["seen"] = normal, d2n, r["radii"] > 0
            # the 9-tuples of all views from nine unbind() calls (a slice per output and view is 900 tensor ops per planning step)
            cols = [r[k].unbind(0) for k in ("rgb", "depth", "normal_post", "opacity", "d2n", "confidence", "importance", "count", "seen")]
            b = self._batched[key] = dict(raw=r, views=list(zip(*cols)))
        return b["views"][i]

    def render_view(self, i=0, require_grad=False, require_importance=False, front_only=False):
        if not require_grad and self._batchable():
            with torch.no_grad():
                return self._batched_view(i, require_importance, front_only)
        with torch.set_grad_enabled(require_grad):
            rgb, depth, normal, opacity, d2n, confidence, importance, count, radii = self._core(
                i, front_only, require_importance)
        return rgb, depth, normal, opacity, d2n, confidence, importance, count, radii > 0

    def _loop_views(self, require_grad, require_importance, front_only):
        per_view = []
        radii_sum = torch.zeros(len(self.gaussian_means), device=self.device, dtype=torch.int32)
        with torch.set_grad_enabled(require_grad):
            for i in range(self.batch_size):
                out = self._core(i, front_only, require_importance)
                per_view.append(out[:8])
                radii_sum = radii_sum + out[8].to(radii_sum.device)
        return per_view, radii_sum

    def render_view_all(self, require_grad=False, require_importance=False, front_only=False):
        if not require_grad and self._batchable():
            with torch.no_grad():
                per_view = [self._batched_view(i, require_importance, front_only) for i in range(self.batch_size)]
                radii_sum = self._batched[(bool(require_importance), bool(front_only))]["raw"]["seen"].sum(0, dtype=torch.int32)
        elif self._own_module and self.gaussian_means.is_cuda:
            # the module's workspace checks wait ONCE for the whole batch of views; a truncated view repeats the loop
            from .rasterizer import deferred_status
            for attempt in range(6):
                with deferred_status() as d:
                    per_view, radii_sum = self._loop_views(require_grad, require_importance, front_only)
                    ok = d.settle()
                if ok:
                    break
                del per_view, radii_sum
            else:
                raise RuntimeError("render_view_all: the rasterizer workspace kept overflowing after six enlargements: " +
                                   "; ".join(d.reports))
        else:
            per_view, radii_sum = self._loop_views(require_grad, require_importance, front_only)
        stack = lambda k: torch.stack([v[k] for v in per_view], 0)
        return (stack(0), stack(1), stack(2), stack(3), stack(4), stack(5), stack(6), stack(7), radii_sum > 0)


# ------------------------------------------------------------------------- loss head
def _directional_sq_diffs(x: torch.Tensor) -> torch.Tensor:
    """(B,C,H,W) -> (B,4,H,W): squared norm over C of the difference with the right, left,
    lower and upper neighbour (zero where the neighbour is outside)."""
    z = torch.zeros_like(x)
    to_right = z.clone(); to_right[..., :, :-1] = x[..., :, :-1] - x[..., :, 1:]
    to_left = z.clone(); to_left[..., :, 1:] = x[..., :, 1:] - x[..., :, :-1]
    to_down = z.clone(); to_down[..., :-1, :] = x[..., :-1, :] - x[..., 1:, :]
    to_up = z.clone(); to_up[..., 1:, :] = x[..., 1:, :] - x[..., :-1, :]
    return torch.stack([to_right, to_left, to_down, to_up], 2).pow(2).sum(1)


def normal_tv_loss(normals, depths, mask, sigma=0.3, batch_total=None):
    nd = _directional_sq_diffs(normals)
    dd = _directional_sq_diffs(depths.detach())
    flat = (dd <= 1e-4).to(nd.dtype)
    term = flat * torch.exp(-nd / (2 * sigma ** 2)) * nd * mask
    if batch_total is None:
        return torch.mean(term)
    return term.sum() / (batch_total * term[0].numel())


def training_losses(rgb_preds, depth_preds, normal_preds, opacity_preds, d2n_preds, rgb_gts, depth_gts,
                    batch_total=None, mask_vis_sum=None):
    """Returns (total, per_frame_error). Weights 1 / 0.8 / 0.1 / 0.1 (gaussian_map.py:119-124).

    Quirk kept from the reference: the consistency term (B,H,W) is multiplied by the
    visibility mask (B,1,H,W), which broadcasts to (B,B,H,W) before the mean, i.e.
    mean_{b,b'} cons[b'] * mask[b] = sum_{b'} cons[b'] * (sum_b mask[b]) / (B*B*H*W).
    View-parallel ranks pass ``batch_total`` = global B and ``mask_vis_sum`` = the
    all-reduced sum over ALL views of the visibility mask; every term is then this rank's
    additive share of the reference's batch loss."""
    B = rgb_preds.shape[0] if batch_total is None else batch_total
    mask_vis = opacity_preds.detach() > 1e-3
    mask_depth = depth_gts > 0.0
    rgb_l = torch.abs((rgb_preds - rgb_gts) * mask_vis)
    depth_l = torch.abs((depth_preds - depth_gts) * mask_depth)
    per_frame = rgb_l.detach().mean(dim=[1, 2, 3]) + depth_l.detach().mean(dim=[1, 2, 3])
    tv = normal_tv_loss(normal_preds, depth_preds, mask_depth, batch_total=B)
    cons = 1 - torch.sum(normal_preds * d2n_preds, 1)  # (b,H,W)
    msum = mask_vis.long().sum(0) if mask_vis_sum is None else mask_vis_sum  # (1,H,W)
    cons = (cons * msum).sum() / (B * B * cons[0].numel())
    total = rgb_l.sum() / (B * rgb_l[0].numel()) + 0.8 * depth_l.sum() / (B * depth_l[0].numel()) + 0.1 * cons + 0.1 * tv
    return total, per_frame
