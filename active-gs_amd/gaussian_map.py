"""``GaussianMap`` with the reference's surface (/root/reference/mapping/gaussian_map.py:17-590) over the fused trainer.

The one-line change that gives an UNTOUCHED ``mapping.Mapper`` and the planners the fused path::

    # /root/reference/mapping/gaussian_map.py
    from active_gs_amd.gaussian_map import GaussianMap      # instead of the class defined there

``IncrementalMapper.init_map`` builds ``GaussianMap(self.cfg.gaussian_map, self.device)`` and calls
``.update(dataframe)`` (mapping/mapper.py:44,101); planners, evaluation and mesh extraction read ``.get_attr()``,
``.background_color``, ``.scene_near / .scene_far`` (planning/confidence.py:27-29, utils/evaluation_tool.py:125-127,
mesh_generation.py:77-79); the voxel map reads the properties ``.get_means / .get_normals / .get_confidences /
.get_opacities`` (mapping/voxel_map.py:71-74); the GUI packet copies all seven ``get_*`` (utils/common.py:109-116); the
recorder calls ``.save(path, index=)`` (utils/common.py:249); ``eval.py`` / ``mesh_generation.py`` / ``visualize.py`` do
``GaussianMap(None, device).load(file)``.  All of that is here, same names, same argument meaning; what runs underneath is
``FusedMapTrainer`` (train loop, loss head, Adam, post-processing, growth and pruning on the C ABI: no torch autograd, no
per-view host synchronisation) instead of ~80 autograd-recorded extension calls per keyframe.

State lives in ONE place - the trainer - and the reference's attribute names are views of it (``_means`` ... ``view_means``,
``training_data``, ``training_performance``, ``is_init``), readable and assignable like the originals (``load()`` and the
fixture generators assign them).  cfg is read the way gaussian_map.py:40-52 reads it: attribute access on
``cfg.bound``, ``cfg.background``, ``cfg.optimizer.*``, ``cfg.sampler.*`` (an OmegaConf node, a SimpleNamespace or anything
else with those attributes).

There is no CPU fallback: a map on a CPU device can be constructed, loaded, saved and read (the getters are the
reference's one-line torch expressions), but ``update / train / add_gaussians / post_processing / prune`` raise without a GPU.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn.functional as F

from .map_trainer import DEFAULT_CFG, make_frame_sampler

_RAW = ("means", "scales", "rotations", "opacities", "harmonics")
_VIEW = ("view_scores", "view_supports", "view_means")


def _cfg_get(node, name, default=None):
    """``node.name`` for attribute-style configs (OmegaConf / SimpleNamespace), ``node[name]`` for mappings."""
    if node is None:
        return default
    if isinstance(node, dict):
        return node.get(name, default)
    try:
        return getattr(node, name)
    except (AttributeError, KeyError):
        try:
            return node[name]
        except Exception:
            return default


def _state_property(name):
    def get(self):
        if self._trainer is not None:
            self._trainer.settle()       # (a train() call whose workspace check is still pending: FusedMapTrainer.settle)
            return getattr(self._trainer, name)
        return self._cold[name]

    def set_(self, value):
        if self._trainer is not None:
            self._trainer.settle()
            setattr(self._trainer, name, value)
            self._trainer._states.clear()        # per-view workspaces are laid out for the row count
        else:
            self._cold[name] = value
    return property(get, set_)


class GaussianMap:
    # the reference's names for the map state (gaussian_map.py:21-33) as views of the trainer's
    _means = _state_property("means")
    _scales = _state_property("scales")
    _rotations = _state_property("rotations")
    _opacities = _state_property("opacities")
    _harmonics = _state_property("harmonics")
    view_scores = _state_property("view_scores")
    view_supports = _state_property("view_supports")
    view_means = _state_property("view_means")
    training_performance = _state_property("training_performance")

    FRAME_SAMPLER = "device"     # where the error-weighted frames of a batch are drawn unless cfg.sampler.draw says otherwise

    def __init__(self, cfg, device, process_group=None):
        """``cfg``, ``device``: the reference's arguments (gaussian_map.py:18).  ``process_group`` (not in the reference, which
        is single process): a torch.distributed group over which the views of every training iteration are sharded
        (rank r renders views r::world, gradients exchanged once per iteration, parameters and Adam replicated)."""
        self.device = torch.device(device)
        self.process_group = process_group
        dev = self.device
        # before the trainer exists (a CPU map, or nothing has needed it yet) the state sits here
        self._cold = dict(means=torch.empty(0, 3, device=dev), scales=torch.empty(0, 3, device=dev),
                          rotations=torch.empty(0, 4, device=dev), opacities=torch.empty(0, device=dev),
                          harmonics=torch.empty(0, 1, 3, device=dev), view_scores=torch.empty(0, device=dev),
                          view_supports=torch.empty(0, device=dev), view_means=torch.empty((0, 3), device=dev),
                          training_performance=torch.tensor([], device=dev))
        self._trainer = None
        self._frames = []
        self._is_init = False
        self.use_view_distribution = True
        self.cfg = cfg
        # defaults of config/mapper/incremental.yaml:12-32 for maps made with cfg = None (eval / mesh / visualize: load())
        self.scene_near, self.scene_far = DEFAULT_CFG["bound"]
        self.sparse_ratio = 0.1
        self.scale_factor = DEFAULT_CFG["scale_factor"]
        self.error_thres = DEFAULT_CFG["error_thres"]
        self.prune_interval = DEFAULT_CFG["prune_interval"]
        self.optimization_steps = DEFAULT_CFG["optimization_steps"]
        self.background_color = torch.tensor(DEFAULT_CFG["background"], dtype=torch.float32).to(dev)
        if cfg is not None:                               # gaussian_map.py:40-52
            self.use_view_distribution = bool(_cfg_get(cfg, "use_view_distribution", True))
            self.scene_near, self.scene_far = (float(x) for x in _cfg_get(cfg, "bound"))
            self.sparse_ratio = _cfg_get(cfg, "sparse_ratio", 0.1)
            self.scale_factor = float(_cfg_get(cfg, "scale_factor"))
            self.error_thres = float(_cfg_get(cfg, "error_thres"))
            self.prune_interval = int(_cfg_get(cfg, "prune_interval"))
            self.optimization_steps = int(_cfg_get(cfg, "optimization_steps"))
            self.background_color = torch.tensor(list(_cfg_get(cfg, "background")), dtype=torch.float32).to(dev)
        # how the error-weighted older frames of a batch are drawn (mapping/utils.py:206-221): "device" = the same
        # distribution (successive sampling without replacement) from torch's device generator, no read-back of the
        # per-frame errors per iteration; "host" = np.random.choice on the host, the reference's own stream (the
        # reference seeds nothing - main.py, mapper.py - so no caller can depend on that stream; fixtures replayed
        # against a seeded capture set this to "host").  cfg.sampler.draw overrides the class default.
        self.frame_sampler = str(_cfg_get(_cfg_get(cfg, "sampler"), "draw", None) or self.FRAME_SAMPLER)
        # activation functions (gaussian_map.py:53-60), kept as attributes like the reference
        self.scaling_activation = torch.exp
        self.scaling_inverse_activation = torch.log
        self.opacity_activation = torch.sigmoid
        self.inverse_opacity_activation = lambda x: torch.log(x / (1 - x))
        self.rotation_activation = F.normalize
        self.optimizer = None

    # ------------------------------------------------------------------ trainer plumbing
    @property
    def num_gaussians(self) -> int:
        """Rows of the map, without waiting for the GPU (every state attribute / getter first settles a training call
        whose workspace check is pending - FusedMapTrainer.settle; the row count does not depend on it)."""
        return int(self._trainer.means.shape[0]) if self._trainer is not None else int(self._cold["means"].shape[0])

    def settle(self) -> None:
        """Wait for / look at what the last ``update()`` left pending (nothing to do for callers that read the map through
        its attributes: they settle themselves)."""
        if self._trainer is not None:
            self._trainer.settle()

    @property
    def training_data(self):
        return self._trainer.frames if self._trainer is not None else self._frames

    @training_data.setter
    def training_data(self, frames):
        if self._trainer is not None:
            self._trainer.frames = frames
        else:
            self._frames = frames

    @property
    def is_init(self):
        return self._trainer.is_init if self._trainer is not None else self._is_init

    @is_init.setter
    def is_init(self, v):
        if self._trainer is not None:
            self._trainer.is_init = bool(v)
        else:
            self._is_init = bool(v)

    def _trainer_cfg(self) -> dict:
        """The attributes (which a caller may have changed since construction, like the reference's) and cfg.optimizer /
        cfg.sampler as the trainer's dictionary."""
        opt, smp = _cfg_get(self.cfg, "optimizer"), _cfg_get(self.cfg, "sampler")
        d = DEFAULT_CFG["lrs"]
        lrs = dict(mean=float(_cfg_get(opt, "mean_lr", d["mean"])), scale=float(_cfg_get(opt, "scale_lr", d["scale"])),
                   rotation=float(_cfg_get(opt, "rotation_lr", d["rotation"])),
                   opacity=float(_cfg_get(opt, "opacity_lr", d["opacity"])),
                   harmonic=float(_cfg_get(opt, "harmonic_lr", d["harmonic"])))
        return dict(bound=(float(self.scene_near), float(self.scene_far)), scale_factor=float(self.scale_factor),
                    optimization_steps=int(self.optimization_steps), prune_interval=int(self.prune_interval),
                    error_thres=float(self.error_thres), use_view_distribution=bool(self.use_view_distribution),
                    background=self._background_host(),
                    batch_size=int(_cfg_get(smp, "batch_size", DEFAULT_CFG["batch_size"])),
                    active_size=int(_cfg_get(smp, "active_size", DEFAULT_CFG["active_size"])),
                    sampler_type=str(_cfg_get(smp, "sampler_type", "weighted")), sampler=self.frame_sampler, lrs=lrs)

    def _background_host(self) -> tuple:
        """``background_color`` as host floats - read back when the attribute is another tensor or has been written to since
        (a read-back per call is a wait for the GPU twice per keyframe)."""
        t = self.background_color
        key = (id(t), t._version) if torch.is_tensor(t) else None
        c = getattr(self, "_bg_cache", None)
        if c is None or key is None or c[0] != key or c[2] is not t:
            vals = tuple(float(x) for x in (t.reshape(-1).tolist() if torch.is_tensor(t) else t))
            c = self._bg_cache = (key, vals, t)
        return c[1]

    def _fused(self):
        """The trainer (made on first use; needs a GPU) with the current attribute values as its configuration."""
        if self.device.type != "cuda":
            raise RuntimeError("GaussianMap: training, growth and pruning run on the GPU (libags_raster.so); "
                               "there is no CPU fallback - construct the map with a cuda device")
        cfg = self._trainer_cfg()
        if self._trainer is None:
            from .fused_map_trainer import FusedMapTrainer
            raw = {k: self._cold[k].to(self.device).float() for k in _RAW + _VIEW}
            tr = FusedMapTrainer(raw, self._frames, cfg, process_group=self.process_group)
            tr.training_performance = self._cold["training_performance"].to(self.device).float()
            tr.is_init = self._is_init
            self._trainer = tr
        else:
            tr = self._trainer
            if tr.cfg.get("bound") != cfg["bound"] or tr.cfg.get("background") != cfg["background"]:
                tr._cams.clear(); tr._store = None; tr._uniform = None     # cameras carry the bounds / the background
                tr.background = self.background_color.to(self.device).float()
            tr.cfg.update(cfg)
        return tr

    # ------------------------------------------------------------------ the reference's methods
    def update(self, dataframe):
        """gaussian_map.py:62-64."""
        self.add_gaussians(dataframe)
        self.train()

    def train(self, steps: Optional[int] = None):
        """gaussian_map.py:66-130: ``optimization_steps`` (or ``steps``) iterations over batches of the newest + error-weighted
        older keyframes, then ``post_processing``."""
        tr = self._fused()
        tr.train(steps)                   # (its last statement is post_processing(), like the reference's)
        tr.is_init = True

    def post_processing(self):
        """gaussian_map.py:141-232."""
        self._fused().post_processing()

    def add_gaussians(self, dataframe):
        """gaussian_map.py:294-468; the frame's tensors are moved to the map's device like mapper.py:95 does."""
        return self._fused().add_gaussians(dataframe)

    def prune(self, prune_mask):
        """gaussian_map.py:234-246 (also removes surfels whose opacity has fallen under 0.1)."""
        deleted = self._fused().prune(prune_mask)
        print(f"delete {deleted} gaussians")

    def cal_mask(self, rgb_gt, depth_gt, pred):
        """gaussian_map.py:470-489: where a keyframe spawns new surfels.  (``add_gaussians`` evaluates the same rule inside
        ``ags_densify_candidates``; this is the reference's statement of it for callers that want the mask itself.)"""
        v, _, h, w = rgb_gt.shape
        device = rgb_gt.device
        if pred is None:
            return torch.ones(v, h, w, device=device).bool().reshape(-1)
        rgb, depth, opacity = pred["rgb"].to(device), pred["depth"].to(device), pred["opacity"].to(device)
        mask = torch.mean((rgb_gt - rgb) ** 2, dim=1) > self.error_thres
        mask = mask | (opacity < 0.5)
        mask = mask | ((depth_gt.squeeze(0) - depth) < -0.05 * depth_gt.squeeze(0))
        return mask.bool().reshape(-1)

    def get_sampler(self, training_data):
        """gaussian_map.py:248-257."""
        return make_frame_sampler(self._trainer_cfg(), training_data)

    def init_training(self):
        """gaussian_map.py:259-292 re-creates Adam for every train() call; the fused trainer does the same inside
        ``train`` (fresh moments and step counter per call).  Here for callers that expect ``self.optimizer`` to exist."""
        from .optimizer import FusedAdam
        tr = self._fused()
        lrs = tr.cfg["lrs"]
        params = [tr.means, tr.scales, tr.rotations, tr.opacities, tr.harmonics]
        self.optimizer = FusedAdam(params, [lrs["mean"], lrs["scale"], lrs["rotation"], lrs["opacity"], lrs["harmonic"]], eps=1e-15)

    def track_performance(self, rgb_loss, depth_loss, frame_ids):
        """gaussian_map.py:132-139."""
        errs = torch.mean(rgb_loss, dim=[1, 2, 3]).detach() + torch.mean(depth_loss, dim=[1, 2, 3]).detach()
        self.training_performance[torch.as_tensor(frame_ids, device=errs.device)] = errs

    # ------------------------------------------------------------------ checkpoints (gaussian_map.py:491-527)
    def save(self, save_path, index="final"):
        from .map_io import compact        # (the arrays are views of larger buffers: torch.save would write those whole)
        map_state = {
            "means": compact(self._means), "scales": compact(self._scales), "harmonics": compact(self._harmonics),
            "opacities": compact(self._opacities), "rotations": compact(self._rotations),
            "view_scores": compact(self.view_scores), "view_supports": compact(self.view_supports),
            "view_means": compact(self.view_means), "near": self.scene_near, "far": self.scene_far,
            "use_view_direction": self.use_view_distribution, "background_color": self.background_color,
            "scale_factor": self.scale_factor,
        }
        torch.save(map_state, f"{save_path}/map_{index}.th")

    def load(self, model_path):
        map_state = torch.load(model_path, map_location=self.device)
        n = map_state["means"].shape[0]
        for k in _RAW + _VIEW:
            setattr(self, "_" + k if k in _RAW else k, map_state[k].to(self.device))
        self._harmonics = self._harmonics.reshape(n, 1, 3)
        self.scene_near = map_state["near"]
        self.scene_far = map_state["far"]
        self.background_color = torch.as_tensor(map_state["background_color"], dtype=torch.float32).to(self.device)
        self.scale_factor = map_state["scale_factor"]
        self.is_init = True

    # ------------------------------------------------------------------ activations (gaussian_map.py:529-590)
    # LIFETIME (differs from the reference, whose growth / prune make fresh tensors): ``_means``, ``_harmonics``, ... and what
    # ``get_means`` / ``get_harmonics`` / ``get_params`` return are VIEWS of the leading rows of the map's buffers
    # (densify.MapArena) - valid until the next ``update()`` / ``add_gaussians()`` / ``prune()``; growth appends in place,
    # prune compacts into the second buffer set and swaps, so a view kept across a prune shows other rows.  A caller that
    # keeps map data across keyframes (a GUI packet, a voxel map) takes ``.clone()``; the activated getters
    # (``get_scales``, ``get_opacities``, ``get_rotations``, ``get_confidences``, ``get_normals``) return fresh tensors.
    @property
    def get_means(self):
        return self._means

    @property
    def get_rotations(self):
        return self.rotation_activation(self._rotations)

    @property
    def get_scales(self):
        return torch.clamp(self.scale_factor * self.scaling_activation(self._scales), min=0, max=0.05)

    @property
    def get_opacities(self):
        return self.opacity_activation(self._opacities)

    @property
    def get_harmonics(self):
        return self._harmonics

    @property
    def get_confidences(self):
        if self._trainer is not None and self._means.is_cuda and self._means.shape[0] > 0:
            self._trainer.cfg["use_view_distribution"] = bool(self.use_view_distribution)
            return self._trainer.confidences()                     # one launch (ags_confidences)
        if self.use_view_distribution:
            view_var = self.view_means.norm(dim=-1)
            view_var = torch.where(torch.isnan(view_var), torch.ones_like(view_var), view_var)
            return torch.clamp(torch.exp(1 - view_var) * self.view_scores, min=0, max=1)
        return torch.clamp(1 - 1 / torch.exp(self.view_supports), min=0, max=1)

    @property
    def get_normals(self):
        r, x, y, z = self.get_rotations.unbind(-1)
        # third column of quaternion_to_matrix (operations.py:261-278: unit quaternion assumed) ...
        col = torch.stack([2 * (x * z + r * y), 2 * (y * z - r * x), 1 - 2 * (x * x + y * y)], -1)
        return self.rotation_activation(col)

    def get_attr(self):
        return (self.get_means, self.get_harmonics, self.get_opacities, self.get_confidences, self.get_scales,
                self.get_rotations)

    def get_params(self):
        return (self._means, self._harmonics, self._opacities, self._scales, self._rotations)
