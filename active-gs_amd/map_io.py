"""Map checkpoints in the reference's ``.th`` format: ``GaussianMap.save`` / ``load``
(/root/reference/mapping/gaussian_map.py:491-527) write/read ``torch.save`` dictionaries with
the keys below, RAW (pre-activation) tensors.  Files written here load in the reference's
``eval.py`` / ``mesh_generation.py`` / ``visualize.py`` and vice versa."""
from __future__ import annotations

import os

import torch

MAP_KEYS = ("means", "scales", "harmonics", "opacities", "rotations", "view_scores", "view_supports", "view_means",
            "near", "far", "use_view_direction", "background_color", "scale_factor")


def compact(t: torch.Tensor) -> torch.Tensor:
    """``t.detach()`` - as a tensor that owns exactly its elements: ``torch.save`` writes a tensor's WHOLE storage, and the
    map's arrays are the leading rows of buffers with room behind them (densify.MapArena)."""
    t = t.detach()
    if t.untyped_storage().nbytes() > t.numel() * t.element_size():
        t = t.clone()
    return t


def map_state(trainer) -> dict:
    """State dict of a ``GaussianMapTrainer`` / ``FusedMapTrainer`` in the reference's schema."""
    getattr(trainer, "settle", lambda: False)()      # a train() call whose workspace check is still pending (FusedMapTrainer)
    near, far = trainer.cfg["bound"]
    return {
        "means": compact(trainer.means), "scales": compact(trainer.scales), "harmonics": compact(trainer.harmonics),
        "opacities": compact(trainer.opacities), "rotations": compact(trainer.rotations),
        "view_scores": compact(trainer.view_scores), "view_supports": compact(trainer.view_supports),
        "view_means": compact(trainer.view_means), "near": near, "far": far,
        "use_view_direction": trainer.cfg["use_view_distribution"], "background_color": trainer.background,
        "scale_factor": trainer.cfg["scale_factor"],
    }


def save_map(trainer, save_path: str, index="final") -> str:
    path = os.path.join(save_path, f"map_{index}.th")
    torch.save(map_state(trainer), path)
    return path


def load_map(model_path: str, device="cpu"):
    """-> (raw parameter dict for the trainers, cfg overrides)."""
    st = torch.load(model_path, map_location=device)
    missing = [k for k in MAP_KEYS if k not in st]
    if missing:
        raise KeyError(f"{model_path} is not an ActiveGS map checkpoint: missing {missing}")
    raw = {k: st[k] for k in ("means", "scales", "harmonics", "opacities", "rotations", "view_scores", "view_supports",
                              "view_means")}
    bg = torch.as_tensor(st["background_color"], dtype=torch.float32).tolist()
    cfg = dict(bound=(st["near"], st["far"]), scale_factor=st["scale_factor"], background=tuple(bg),
               use_view_distribution=bool(st["use_view_direction"]))
    return raw, cfg
