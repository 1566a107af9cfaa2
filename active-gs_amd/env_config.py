"""The AGS_* variables of INTEGRATION.md, mapped onto the package's knobs by an EXPLICIT call.

The library reads no environment variable and neither does any other module of this package: kernel selection travels
as ``AgsTuning`` with the workspace, everything else is a constructor argument or a class attribute.  A launcher that
wants the documented variables honoured - bench.py, the tests' conftest, the experiment scripts under
profiles/experiments/ - calls ``apply_env(os.environ)`` once, before it makes its first workspace or trainer.  The
function is pure in its argument (any mapping) and returns what it changed."""
from . import _lib


def apply_env(env) -> dict:
    changed = {}
    if env.get("AGS_LIB_PATH"):
        _lib.lib_path_override = changed["lib_path"] = env["AGS_LIB_PATH"]
    tuning_keys = ("AGS_BWD_REDUCE", "AGS_BWD_BF16", "AGS_BWD_MFMA", "AGS_RENDER_SLOTS", "AGS_PRE_CULL_MIN_N",
                   "AGS_TSORT_NO_WAVE", "AGS_BUCKET_NO_SCAN", "AGS_VIEW_GROUP")
    if any(env.get(k) is not None for k in tuning_keys):
        _lib.set_default_tuning(_lib.tuning_from_env(env), cull_pinned=env.get("AGS_PRE_CULL_MIN_N") is not None)
        changed["tuning"] = {k: env[k] for k in tuning_keys if env.get(k) is not None}
    trainer_keys = {"AGS_VIEW_STREAMS": ("VIEW_STREAMS", int), "AGS_DENSE_CHUNKS": ("DENSE_CHUNKS", int),
                    "AGS_DP_FORCE": ("DP_FORCE", lambda v: v == "1"),
                    "AGS_DP_GRAPH_COLLECTIVES": ("GRAPH_COLLECTIVES", lambda v: v != "0"),
                    "AGS_CULL_ADAPT": ("CULL_ADAPT", lambda v: v != "0")}
    if any(env.get(k) is not None for k in trainer_keys):
        from .trainer import SurfelTrainer
        for k, (attr, conv) in trainer_keys.items():
            if env.get(k) is not None:
                setattr(SurfelTrainer, attr, conv(env[k]))
                changed[attr] = getattr(SurfelTrainer, attr)
    if env.get("AGS_MAPPER_DEFER_SETTLE") is not None:
        from .fused_map_trainer import FusedMapTrainer
        FusedMapTrainer.DEFER_SETTLE = changed["DEFER_SETTLE"] = env["AGS_MAPPER_DEFER_SETTLE"] != "0"
    if env.get("AGS_FUSE_LOSS_STAGE1") is not None:
        from .fused_map_trainer import FusedMapTrainer
        FusedMapTrainer.FUSE_LOSS_STAGE1 = changed["FUSE_LOSS_STAGE1"] = env["AGS_FUSE_LOSS_STAGE1"] != "0"
    if env.get("AGS_FRAME_SAMPLER") is not None:
        from .gaussian_map import GaussianMap
        GaussianMap.FRAME_SAMPLER = changed["FRAME_SAMPLER"] = env["AGS_FRAME_SAMPLER"]
    if env.get("AGS_DROPIN_STATUS") is not None:
        from . import rasterizer
        rasterizer.DROPIN_STATUS = changed["DROPIN_STATUS"] = env["AGS_DROPIN_STATUS"]
    return changed
