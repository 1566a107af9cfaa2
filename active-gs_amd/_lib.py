"""ctypes binding of libags_raster.so (include/ags_raster.h).

There is no CPU fallback: if the library is missing or cannot be loaded every entry
point raises.  torch is imported first on purpose so the library binds to the HIP
runtime torch already loaded (same SONAME, one runtime per process)."""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (must precede the CDLL load, see module docstring)

from . import build as _build

c_f32p = C.c_void_p


class AgsCamera(C.Structure):
    _fields_ = [("image_height", C.c_int32), ("image_width", C.c_int32), ("tanfovx", C.c_float),
                ("tanfovy", C.c_float), ("scale_modifier", C.c_float), ("weight_thres", C.c_float),
                ("normalize_depth", C.c_int32), ("perpix_depth", C.c_int32), ("want_stats", C.c_int32),
                ("front_only", C.c_int32), ("viewmatrix", c_f32p), ("projmatrix", c_f32p), ("bg", c_f32p),
                ("render_mask", c_f32p), ("config", c_f32p)]


class AgsGaussians(C.Structure):
    _fields_ = [("n", C.c_int32), ("means3D", c_f32p), ("scales", c_f32p), ("rotations", c_f32p),
                ("opacities", c_f32p), ("colors", c_f32p), ("confidences", c_f32p), ("raw_params", C.c_int32),
                ("scale_factor", C.c_float), ("max_scale", C.c_float)]


class AgsImages(C.Structure):
    _fields_ = [("rgb", c_f32p), ("normal", c_f32p), ("depth", c_f32p), ("opacity", c_f32p), ("confidence", c_f32p)]


class AgsRowSet(C.Structure):
    _fields_ = [("member", C.c_void_p), ("rows", C.c_void_p), ("count", C.c_void_p)]


class AgsPerGaussian(C.Structure):
    _fields_ = [("importance", c_f32p), ("count", C.c_void_p), ("radii", C.c_void_p), ("touched", AgsRowSet)]


class AgsImageGrads(C.Structure):
    _fields_ = [("d_rgb", c_f32p), ("d_normal", c_f32p), ("d_depth", c_f32p), ("d_opacity", c_f32p),
                ("d_confidence", c_f32p)]


class AgsGaussianGrads(C.Structure):
    _fields_ = [("d_means3D", c_f32p), ("d_scales", c_f32p), ("d_rotations", c_f32p), ("d_opacities", c_f32p),
                ("d_colors", c_f32p), ("d_means2D", c_f32p), ("accumulate", C.c_int32), ("adam_clock", C.c_void_p),
                ("adam_lr", C.c_float * 5), ("adam_beta1", C.c_float), ("adam_beta2", C.c_float),
                ("touched", AgsRowSet), ("fused_adam", C.c_void_p), ("adam_eps", C.c_float),
                ("pack_segment", c_f32p), ("pack_capacity", C.c_int32), ("defer_rows", C.c_int32),
                ("row_begin", C.c_int32), ("row_end", C.c_int32)]



class AgsTuning(C.Structure):
    _fields_ = [("bwd_reduce", C.c_int32), ("render_slots", C.c_int32), ("cull_first_min_n", C.c_int32),
                ("tile_sort_no_wave", C.c_int32), ("bucket_no_scan", C.c_int32), ("view_group", C.c_int32),
                ("reserved", C.c_int32 * 2)]


BWD_F32, BWD_BF16_SPLIT, BWD_VALU, BWD_BF16X3 = 0, 1, 2, 3
_BWD_NAMES = {"f32": BWD_F32, "bf16": BWD_BF16_SPLIT, "bf16_split": BWD_BF16_SPLIT, "valu": BWD_VALU, "bf16x3": BWD_BF16X3}


def tuning_from_env(env) -> AgsTuning:
    """Neither the library nor this package reads an environment variable (AgsTuning travels with the workspace).  This
    is a pure function of the mapping it is given: a launcher that wants the AGS_* variables INTEGRATION.md lists
    honoured (bench.py, the tests' conftest, experiment scripts) passes ``os.environ`` through ``env_config.apply_env``:
      AGS_BWD_REDUCE=f32|bf16|bf16x3|valu  (AGS_BWD_BF16=1 = bf16, AGS_BWD_MFMA=0 = valu)   blend backward's per-surfel sums
      AGS_RENDER_SLOTS=1|2|4, AGS_PRE_CULL_MIN_N=<rows> (0: always), AGS_VIEW_GROUP=<views per workgroup of a batched forward>,
      AGS_TSORT_NO_WAVE, AGS_BUCKET_NO_SCAN"""
    t = AgsTuning()
    mode = env.get("AGS_BWD_REDUCE")
    if mode is not None:
        if mode not in _BWD_NAMES:
            raise ValueError(f"AGS_BWD_REDUCE={mode!r}: expected one of {sorted(_BWD_NAMES)}")
        t.bwd_reduce = _BWD_NAMES[mode]
    elif env.get("AGS_BWD_BF16", "0") not in ("", "0"):
        t.bwd_reduce = BWD_BF16_SPLIT
    elif env.get("AGS_BWD_MFMA") == "0":
        t.bwd_reduce = BWD_VALU
    if env.get("AGS_RENDER_SLOTS") in ("1", "2", "4"):
        t.render_slots = int(env["AGS_RENDER_SLOTS"])
    if env.get("AGS_PRE_CULL_MIN_N") is not None:
        v = int(env["AGS_PRE_CULL_MIN_N"])
        t.cull_first_min_n = 1 if v <= 0 else min(v, 0x7FFFFFFF)
    if env.get("AGS_VIEW_GROUP") is not None:
        t.view_group = int(env["AGS_VIEW_GROUP"])
    t.tile_sort_no_wave = int(env.get("AGS_TSORT_NO_WAVE") is not None)
    t.bucket_no_scan = int(env.get("AGS_BUCKET_NO_SCAN") is not None)
    return t


_default_tuning = None
cull_choice_pinned = False     # set_default_tuning(..., cull_pinned=True): trainers leave the per-Gaussian kernel choice alone


def default_tuning() -> AgsTuning:
    """The process's default selection: the library's own defaults (all zero) unless a launcher has set another
    (``set_default_tuning``); kept alive for the structs that point at it."""
    global _default_tuning
    if _default_tuning is None:
        _default_tuning = AgsTuning()
    return _default_tuning


def set_default_tuning(t: AgsTuning, cull_pinned: bool = False) -> None:
    """Replace the process's default selection (before the first workspace is made: structs made earlier keep the old one)."""
    global _default_tuning, cull_choice_pinned
    _default_tuning, cull_choice_pinned = t, bool(cull_pinned)


def make_tuning(bwd_reduce=None, render_slots=None, cull_first_min_n=None, view_group=None) -> AgsTuning:
    """A copy of the default selection with some fields replaced (bwd_reduce: "f32" | "bf16x3" | "bf16" | "valu" or AGS_BWD_*;
    view_group: k > 1 = a batched forward's per-Gaussian stage loads a row once for k consecutive views; 0 / 1 = off)."""
    d = default_tuning()
    t = AgsTuning(d.bwd_reduce, d.render_slots, d.cull_first_min_n, d.tile_sort_no_wave, d.bucket_no_scan, d.view_group)
    if view_group is not None:
        t.view_group = int(view_group)
    if bwd_reduce is not None:
        t.bwd_reduce = _BWD_NAMES[bwd_reduce] if isinstance(bwd_reduce, str) else int(bwd_reduce)
    if render_slots is not None:
        t.render_slots = int(render_slots)
    if cull_first_min_n is not None:
        t.cull_first_min_n = int(cull_first_min_n)
    return t


class AgsWorkspace(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("bytes", C.c_size_t), ("max_instances", C.c_int64),
                ("binning_mode", C.c_int32), ("tuning", C.POINTER(AgsTuning)), ("early_status_host", C.c_void_p),
                ("early_status_event", C.c_void_p)]


def workspace(ptr_, nbytes, max_instances, binning_mode, tuning: "AgsTuning | None" = None) -> AgsWorkspace:
    """AgsWorkspace with the caller's (or the process's default) kernel selection attached."""
    t = tuning if tuning is not None else default_tuning()
    ws = AgsWorkspace(ptr_, nbytes, int(max_instances), int(binning_mode), C.pointer(t))
    ws._tuning_ref = t           # (the pointer above does not own the struct)
    return ws


class AgsViewRef(C.Structure):
    _fields_ = [("cam", C.POINTER(AgsCamera)), ("radii", C.c_void_p), ("ws", C.POINTER(AgsWorkspace))]


class AgsStatus(C.Structure):
    _fields_ = [("num_instances", C.c_uint32), ("num_sorted", C.c_uint32), ("overflow", C.c_uint32),
                ("num_visible", C.c_uint32), ("peak_instances", C.c_uint32), ("overflow_passes", C.c_uint32),
                ("max_tile_instances", C.c_uint32), ("needed_instances", C.c_uint32), ("reserved0", C.c_uint32),
                ("early_tile_need", C.c_uint32), ("reserved", C.c_uint32 * 6)]


class AgsAdamTensors(C.Structure):
    _fields_ = [("param", c_f32p * 5), ("grad", c_f32p * 5), ("exp_avg", c_f32p * 5), ("exp_avg_sq", c_f32p * 5),
                ("numel", C.c_int64 * 5), ("lr", C.c_float * 5), ("touched", AgsRowSet), ("zero_grad", C.c_int32),
                ("state_rows", c_f32p)]


class AgsLossConfig(C.Structure):
    _fields_ = [("image_height", C.c_int32), ("image_width", C.c_int32), ("fov_x", C.c_float), ("fov_y", C.c_float),
                ("batch_total", C.c_int32), ("w_rgb", C.c_float), ("w_depth", C.c_float), ("w_cons", C.c_float),
                ("w_tv", C.c_float), ("sigma", C.c_float), ("accum_stride", C.c_int32), ("num_views", C.c_int32),
                ("gt_frame_index", C.c_void_p)]


class AgsLossEpilogue(C.Structure):
    _fields_ = [("cfg", C.POINTER(AgsLossConfig)), ("gt_rgb", C.c_void_p), ("gt_depth", C.c_void_p), ("n_img", C.c_void_p),
                ("d_rgb", C.c_void_p), ("d_depth", C.c_void_p), ("msum", C.c_void_p), ("accum", C.c_void_p)]


class AgsNextIteration(C.Structure):
    _fields_ = [("uniforms", C.c_void_p), ("n_weights", C.c_int32), ("k", C.c_int32), ("first_random", C.c_int32),
                ("views", C.c_int32), ("all_view", C.c_void_p), ("all_proj", C.c_void_p), ("dst_view", C.c_void_p),
                ("dst_proj", C.c_void_p), ("msum", C.c_void_p)]


class AgsMapArrays(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("means", "scales", "rotations", "opacities", "harmonics", "view_scores",
                                          "view_supports", "view_means")]


class AgsActivation(C.Structure):
    _fields_ = [("n", C.c_int32), ("scale_factor", C.c_float), ("max_scale", C.c_float), ("raw_scales", c_f32p),
                ("raw_rotations", c_f32p), ("raw_opacities", c_f32p)]


class AgsKeyframe(C.Structure):
    _fields_ = [("image_height", C.c_int32), ("image_width", C.c_int32), ("rgb", c_f32p), ("depth", c_f32p),
                ("intrinsic_inv", c_f32p), ("extrinsic", c_f32p)]


class AgsDensifyPred(C.Structure):
    _fields_ = [("rgb", c_f32p), ("depth", c_f32p), ("opacity", c_f32p)]


class AgsCandidates(C.Structure):
    _fields_ = [("means", c_f32p), ("rotations", c_f32p), ("harmonics", c_f32p), ("select", C.c_void_p)]


EXPORTS = ["ags_workspace_bytes", "ags_workspace_region", "ags_workspace_init", "ags_workspace_init_batch", "ags_workspace_discard_pass", "ags_forward", "ags_forward_batch", "ags_forward_batch_loss",
           "ags_forward_batch_workspace_bytes", "ags_backward", "ags_backward_batch", "ags_backward_rows", "ags_backward_fused_next", "ags_forward_resume", "ags_read_status", "ags_read_status_async", "ags_adam_step",
           "ags_adam_step_device", "ags_rows_segment_floats", "ags_rows_pack", "ags_rows_unpack", "ags_rows_index", "ags_adam_step_gathered", "ags_activate", "ags_activate_backward", "ags_loss_stage1", "ags_loss_stage2", "ags_stage_frames", "ags_loss_finish", "ags_loss_finish_next", "ags_zero_many", "ags_weighted_topk", "ags_facade_post", "ags_facade_post_batch", "ags_facade_post_backward", "ags_smooth_depth", "ags_densify_candidates",
           "ags_voxel_select_bytes", "ags_voxel_select", "ags_prune_keep", "ags_view_stats_update", "ags_confidences", "ags_compact_plan_bytes", "ags_compact_plan",
           "ags_compact_rows", "ags_map_append", "ags_map_compact", "ags_profile_enable", "ags_profile_read",
           "ags_error_string", "ags_version"]

_lib = None


lib_path_override = None       # an alternative build of the same library (kernel experiments; env_config: AGS_LIB_PATH)


def library_path() -> str:
    return lib_path_override or _build.LIB


def load() -> C.CDLL:
    """Load (never build) the shared library; raise if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the rasterizer.")
    lib = C.CDLL(path)
    lib.ags_workspace_bytes.restype = C.c_size_t
    lib.ags_workspace_bytes.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int64]
    lib.ags_workspace_region.restype = C.c_int
    lib.ags_workspace_region.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_int32,
                                         C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    lib.ags_workspace_init.restype = C.c_int
    lib.ags_workspace_init.argtypes = [C.POINTER(AgsWorkspace), C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    lib.ags_workspace_init_batch.restype = C.c_int
    lib.ags_workspace_init_batch.argtypes = [C.POINTER(AgsWorkspace), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    lib.ags_workspace_discard_pass.restype = C.c_int
    lib.ags_workspace_discard_pass.argtypes = [C.POINTER(AgsWorkspace), C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    lib.ags_forward.restype = C.c_int
    lib.ags_forward.argtypes = [C.POINTER(AgsCamera), C.POINTER(AgsGaussians), C.POINTER(AgsImages),
                                C.POINTER(AgsPerGaussian), C.POINTER(AgsWorkspace), C.c_void_p]
    lib.ags_forward_batch.restype = C.c_int
    lib.ags_forward_batch.argtypes = [C.POINTER(AgsCamera), C.c_int32, C.POINTER(AgsGaussians), C.POINTER(AgsImages),
                                      C.POINTER(AgsPerGaussian), C.POINTER(AgsWorkspace), C.c_void_p]
    lib.ags_forward_batch_loss.restype = C.c_int
    lib.ags_forward_batch_loss.argtypes = [C.POINTER(AgsCamera), C.c_int32, C.POINTER(AgsGaussians), C.POINTER(AgsImages),
                                           C.POINTER(AgsPerGaussian), C.POINTER(AgsWorkspace), C.POINTER(AgsLossEpilogue),
                                           C.c_void_p]
    lib.ags_forward_batch_workspace_bytes.restype = C.c_size_t
    lib.ags_forward_batch_workspace_bytes.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int64]
    lib.ags_backward.restype = C.c_int
    lib.ags_backward.argtypes = [C.POINTER(AgsCamera), C.POINTER(AgsGaussians), C.POINTER(AgsImages),
                                 C.POINTER(AgsPerGaussian), C.POINTER(AgsImageGrads), C.POINTER(AgsGaussianGrads),
                                 C.POINTER(AgsWorkspace), C.c_void_p]
    lib.ags_backward_rows.restype = C.c_int
    lib.ags_backward_rows.argtypes = [C.POINTER(AgsViewRef), C.c_int32, C.POINTER(AgsGaussians), C.POINTER(AgsGaussianGrads),
                                      C.c_void_p]
    lib.ags_backward_fused_next.restype = C.c_int
    lib.ags_backward_fused_next.argtypes = [C.POINTER(AgsCamera), C.POINTER(AgsGaussians), C.POINTER(AgsImages),
                                            C.POINTER(AgsPerGaussian), C.POINTER(AgsImageGrads), C.POINTER(AgsGaussianGrads),
                                            C.POINTER(AgsWorkspace), C.POINTER(AgsCamera), C.POINTER(AgsPerGaussian),
                                            C.POINTER(AgsWorkspace), C.c_int32, C.c_void_p]
    lib.ags_forward_resume.restype = C.c_int
    lib.ags_forward_resume.argtypes = [C.POINTER(AgsCamera), C.POINTER(AgsGaussians), C.POINTER(AgsImages),
                                       C.POINTER(AgsPerGaussian), C.POINTER(AgsWorkspace), C.c_void_p]
    lib.ags_backward_batch.restype = C.c_int
    lib.ags_backward_batch.argtypes = [C.POINTER(AgsCamera), C.c_int32, C.POINTER(AgsGaussians), C.POINTER(AgsImages),
                                       C.POINTER(AgsPerGaussian), C.POINTER(AgsImageGrads), C.POINTER(AgsGaussianGrads),
                                       C.POINTER(AgsWorkspace), C.c_void_p]
    lib.ags_read_status.restype = C.c_int
    lib.ags_read_status.argtypes = [C.POINTER(AgsWorkspace), C.POINTER(AgsStatus), C.c_void_p]
    lib.ags_read_status_async.restype = C.c_int
    lib.ags_read_status_async.argtypes = [C.POINTER(AgsWorkspace), C.c_void_p, C.c_void_p]
    lib.ags_adam_step.restype = C.c_int
    lib.ags_adam_step.argtypes = [C.POINTER(AgsAdamTensors), C.c_float, C.c_float, C.c_float, C.c_int32, C.c_void_p]
    lib.ags_adam_step_device.restype = C.c_int
    lib.ags_adam_step_device.argtypes = [C.POINTER(AgsAdamTensors), C.c_float, C.c_float, C.c_float, C.c_void_p,
                                         C.c_int32, C.c_void_p]
    lib.ags_rows_segment_floats.restype = C.c_size_t
    lib.ags_rows_segment_floats.argtypes = [C.c_int32]
    lib.ags_rows_pack.restype = C.c_int
    lib.ags_rows_pack.argtypes = [C.POINTER(AgsRowSet), C.POINTER(C.c_void_p * 5), C.c_int32, C.c_void_p, C.c_void_p]
    lib.ags_rows_unpack.restype = C.c_int
    lib.ags_rows_unpack.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p * 5), C.POINTER(AgsRowSet), C.c_void_p]
    lib.ags_rows_index.restype = C.c_int
    lib.ags_rows_index.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.POINTER(AgsRowSet), C.c_void_p]
    lib.ags_adam_step_gathered.restype = C.c_int
    lib.ags_adam_step_gathered.argtypes = [C.POINTER(AgsAdamTensors), C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_float,
                                           C.c_float, C.c_float, C.c_void_p, C.c_int32, C.c_void_p]
    lib.ags_activate.restype = C.c_int
    lib.ags_activate.argtypes = [C.POINTER(AgsActivation), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ags_activate_backward.restype = C.c_int
    lib.ags_activate_backward.argtypes = [C.POINTER(AgsActivation), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ags_loss_stage1.restype = C.c_int
    lib.ags_loss_stage1.argtypes = [C.POINTER(AgsLossConfig), C.POINTER(AgsImages)] + [C.c_void_p] * 7 + [
        C.c_int32, C.c_int32, C.c_void_p]
    lib.ags_loss_stage2.restype = C.c_int
    lib.ags_loss_stage2.argtypes = [C.POINTER(AgsLossConfig), C.POINTER(AgsImages)] + [C.c_void_p] * 6 + [C.c_void_p]
    lib.ags_facade_post.restype = C.c_int
    lib.ags_facade_post.argtypes = [C.c_int32, C.c_int32, C.c_float, C.c_float] + [C.c_void_p] * 5 + [C.c_void_p]
    lib.ags_facade_post_batch.restype = C.c_int
    lib.ags_facade_post_batch.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float] + [C.c_void_p] * 5 + [C.c_void_p]
    lib.ags_facade_post_backward.restype = C.c_int
    lib.ags_facade_post_backward.argtypes = [C.c_int32, C.c_int32, C.c_float, C.c_float] + [C.c_void_p] * 7 + [C.c_void_p]
    lib.ags_weighted_topk.restype = C.c_int
    lib.ags_weighted_topk.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    lib.ags_stage_frames.restype = C.c_int
    lib.ags_stage_frames.argtypes = [C.c_int32, C.c_int32, C.c_int32] + [C.c_void_p] * 10 + [C.c_void_p]
    lib.ags_loss_finish.restype = C.c_int
    lib.ags_loss_finish.argtypes = [C.POINTER(AgsLossConfig), C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p]
    lib.ags_zero_many.restype = C.c_int
    lib.ags_zero_many.argtypes = [C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_void_p]
    lib.ags_loss_finish_next.restype = C.c_int
    lib.ags_loss_finish_next.argtypes = [C.POINTER(AgsLossConfig), C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.POINTER(AgsNextIteration), C.c_void_p]
    lib.ags_smooth_depth.restype = C.c_int
    lib.ags_smooth_depth.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_float,
                                     C.c_void_p]
    lib.ags_densify_candidates.restype = C.c_int
    lib.ags_densify_candidates.argtypes = [C.POINTER(AgsKeyframe), C.c_void_p, C.POINTER(AgsDensifyPred), C.c_float,
                                           C.POINTER(AgsCandidates), C.c_void_p]
    lib.ags_voxel_select_bytes.restype = C.c_size_t
    lib.ags_voxel_select_bytes.argtypes = [C.c_int32]
    lib.ags_voxel_select.restype = C.c_int
    lib.ags_voxel_select.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.ags_view_stats_update.restype = C.c_int
    lib.ags_view_stats_update.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int32,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ags_confidences.restype = C.c_int
    lib.ags_confidences.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    lib.ags_prune_keep.restype = C.c_int
    lib.ags_prune_keep.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]
    lib.ags_compact_plan_bytes.restype = C.c_size_t
    lib.ags_compact_plan_bytes.argtypes = [C.c_int32]
    lib.ags_compact_plan.restype = C.c_int
    lib.ags_compact_plan.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.ags_compact_rows.restype = C.c_int
    lib.ags_compact_rows.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ags_map_append.restype = C.c_int
    lib.ags_map_append.argtypes = [C.c_int32, C.c_void_p, C.POINTER(AgsCandidates), C.c_float, C.POINTER(AgsMapArrays), C.c_void_p]
    lib.ags_map_compact.restype = C.c_int
    lib.ags_map_compact.argtypes = [C.c_int32, C.c_void_p, C.POINTER(AgsMapArrays), C.POINTER(AgsMapArrays), C.c_void_p]
    lib.ags_profile_enable.restype = C.c_int
    lib.ags_profile_enable.argtypes = [C.c_int32]
    lib.ags_profile_read.restype = C.c_int
    lib.ags_profile_read.argtypes = [C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int32)]
    lib.ags_error_string.restype = C.c_char_p
    lib.ags_error_string.argtypes = [C.c_int]
    lib.ags_version.restype = C.c_int
    _lib = lib
    return lib


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def current_stream() -> int:
    """Raw handle of torch's current stream on the current device.  ``torch.cuda.current_stream().cuda_stream`` builds a
    Stream object per call (~9 us: a tenth of the mapper loop's wall time at ~10 launches per iteration); the raw getter
    is a plain C call."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def check(code: int, what: str) -> None:
    if code != 0:
        raise RuntimeError(f"{what} failed: {load().ags_error_string(code).decode()} ({code})")


def zero_many(tensors) -> None:
    """Zero the given (contiguous, 16-byte aligned) device tensors with ONE launch per sixteen of them (``ags_zero_many``)."""
    ts = [t for t in tensors if t is not None and t.numel() > 0]
    lib = load()
    for a in range(0, len(ts), 16):
        part = ts[a:a + 16]
        for t in part:
            if not t.is_contiguous() or not t.is_cuda or (t.data_ptr() & 15) or (t.numel() * t.element_size()) & 3:
                raise ValueError("zero_many: contiguous GPU tensors at 16-byte aligned addresses, sizes in whole words")
        regions = (C.c_void_p * len(part))(*[t.data_ptr() for t in part])
        sizes = (C.c_size_t * len(part))(*[t.numel() * t.element_size() for t in part])
        check(lib.ags_zero_many(len(part), regions, sizes, current_stream()), "ags_zero_many")


def ptr(t) -> int | None:
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()
