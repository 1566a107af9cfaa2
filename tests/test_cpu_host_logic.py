"""CPU suite, part 2: host logic against fixtures captured from the REFERENCE's own Python
(tests/golden/make_golden.py drove /root/reference's GaussianRenderer / GaussianMap.train()),
and the C ABI library's load/export/argument checks (no GPU compute)."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

import _oracle_module

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def test_camera_conventions_match_reference_renderer():
    from active_gs_amd.camera import camera_matrices
    for c in torch.load(os.path.join(GOLD, "camera.pt")):
        cm = camera_matrices(c["c2w"], c["K"], c["near"], c["far"])
        assert torch.allclose(2 * torch.atan(cm["tanfov"]), c["fovs"], atol=1e-6)
        assert torch.allclose(cm["viewmatrix"], c["view_matrices"], atol=1e-6)
        assert torch.allclose(cm["projmatrix"], c["projection_matrices"], atol=1e-5)
        assert torch.equal(cm["campos"], c["cam_pos"])
        # row-vector convention: translation in the last row, w = view-space z
        assert torch.allclose(cm["viewmatrix"][:, :3, 3], torch.zeros(3, 3), atol=1e-6)   # (an LU inverse: not exact zeros on every host)
        assert torch.allclose(cm["projmatrix"][:, :, 3], cm["viewmatrix"][:, :, 2], atol=1e-6)


def test_facade_mirror_matches_reference_renderer():
    from active_gs_amd.facade import SurfelRenderer
    d = torch.load(os.path.join(GOLD, "facade.pt"))
    leaves = {k: v.clone().requires_grad_(True) for k, v in d["attr"].items()}
    attr = (leaves["means"], leaves["harmonics"], leaves["opacities"], d["confidences"], leaves["scales"],
            leaves["rotations"])
    r = SurfelRenderer(d["c2w"], d["K"], attr, d["bg"], (d["near"], d["far"]), (d["h"], d["w"]), "cpu",
                       rasterizer_module=_oracle_module)
    outs = r.render_view_all(require_grad=True)
    assert len(outs) == 9
    for o, ref in zip(outs, d["outputs"]):
        assert o.shape == ref.shape and o.dtype == ref.dtype
        if o.dtype.is_floating_point:
            assert torch.allclose(o.detach(), ref, rtol=1e-4, atol=1e-5)
        else:
            assert torch.equal(o, ref)
    scalar = sum((o * g).sum() for o, g in zip(outs[:6], d["weights"]))
    assert torch.allclose(scalar.detach(), d["scalar"], rtol=1e-4)
    scalar.backward()
    for k, ref in d["grads"].items():
        rel = (leaves[k].grad - ref).abs().sum() / ref.abs().sum().clamp_min(1e-12)
        assert rel < 1e-3, (k, float(rel))


def _train_from_fixture(d, **kw):
    from active_gs_amd.map_trainer import GaussianMapTrainer
    cfg = d["cfg"]
    mine = dict(bound=tuple(cfg["bound"]), scale_factor=cfg["scale_factor"], optimization_steps=cfg["optimization_steps"],
                prune_interval=cfg["prune_interval"], background=tuple(cfg["background"]),
                batch_size=cfg["sampler"]["batch_size"], active_size=cfg["sampler"]["active_size"],
                use_view_distribution=cfg["use_view_distribution"],
                lrs=dict(mean=cfg["optimizer"]["mean_lr"], scale=cfg["optimizer"]["scale_lr"],
                         rotation=cfg["optimizer"]["rotation_lr"], opacity=cfg["optimizer"]["opacity_lr"],
                         harmonic=cfg["optimizer"]["harmonic_lr"]))
    t = GaussianMapTrainer(d["raw_init"], d["frames"], mine, rasterizer_module=_oracle_module,
                           optimizer_factory=lambda p, lrs: _oracle_module.OracleAdam(p, lrs), **kw)
    np.random.seed(7)
    t.train()
    return t


def check_train_against_fixture(t, d, tol=2e-4):
    for k, ref in d["raw_final"].items():
        got = getattr(t, k)
        assert got.shape == ref.shape, k
        assert torch.allclose(got, ref, rtol=tol, atol=tol), (k, float((got - ref).abs().max()))
    assert torch.allclose(t.training_performance, d["training_performance"], rtol=1e-3, atol=1e-5)
    assert torch.equal(t.view_supports, d["view_supports"])
    assert torch.allclose(t.view_scores, d["view_scores"], atol=1e-4)
    assert torch.allclose(t.view_means, d["view_means"], atol=1e-4)


def test_train_mirror_matches_reference_train():
    d = torch.load(os.path.join(GOLD, "train.pt"))
    t = _train_from_fixture(d)
    check_train_against_fixture(t, d)
    assert len(t.last_losses) == d["steps"]


# ------------------------------------------------------------------ C ABI, no GPU compute
def test_library_exports_every_declared_symbol(agslib):
    hdr = open(os.path.join(ROOT, "include", "ags_raster.h")).read()
    declared = set(re.findall(r"\b(ags_[a-z_0-9]+)\s*\(", hdr))
    assert {"ags_forward", "ags_backward", "ags_adam_step", "ags_workspace_bytes"} <= declared
    from active_gs_amd import _lib
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    for sym in declared:
        assert hasattr(agslib, sym), sym
    assert agslib.ags_version() >= 100


def test_workspace_size_and_argument_checks(agslib):
    from active_gs_amd import _lib
    a = agslib.ags_workspace_bytes(200_000, 680, 1200, 1_000_000)
    b = agslib.ags_workspace_bytes(200_000, 680, 1200, 2_000_000)
    assert 0 < a < b and (b - a) >= 1_000_000 * 24
    assert agslib.ags_workspace_bytes(10, 0, 10, 10) == 0
    # validation happens before any HIP call, so these run without a GPU
    assert agslib.ags_forward(None, None, None, None, None, None) == -1
    cam, g, im, pg = _lib.AgsCamera(), _lib.AgsGaussians(), _lib.AgsImages(), _lib.AgsPerGaussian()
    ws = _lib.AgsWorkspace(None, 0, 10, 0)
    assert agslib.ags_forward(C.byref(cam), C.byref(g), C.byref(im), C.byref(pg), C.byref(ws), None) == -1
    assert agslib.ags_adam_step(None, 0.9, 0.999, 1e-15, 1, None) == -1
    assert agslib.ags_error_string(-2).decode().startswith("workspace")
    # round-3 entry points: argument checks come before any HIP call too
    assert agslib.ags_backward_rows(None, 1, None, None, None) == -1
    refs = (_lib.AgsViewRef * 1)()
    gg = _lib.AgsGaussianGrads()
    g.n = 4
    assert agslib.ags_backward_rows(refs, 0, C.byref(g), C.byref(gg), None) == -1          # no views
    assert agslib.ags_backward_rows(refs, 17, C.byref(g), C.byref(gg), None) == -1         # more than AGS_MAX_ROW_VIEWS
    assert agslib.ags_backward_rows(refs, 1, C.byref(g), C.byref(gg), None) == -1          # no row set
    assert agslib.ags_workspace_discard_pass(C.byref(ws), 10, 32, 32, None) == -1          # no pointer
    assert agslib.ags_read_status_async(C.byref(ws), None, None) == -1
    off, nb = C.c_size_t(), C.c_size_t()
    assert agslib.ags_workspace_region(100, 64, 64, 1 << 16, 2, 1, C.byref(off), C.byref(nb)) == 0
    assert nb.value == 64 * 64 * 4 and off.value % 256 == 0 and off.value + nb.value <= agslib.ags_workspace_bytes(100, 64, 64, 1 << 16)
    assert agslib.ags_workspace_region(100, 64, 64, 1 << 16, 2, 99, C.byref(off), C.byref(nb)) == -1
    assert agslib.ags_workspace_region(100, 64, 64, 1 << 16, 1, 4, C.byref(off), C.byref(nb)) == -1    # no key array in radix mode


def test_product_refuses_cpu_tensors(agslib):
    """No CPU fallback: the drop-in module and the trainer fail loudly off-GPU."""
    from diff_gaussian_rasterization_2d import GaussianRasterizationSettings, GaussianRasterizer
    from active_gs_amd.trainer import SurfelTrainer
    from active_gs_amd.synthetic import make_room_scene
    s = GaussianRasterizationSettings(32, 32, 1.0, 1.0, torch.zeros(4), 1.0, torch.eye(4), torch.eye(4), 0,
                                      torch.zeros(3), False, torch.tensor([]), 0.03, False, torch.tensor([1., 1, 1, 0, 0]))
    z = torch.zeros
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        GaussianRasterizer(s)(z(4, 3), z(4, 3), z(4, 1), z(4), None, z(4, 3), z(4, 3), z(4, 4), None)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        SurfelTrainer(make_room_scene(8))
    import glob
    srcs = glob.glob(os.path.join(ROOT, "active-gs_amd", "*.py")) + glob.glob(os.path.join(ROOT, "diff_gaussian_rasterization_2d", "*.py"))
    for f in srcs:  # the product never imports the checker
        assert "oracle" not in open(f).read().replace("CPU oracle", "").replace("the oracle", ""), f


def test_map_checkpoint_roundtrip_in_reference_schema(tmp_path):
    """Keys / raw tensors of GaussianMap.save (gaussian_map.py:491-507) and a load round trip."""
    from active_gs_amd.map_io import MAP_KEYS, load_map, save_map
    from active_gs_amd.map_trainer import GaussianMapTrainer
    d = torch.load(os.path.join(GOLD, "train.pt"))
    t = GaussianMapTrainer(d["raw_init"], d["frames"], dict(bound=(0.001, 10.0)), rasterizer_module=_oracle_module,
                           optimizer_factory=lambda p, lrs: _oracle_module.OracleAdam(p, lrs))
    t.view_supports += 2.0
    path = save_map(t, str(tmp_path), index=3)
    assert path.endswith("map_3.th")
    st = torch.load(path)
    assert set(st.keys()) == set(MAP_KEYS)
    assert st["harmonics"].shape == (d["n"], 1, 3) and st["scales"].shape == (d["n"], 3)
    raw, cfg = load_map(path)
    t2 = GaussianMapTrainer(raw, d["frames"], cfg, rasterizer_module=_oracle_module,
                            optimizer_factory=lambda p, lrs: _oracle_module.OracleAdam(p, lrs))
    for k in ("means", "scales", "rotations", "opacities", "harmonics", "view_supports", "view_scores", "view_means"):
        assert torch.equal(getattr(t2, k), getattr(t, k)), k
    assert t2.cfg["bound"] == (0.001, 10.0) and t2.cfg["scale_factor"] == 0.01


def test_public_header_is_plain_c(tmp_path):
    """include/ags_raster.h is the C ABI: it must compile as C (gcc, -std=c99 -pedantic) and the struct
    layouts the ctypes binding assumes must match what a C compiler lays out."""
    import subprocess
    from active_gs_amd import _lib
    src = tmp_path / "probe.c"
    fields = {"AgsCamera": _lib.AgsCamera, "AgsGaussians": _lib.AgsGaussians, "AgsImages": _lib.AgsImages,
              "AgsPerGaussian": _lib.AgsPerGaussian, "AgsImageGrads": _lib.AgsImageGrads,
              "AgsGaussianGrads": _lib.AgsGaussianGrads, "AgsWorkspace": _lib.AgsWorkspace, "AgsTuning": _lib.AgsTuning,
              "AgsStatus": _lib.AgsStatus,
              "AgsAdamTensors": _lib.AgsAdamTensors, "AgsActivation": _lib.AgsActivation,
              "AgsLossConfig": _lib.AgsLossConfig, "AgsRowSet": _lib.AgsRowSet, "AgsKeyframe": _lib.AgsKeyframe,
              "AgsDensifyPred": _lib.AgsDensifyPred, "AgsCandidates": _lib.AgsCandidates,
              "AgsNextIteration": _lib.AgsNextIteration, "AgsMapArrays": _lib.AgsMapArrays}
    lines = ['#include "ags_raster.h"', "#include <stdio.h>", "#include <stddef.h>", "int main(void) {"]
    for name, cls in fields.items():
        lines.append(f'  printf("{name} %zu\\n", sizeof({name}));')
        for fname, _ in cls._fields_:
            lines.append(f'  printf("{name}.{fname} %zu\\n", offsetof({name}, {fname}));')
    lines += ["  return 0;", "}"]
    src.write_text("\n".join(lines))
    exe = tmp_path / "probe"
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           str(src), "-o", str(exe)])
    out = dict(l.split() for l in subprocess.check_output([str(exe)], text=True).splitlines())
    for name, cls in fields.items():
        assert int(out[name]) == C.sizeof(cls), name
        for fname, _ in cls._fields_:
            assert int(out[f"{name}.{fname}"]) == getattr(cls, fname).offset, f"{name}.{fname}"


def test_device_frame_sampler_draws_the_reference_distribution():
    """cfg sampler="device" (fused_map_trainer.weighted_choice_without_replacement) against
    np.random.choice(replace=False, p=...), which is what the reference's WeightedSampler calls:
    the same inclusion frequencies and the same distribution of the first pick."""
    import numpy as np
    from active_gs_amd.fused_map_trainer import weighted_choice_without_replacement
    w = torch.tensor([0.5, 1.0, 2.0, 4.0, 0.25, 3.0])
    k, trials = 3, 20000
    torch.manual_seed(0)
    np.random.seed(0)
    inc_t, inc_n = torch.zeros(6), np.zeros(6)
    pair_t, pair_n = torch.zeros(6, 6), np.zeros((6, 6))
    p = (w / w.sum()).numpy().astype(np.float64)
    p /= p.sum()
    for _ in range(trials):
        a = weighted_choice_without_replacement(w, k)
        assert a.numel() == k and len(set(a.tolist())) == k
        inc_t[a] += 1
        pair_t[a[0], a[1]] += 1                      # order matters: topk returns the largest key first
        b = np.random.choice(6, size=k, replace=False, p=p)
        inc_n[b] += 1
        pair_n[b[0], b[1]] += 1
    assert np.abs(inc_t.numpy() - inc_n).max() / trials < 0.015
    assert np.abs(pair_t.numpy() - pair_n).max() / trials < 0.01


def test_bench_gpus_flag_launches_ranks_or_refuses():
    """``bench.py --gpus N`` is not decorative: without a launcher it starts N ranks itself (here, with no GPU, both
    ranks stop at the no-CPU-fallback check and the parent reports the failed job with a non-zero exit code and no
    result line); under a launcher whose WORLD_SIZE disagrees it refuses to print a line for the wrong job size."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env["AGS_BENCH_SHARE_GPU"] = "1"                       # skip the device-count check of the launcher
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and not [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert "2-rank job" in r.stderr
    env2 = dict(env, RANK="0", LOCAL_RANK="0", WORLD_SIZE="2")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4"], env=env2, capture_output=True,
                       text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "0"], env=env, capture_output=True,
                       text=True, timeout=120)
    assert r.returncode != 0


def test_reference_written_checkpoint_loads_and_ours_has_what_its_load_reads(tmp_path):
    """tests/golden/map_ref.th was written by the REFERENCE's GaussianMap.save (gaussian_map.py:491-507; generated by
    tests/golden/make_map_ref.py, which also ran the reference's load() on a file from map_io.save_map - recorded in
    map_ref.json).  Here: map_io.load_map reads the reference's file, and a file from map_io.save_map offers every
    access the reference's load() makes (:509-527), with the same types."""
    import json
    from active_gs_amd import map_io
    gold = os.path.join(ROOT, "tests", "golden")
    meta = json.load(open(os.path.join(gold, "map_ref.json")))
    assert meta["reference_load_of_map_io_file"] == "ok"
    raw, cfg = map_io.load_map(os.path.join(gold, "map_ref.th"))
    n = raw["means"].shape[0]
    assert n == 300 and raw["scales"].shape == (n, 3) and raw["rotations"].shape == (n, 4) and raw["harmonics"].shape == (n, 1, 3)
    assert raw["opacities"].shape == (n,) and raw["view_means"].shape == (n, 3) and raw["view_supports"].shape == (n,)
    assert float(raw["scales"][:, 2].max()) == -1e10                       # raw (pre-activation) tensors: z-scale of a surfel
    assert cfg["bound"] == (0.001, 10.0) and cfg["scale_factor"] == 0.01 and cfg["background"] == (0.0, 0.0, 0.0, 0.0)
    assert cfg["use_view_distribution"] is True
    with pytest.raises(KeyError):
        torch.save({"means": raw["means"]}, tmp_path / "bad.th")
        map_io.load_map(str(tmp_path / "bad.th"))

    class T:
        pass
    t = T()
    for k, v in raw.items():
        setattr(t, k, v)
    t.cfg = dict(bound=cfg["bound"], use_view_distribution=True, scale_factor=0.01)
    t.background = torch.zeros(4)
    path = map_io.save_map(t, str(tmp_path), 7)
    assert path.endswith("map_7.th")                                        # f"{save_path}/map_{index}.th"
    st, ref = torch.load(path), torch.load(os.path.join(gold, "map_ref.th"))
    assert sorted(st.keys()) == sorted(ref.keys()) == meta["keys"]
    for k in ("means", "scales", "harmonics", "opacities", "rotations", "view_scores", "view_supports", "view_means"):
        assert torch.equal(st[k], ref[k]) and st[k].dtype == ref[k].dtype    # what load() assigns as is
    assert isinstance(st["near"], float) and isinstance(st["far"], float) and isinstance(st["scale_factor"], float)
    bg = torch.tensor(torch.as_tensor(st["background_color"]).tolist(), dtype=torch.float32)   # load(): torch.tensor(..., float32)
    assert bg.shape == (4,) and st["use_view_direction"] == ref["use_view_direction"]


def test_frame_samplers_pick_the_reference_frames():
    """tests/golden/sampler.pt: frames picked by the reference's UniformSampler (torch.randperm) and WeightedSampler
    (np.random.choice), /root/reference/mapping/utils.py:190-261, on seeded inputs - three draws in a row each."""
    from active_gs_amd.map_trainer import UniformFrameSampler, WeightedFrameSampler, make_frame_sampler
    cases = torch.load(os.path.join(GOLD, "sampler.pt"), weights_only=False)
    for c in cases:
        frames = c["frames"]
        torch.manual_seed(c["seed"])
        us = UniformFrameSampler({10 * i: f for i, f in enumerate(frames)}, c["batch"], c["active"])
        assert us.v == c["uniform_v"]
        for want in c["uniform"]:
            rgb, depth, extr, intr, ids = us.next_frames()
            assert rgb.shape[0] == want["n"] and len(ids) == want["n"]
            assert torch.equal(rgb.sum(dim=(1, 2, 3)), want["rgb_sum"]) and torch.equal(extr[:, 0, 0], want["extr0"])
        np.random.seed(c["seed"])
        ws = WeightedFrameSampler(frames, c["batch"], c["active"])
        for want in c["weighted"]:
            rgb, depth, extr, intr, ids = ws.next_frames(c["weight"].clone())
            assert torch.equal(torch.as_tensor(np.asarray(ids)), want["ids"])
            assert torch.equal(rgb.sum(dim=(1, 2, 3)), want["rgb_sum"])
    # the config switch of gaussian_map.py:253-256
    assert isinstance(make_frame_sampler(dict(sampler_type="uniform", batch_size=8, active_size=3), cases[0]["frames"]),
                      UniformFrameSampler)
    assert isinstance(make_frame_sampler(dict(batch_size=8, active_size=3), cases[0]["frames"]), WeightedFrameSampler)
    with pytest.raises(ValueError):
        make_frame_sampler(dict(sampler_type="nope", batch_size=8, active_size=3), cases[0]["frames"])


def test_spherical_harmonics_follow_the_reference_viewer_shader():
    """eval_sh (the drop-in module's `shs=` branch) against a literal transcription of the SH block of the reference
    tree's own viewer shader, /root/reference/visualization/gl_render/shaders/gau_vert.glsl:3-18,173-205 - the only
    statement of the basis the reference holds (its mapper passes colors_precomp)."""
    from active_gs_amd.rasterizer import eval_sh
    C0, C1 = 0.28209479177387814, 0.4886025119029199
    C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
    C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
          1.445305721320277, -0.5900435899266435]
    rng = np.random.default_rng(3)
    n = 257
    sh = rng.standard_normal((n, 16, 3))
    d = rng.standard_normal((n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    for deg in range(4):
        want = np.zeros((n, 3))
        for i in range(n):     # the shader, statement by statement (render_mod >= deg, sh_dim = 48)
            x, y, z = d[i]
            g = lambda k: sh[i, k]
            color = C0 * g(0)
            if deg >= 1:
                color = color - C1 * y * g(1) + C1 * z * g(2) - C1 * x * g(3)
                if deg >= 2:
                    xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
                    color = (color + C2[0] * xy * g(4) + C2[1] * yz * g(5) + C2[2] * (2.0 * zz - xx - yy) * g(6)
                             + C2[3] * xz * g(7) + C2[4] * (xx - yy) * g(8))
                    if deg >= 3:
                        color = (color + C3[0] * y * (3.0 * xx - yy) * g(9) + C3[1] * xy * z * g(10)
                                 + C3[2] * y * (4.0 * zz - xx - yy) * g(11) + C3[3] * z * (2.0 * zz - 3.0 * xx - 3.0 * yy) * g(12)
                                 + C3[4] * x * (4.0 * zz - xx - yy) * g(13) + C3[5] * z * (xx - yy) * g(14)
                                 + C3[6] * x * (xx - 3.0 * yy) * g(15))
            want[i] = color
        got = eval_sh(deg, torch.from_numpy(sh), torch.from_numpy(d)).numpy()
        assert np.abs(got - want).max() < 1e-12, deg
    with pytest.raises(ValueError):
        eval_sh(2, torch.zeros(4, 4, 3), torch.zeros(4, 3))


def test_kernel_selection_travels_with_the_workspace_not_with_the_environment(agslib):
    """include/ags_raster.h: AgsTuning is handed over with the workspace; the LIBRARY reads no environment variable (its
    code object has no getenv import), the Python binding maps the documented AGS_* variables onto the struct, and an
    out-of-range selection is refused by the entry points."""
    import subprocess
    from active_gs_amd import _lib, build
    syms = subprocess.check_output(["nm", "-D", "--undefined-only", build.LIB], text=True)
    assert "getenv" not in syms
    t = _lib.tuning_from_env({})
    assert (t.bwd_reduce, t.render_slots, t.cull_first_min_n, t.tile_sort_no_wave, t.bucket_no_scan) == (0, 0, 0, 0, 0)
    assert _lib.tuning_from_env({"AGS_BWD_REDUCE": "bf16"}).bwd_reduce == _lib.BWD_BF16_SPLIT
    assert _lib.tuning_from_env({"AGS_BWD_BF16": "1"}).bwd_reduce == _lib.BWD_BF16_SPLIT
    assert _lib.tuning_from_env({"AGS_BWD_MFMA": "0"}).bwd_reduce == _lib.BWD_VALU
    assert _lib.tuning_from_env({"AGS_RENDER_SLOTS": "2", "AGS_PRE_CULL_MIN_N": "0"}).render_slots == 2
    assert _lib.tuning_from_env({"AGS_PRE_CULL_MIN_N": "0"}).cull_first_min_n == 1          # "always"
    with pytest.raises(ValueError):
        _lib.tuning_from_env({"AGS_BWD_REDUCE": "fp8"})
    # the entry points validate the struct before they touch anything (null device pointers here: AGS_E_INVALID either way,
    # so compare with a call that passes its checks up to the workspace size)
    bad = _lib.AgsTuning(7, 0, 0, 0, 0)
    ws = _lib.workspace(1 << 20, 0, 10, 0, bad)         # a non-null "pointer", 0 bytes
    cam = _lib.AgsCamera(16, 16, 1.0, 1.0, 1.0, 0.03, 1, 1, 0, 0, 1 << 20, 1 << 20, 1 << 20, None, None)
    g = _lib.AgsGaussians(0, None, None, None, None, None, None, 0, 0.01, 0.05)
    im = _lib.AgsImages(*([1 << 20] * 5))
    pg = _lib.AgsPerGaussian(None, None, None, _lib.AgsRowSet(None, None, None))
    assert agslib.ags_forward(C.byref(cam), C.byref(g), C.byref(im), C.byref(pg), C.byref(ws), None) == -1      # AGS_E_INVALID
    ok = _lib.workspace(1 << 20, 0, 10, 0, _lib.AgsTuning(1, 2, 0, 0, 0))
    assert agslib.ags_forward(C.byref(cam), C.byref(g), C.byref(im), C.byref(pg), C.byref(ok), None) == -2      # AGS_E_WORKSPACE


def test_gaussian_map_class_reads_cfg_like_the_reference_and_serves_a_cpu_map(tmp_path):
    """active_gs_amd.gaussian_map.GaussianMap: the reference's constructor contract (gaussian_map.py:18-60: attribute
    access on cfg.bound / cfg.background / cfg.optimizer.* / cfg.sampler.*; cfg = None for maps that are only loaded),
    its getters on a CPU map, the checkpoint in the reference's schema - and no CPU fallback for anything that computes."""
    from types import SimpleNamespace as NS
    from active_gs_amd.gaussian_map import GaussianMap
    cfg = NS(bound=[0.002, 8.0], background=[0.1, 0.2, 0.3, 0.0], sparse_ratio=0.1, error_thres=0.3, scale_factor=0.02,
             optimization_steps=7, prune_interval=4, use_view_distribution=True,
             sampler=NS(sampler_type="weighted", batch_size=6, active_size=2),
             optimizer=NS(mean_lr=1e-3, rotation_lr=2e-3, opacity_lr=3e-3, scale_lr=4e-3, harmonic_lr=5e-3))
    gm = GaussianMap(cfg, "cpu")
    assert (gm.scene_near, gm.scene_far, gm.scale_factor, gm.error_thres, gm.prune_interval, gm.optimization_steps) == \
           (0.002, 8.0, 0.02, 0.3, 4, 7)
    assert torch.equal(gm.background_color, torch.tensor([0.1, 0.2, 0.3, 0.0])) and not gm.is_init and gm.training_data == []
    tc = gm._trainer_cfg()
    assert tc["batch_size"] == 6 and tc["active_size"] == 2 and tc["lrs"] == dict(mean=1e-3, scale=4e-3, rotation=2e-3,
                                                                                 opacity=3e-3, harmonic=5e-3)
    assert tc["bound"] == (0.002, 8.0) and tc["optimization_steps"] == 7
    gm.optimization_steps = 3                              # attributes stay live, like the reference's
    assert gm._trainer_cfg()["optimization_steps"] == 3
    # a reference-written checkpoint through load(): the getters are the reference's expressions
    ref = torch.load(os.path.join(GOLD, "map_ref.th"))
    g2 = GaussianMap(None, "cpu")
    g2.load(os.path.join(GOLD, "map_ref.th"))
    assert g2.is_init and g2.scene_far == ref["far"] and g2.scale_factor == ref["scale_factor"]
    means, harmonics, opac, conf, scales, rot = g2.get_attr()
    n = ref["means"].shape[0]
    assert means.shape == (n, 3) and harmonics.shape == (n, 1, 3) and opac.shape == conf.shape == (n,)
    assert torch.equal(opac, torch.sigmoid(ref["opacities"])) and torch.equal(rot, torch.nn.functional.normalize(ref["rotations"]))
    assert torch.equal(scales, torch.clamp(ref["scale_factor"] * torch.exp(ref["scales"]), min=0, max=0.05))
    var = ref["view_means"].norm(dim=-1)
    assert torch.allclose(conf, torch.clamp(torch.exp(1 - var) * ref["view_scores"], min=0, max=1))
    nrm = g2.get_normals                                    # voxel_map.py:72; third column of the rotation matrix
    r, x, y, z = rot.unbind(-1)
    assert torch.allclose(nrm, torch.nn.functional.normalize(torch.stack([2 * (x * z + r * y), 2 * (y * z - r * x),
                                                                          1 - 2 * (x * x + y * y)], -1)))
    g2.save(str(tmp_path), index="007")                     # utils/common.py:249
    back = torch.load(os.path.join(str(tmp_path), "map_007.th"))
    assert sorted(back) == sorted(ref) and all(torch.equal(back[k], ref[k]) for k in ("means", "scales", "view_means"))
    for call in (lambda: g2.train(), lambda: g2.post_processing(), lambda: g2.update({}), lambda: g2.prune(torch.zeros(n))):
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            call()


def test_drop_in_classes_cover_the_reference_classes_public_surface():
    """tests/golden/class_surface.json = the method names / parameter lists / properties / constructor-set attributes of the
    reference's ``mapping.gaussian_map.GaussianMap`` and ``utils.operations.GaussianRenderer`` (captured by importing the
    reference in the build container, make_surface.py).  The classes that replace them BY NAME must offer every one of them:
    same methods taking the reference's parameters in the reference's order (more, defaulted, ones may follow), the same
    properties as properties, and every attribute an outside caller could read."""
    import inspect
    import json
    from types import SimpleNamespace as NS
    from active_gs_amd.facade import SurfelRenderer
    from active_gs_amd.gaussian_map import GaussianMap
    ref = json.load(open(os.path.join(GOLD, "class_surface.json")))
    cfg = NS(bound=[0.001, 10.0], background=[0.0, 0.0, 0.0, 0.0], sparse_ratio=0.1, error_thres=0.25, scale_factor=0.01,
             optimization_steps=10, prune_interval=5, use_view_distribution=True,
             sampler=NS(sampler_type="weighted", batch_size=8, active_size=3),
             optimizer=NS(mean_lr=5e-4, rotation_lr=5e-4, opacity_lr=1e-2, scale_lr=1e-2, harmonic_lr=1e-4))
    extr = torch.eye(4)[None]
    K = torch.tensor([[[0.866, 0, 0.5], [0, 0.866, 0.5], [0, 0, 1.0]]])
    attr = (torch.zeros(1, 3), torch.zeros(1, 1, 3), torch.zeros(1), torch.zeros(1), torch.zeros(1, 3), torch.tensor([[1.0, 0, 0, 0]]))
    instances = {"GaussianMap": (GaussianMap, GaussianMap(cfg, "cpu")),
                 "GaussianRenderer": (SurfelRenderer, SurfelRenderer(extr, K, attr, torch.zeros(4), (0.001, 10.0), (16, 16), "cpu",
                                                                     rasterizer_module=_oracle_module))}
    for name, (cls, inst) in instances.items():
        want = ref[name]
        for m, params in want["methods"].items():
            assert callable(getattr(cls, m, None)), (name, m)
            mine = list(inspect.signature(getattr(cls, m)).parameters.values())
            assert [p.name for p in mine[:len(params)]] == [p["name"] for p in params], (name, m)
            for p, q in zip(mine, params):
                assert (p.default is not inspect.Parameter.empty) == q["has_default"], (name, m, p.name)
                if q["has_default"]:
                    assert repr(p.default) == q["default"], (name, m, p.name)
            assert all(p.default is not inspect.Parameter.empty for p in mine[len(params):]), (name, m)   # extras are optional
        for prop in want["properties"]:
            assert isinstance(inspect.getattr_static(cls, prop), property), (name, prop)
        for a in want["instance_attributes"]:
            assert hasattr(inst, a), (name, a)
    assert instances["GaussianRenderer"][1].raydir_map.shape == (3, 16, 16)


def test_row_range_form_of_backward_rows_accepts_a_further_group_of_views(agslib):
    """Round 5: a data-parallel rank with more views than one ``ags_backward_rows`` launch joins (AGS_MAX_ROW_VIEWS = 16: all
    32 views of configuration 4 on one or two GPUs) calls the row-range form once per group, the later groups ADDING to the
    chunk's rows (accumulate 1).  The argument checks run before any HIP call: with a view whose workspace is too small the
    call gets as far as that size check (AGS_E_WORKSPACE = -2) exactly when the range form's own checks let it through -
    accumulate 0 and 1 do, accumulate 2, a fused tail or a range outside the map do not (AGS_E_INVALID = -1)."""
    from active_gs_amd import _lib
    buf = (C.c_float * 400)()
    addr = C.cast(buf, C.c_void_p).value
    cam = _lib.AgsCamera()
    cam.image_height, cam.image_width, cam.viewmatrix, cam.projmatrix = 32, 32, addr, addr
    ws = _lib.AgsWorkspace(addr, 1, 1 << 16, 2)                        # one byte of workspace: too small for anything
    refs = (_lib.AgsViewRef * 1)()
    refs[0].cam, refs[0].radii, refs[0].ws = C.pointer(cam), addr, C.pointer(ws)
    g, gg = _lib.AgsGaussians(), _lib.AgsGaussianGrads()
    g.n = 100
    for f in ("means3D", "scales", "rotations", "opacities"):
        setattr(g, f, addr)
    for f in ("d_means3D", "d_scales", "d_rotations", "d_opacities", "d_colors"):
        setattr(gg, f, addr)
    gg.row_begin, gg.row_end = 0, 64
    for acc, want in ((0, -2), (1, -2), (2, -1)):
        gg.accumulate = acc
        assert agslib.ags_backward_rows(refs, 1, C.byref(g), C.byref(gg), None) == want, acc
    gg.accumulate, gg.row_end = 1, 101                                 # a range that leaves the map
    assert agslib.ags_backward_rows(refs, 1, C.byref(g), C.byref(gg), None) == -1
    gg.row_end, gg.pack_segment = 64, addr                             # no exchange segment in the range form
    assert agslib.ags_backward_rows(refs, 1, C.byref(g), C.byref(gg), None) == -1


def test_rows_of_two_grown_maps_are_paired_by_origin():
    """tests/_origin.py: the final parameters of two maps that were GROWN are compared over the rows both hold - spawned by
    the same keyframe at the same place and kept by both - whatever the two row counts are."""
    from _origin import common_rows
    gen = torch.Generator().manual_seed(0)
    k0 = torch.rand(50, 3, generator=gen)
    k1 = torch.rand(30, 3, generator=gen) + 2.0
    # mine: keyframe 0 spawned the same rows in another order and one extra row; keyframe 1 misses a row
    perm = torch.randperm(50, generator=gen)
    mine0 = torch.cat([k0[perm] + 1e-6, torch.tensor([[9.0, 9.0, 9.0]])])
    mine1 = k1[1:]
    ref_origin = torch.cat([torch.arange(50), (1 << 32) + torch.arange(30)])
    ref_origin = ref_origin[ref_origin != 7]                                   # the reference pruned its row (0, 7)
    my_origin = torch.cat([torch.arange(51), (1 << 32) + torch.arange(29)])
    ri, mi, stats = common_rows([k0, k1], ref_origin, [mine0, mine1], my_origin)
    assert stats[0]["same_place"] == 50 and stats[1]["same_place"] == 29
    assert ri.numel() == mi.numel() == 49 + 29                                 # (0, 7) is gone on one side, (1, 0) was never spawned on the other
    ref_rows = torch.cat([k0, k1])[torch.cat([torch.arange(50) != 7, torch.ones(30, dtype=torch.bool)])]
    my_rows = torch.cat([mine0, mine1])
    assert float((ref_rows[ri] - my_rows[mi]).abs().max()) < 1e-5              # every pair IS the same spawned row


def test_bench_strong_configurations_and_replica_checksums():
    """bench.py's strong-scaled configurations 4 and 5: BASELINE.json's sizes by default (32 views of 1.5 M surfels @1200x680, 8
    views of 5 M @2048x2048), a test's reduced sizes from AGS_BENCH_STRONG; the replica check is a bit checksum per tensor."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    was = os.environ.pop("AGS_BENCH_STRONG", None)
    try:
        c = bench.strong_configs()
        assert (c["c4"]["n"], c["c4"]["views"], c["c4"]["h"], c["c4"]["w"]) == (1_500_000, 32, 680, 1200)
        assert (c["c5"]["n"], c["c5"]["views"], c["c5"]["h"], c["c5"]["w"]) == (5_000_000, 8, 2048, 2048) and "reduced" not in c["c5"]
        os.environ["AGS_BENCH_STRONG"] = "c4=150000,8,680,1200"
        c = bench.strong_configs()
        assert c["c4"]["n"] == 150000 and c["c4"]["views"] == 8 and c["c4"]["reduced"] and c["c5"]["n"] == 5_000_000
    finally:
        os.environ.pop("AGS_BENCH_STRONG", None)
        if was is not None:
            os.environ["AGS_BENCH_STRONG"] = was

    class T:
        params = [torch.arange(6, dtype=torch.float32).reshape(2, 3), torch.ones(4)]
    ok, sums = bench.replica_checksums(T, False)
    assert ok and sums == [int(t.view(torch.int32).to(torch.int64).sum()) for t in T.params]
    T.params[1][2] = 1.0000001                                                 # one ulp: another checksum
    assert bench.replica_checksums(T, False)[1] != sums


def test_the_package_reads_no_environment_variable():
    """Kernel selection travels as AgsTuning, everything else is an argument or a class attribute: no module of the
    package (nor the drop-in module) touches os.environ; launchers call env_config.apply_env(mapping) explicitly, and
    that function maps the documented AGS_* variables without reading the process environment itself."""
    import ast
    import glob
    files = glob.glob(os.path.join(ROOT, "active-gs_amd", "*.py")) + glob.glob(os.path.join(ROOT, "diff_gaussian_rasterization_2d", "*.py"))
    assert len(files) > 15
    for f in files:
        for node in ast.walk(ast.parse(open(f).read())):
            assert not (isinstance(node, ast.Attribute) and node.attr in ("environ", "getenv", "environb", "putenv")), (f, node.lineno)
    from active_gs_amd import _lib, env_config
    from active_gs_amd.trainer import SurfelTrainer
    saved = (_lib._default_tuning, _lib.cull_choice_pinned, SurfelTrainer.DENSE_CHUNKS, SurfelTrainer.CULL_ADAPT)
    try:
        ch = env_config.apply_env({"AGS_BWD_REDUCE": "bf16", "AGS_PRE_CULL_MIN_N": "0", "AGS_DENSE_CHUNKS": "2", "AGS_CULL_ADAPT": "0"})
        assert _lib.default_tuning().bwd_reduce == _lib.BWD_BF16_SPLIT and _lib.default_tuning().cull_first_min_n == 1
        assert _lib.cull_choice_pinned and SurfelTrainer.DENSE_CHUNKS == 2 and SurfelTrainer.CULL_ADAPT is False
        assert set(ch) == {"tuning", "DENSE_CHUNKS", "CULL_ADAPT"}
        assert env_config.apply_env({}) == {}
    finally:
        _lib._default_tuning, _lib.cull_choice_pinned, SurfelTrainer.DENSE_CHUNKS, SurfelTrainer.CULL_ADAPT = saved
