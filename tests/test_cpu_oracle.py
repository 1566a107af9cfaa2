"""CPU suite, part 1: the oracle itself — against its committed vectors (drift), against
finite differences in float64 (autograd consistency), and the hand-derived backward of the
product's math header (tests/host_emu drives active-gs_amd/csrc/surfel_math.h) against it."""
import os

import pytest
import torch

import _emu
from _scenes import oracle_inputs, room_case
from oracle.surfel_oracle import OracleSettings, adam_step, rasterize

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _case(d):
    a, S = room_case(d["n"], d["h"], d["w"], view=d["view"], seed=d["seed"], scale_mult=d["mult"], config=d["config"])
    return a, S


@pytest.mark.parametrize("tag", ["small", "c1"])
def test_oracle_reproduces_committed_vectors(tag):
    d = torch.load(os.path.join(GOLD, f"oracle_{tag}.pt"))
    a, S = _case(d)
    ins = oracle_inputs(a)
    for x, y in zip(ins, d["inputs"]):
        assert torch.equal(x.detach(), y)  # the seeded scene generator did not drift either
    outs = rasterize(*ins, S)
    for o, r in zip(outs, d["outputs"]):
        if o.dtype.is_floating_point:
            assert torch.allclose(o.detach(), r, rtol=1e-5, atol=1e-6)
        else:
            assert torch.equal(o, r)
    sum((o * g).sum() for o, g in zip(outs[:5], d["image_grads"])).backward()
    for i, r in d["grads"].items():
        rel = (ins[i].grad - r).abs().sum() / r.abs().sum().clamp_min(1e-12)
        assert rel < 1e-4, (i, rel)


def test_oracle_autograd_vs_finite_differences_fp64():
    a, S = room_case(12, 32, 32, view=0, seed=3, scale_mult=8.0)
    ins = [t.double() for t in oracle_inputs(a, requires_grad=False)]
    S64 = OracleSettings(S.image_height, S.image_width, S.tanfovx, S.tanfovy, S.bg.double(), 1.0,
                         S.viewmatrix.double(), S.projmatrix.double(), config=S.config)
    gen = torch.Generator().manual_seed(1)
    wts = None

    def f(means, opac, col, sc, rot):
        nonlocal wts
        outs = rasterize(means, ins[1], opac, ins[3], col, sc, rot, S64)
        if wts is None:
            wts = [torch.randn(o.shape, generator=gen, dtype=torch.float64) for o in outs[:5]]
        return sum((o * w).sum() for o, w in zip(outs[:5], wts))

    leaves = [ins[i].clone().requires_grad_(True) for i in (0, 2, 4, 5, 6)]
    f(*leaves).backward()
    eps = 1e-6
    gen2 = torch.Generator().manual_seed(2)
    for li, leaf in enumerate(leaves):
        for _ in range(6):  # random coordinates; thresholds (1/255, T<1e-4) make rare kinks
            idx = tuple(int(torch.randint(0, s, (1,), generator=gen2)) for s in leaf.shape)
            if li == 3 and idx[-1] == 2:
                continue  # z-scale is identically zero for surfels
            args_p = [l.detach().clone() for l in leaves]
            args_m = [l.detach().clone() for l in leaves]
            args_p[li][idx] += eps
            args_m[li][idx] -= eps
            fd = (f(*args_p) - f(*args_m)) / (2 * eps)
            an = leaf.grad[idx]
            assert abs(fd - an) <= 1e-4 * max(1.0, abs(an)), (li, idx, float(fd), float(an))


def test_oracle_adam_matches_torch_optim_vector():
    d = torch.load(os.path.join(GOLD, "adam.pt"))
    p = [x.clone() for x in d["p0"]]
    m = [torch.zeros_like(x) for x in p]
    v = [torch.zeros_like(x) for x in p]
    for t, grads in enumerate(d["grads"], start=1):
        adam_step(p, grads, m, v, d["lrs"], t, eps=d["eps"])
    for a, b in zip(p, d["p3"]):
        assert torch.allclose(a, b, rtol=1e-6, atol=1e-8)


@pytest.mark.parametrize("tag", ["small", "c1"])
def test_product_math_header_matches_oracle(tag):
    """surfel_math.h (what the HIP kernels execute per lane), driven sequentially on the CPU."""
    d = torch.load(os.path.join(GOLD, f"oracle_{tag}.pt"))
    a, S = _case(d)
    means, _, opac, conf, col, sc, rot = d["inputs"]
    e = _emu.forward(S, means, sc, rot, opac, col, conf)
    names = ["rgb", "normal", "depth", "opacity", "confidence", "importance", "count", "radii"]
    for n, r in zip(names, d["outputs"]):
        if r.dtype.is_floating_point:
            tol = 1e-4 if n in ("depth", "importance") else 1e-5
            assert (e[n] - r).abs().mean() < tol, n
        else:
            assert (e[n] != r).float().mean() < 1e-3, n
    gr = d["image_grads"]
    g = _emu.backward(S, means, sc, rot, opac, col, conf, e, gr[0], gr[1], gr[2], gr[3], gr[4])
    for n, i in [("means", 0), ("means2d", 1), ("opac", 2), ("colors", 4), ("scales", 5), ("rots", 6)]:
        ref = d["grads"][i].reshape(g[n].shape)
        rel = (ref - g[n]).abs().sum() / ref.abs().sum().clamp_min(1e-12)
        assert rel < 1e-3, (n, float(rel))


# ---------------------------------------------------------------------------------------------
# map growth / pruning oracle (oracle/densify_oracle.py) against the reference's own
# GaussianMap.add_gaussians / prune outputs (tests/golden/densify.pt, make_golden.py)
def _empty_state():
    z = torch.zeros
    return dict(means=z(0, 3), scales=z(0, 3), rotations=z(0, 4), opacities=z(0), harmonics=z(0, 1, 3),
                view_scores=z(0), view_supports=z(0), view_means=z(0, 3))


def _same_state(a, b, rot_tol=5e-4):
    for k in b:
        assert a[k].shape == b[k].shape, k
        tol = rot_tol if k == "rotations" else 1e-6
        assert float((a[k] - b[k]).abs().max()) <= tol if a[k].numel() else True, k


def test_densify_oracle_matches_reference_add_gaussians_and_prune():
    from oracle import densify_oracle as dor
    g = torch.load(os.path.join(GOLD, "densify.pt"))
    first = dor.add_gaussians(_empty_state(), g["frames"][0], None, g["error_thres"])
    assert first["means"].shape[0] == g["first"]["state"]["means"].shape[0] > 1000
    _same_state(first, g["first"]["state"])
    pred = g["second"]["pred"]
    p2 = dict(rgb=pred["rgb"][0], depth=pred["depth"][0], opacity=pred["opacity"][0])
    second = dor.add_gaussians(g["second"]["before"], g["frames"][1], p2, g["error_thres"])
    n0 = g["second"]["before"]["means"].shape[0]
    assert n0 < second["means"].shape[0] == g["second"]["state"]["means"].shape[0] < n0 + g["h"] * g["w"]
    _same_state(second, g["second"]["state"])
    pruned = dor.prune(g["before_prune"], g["prune_mask"])
    _same_state(pruned, g["after_prune"], rot_tol=0.0)


def test_densify_oracle_voxel_rule_and_bilateral_properties():
    import numpy as np
    from oracle import densify_oracle as dor
    gen = torch.Generator().manual_seed(0)
    pts = torch.rand(5000, 3, generator=gen) * 0.2                      # 10^3 voxels of 2 cm: many collisions
    sel = torch.rand(5000, generator=gen) > 0.3
    keep = dor.voxel_select_last(pts, sel)
    keys = dor.voxel_keys(pts)
    assert not bool((keep & ~sel).any())
    kept_keys = {tuple(k.tolist()) for k in keys[keep]}
    assert len(kept_keys) == int(keep.sum()) == len({tuple(k.tolist()) for k in keys[sel]})
    for i in torch.nonzero(keep).flatten()[:50]:                        # the kept point is the last of its voxel
        same = (keys == keys[i]).all(1) & sel
        assert int(torch.nonzero(same).max()) == int(i)
    # bilateral: constants are fixed points, output stays within the local range, invalid -> -1
    img = np.full((40, 50), 2.5, np.float32)
    assert np.allclose(dor.bilateral_filter(img), 2.5)
    rng = np.random.default_rng(1)
    img = (2.0 + 0.01 * rng.standard_normal((40, 50))).astype(np.float32)
    img[:, 25:] += 3.0                                                  # a depth edge >> sigma_color survives
    out = dor.bilateral_filter(img)
    assert out.min() >= img.min() - 1e-6 and out.max() <= img.max() + 1e-6
    assert abs(out[:, :25].mean() - 2.0) < 1e-2 and abs(out[:, 25:].mean() - 5.0) < 1e-2
    assert out[:, :25].std() < 0.5 * img[:, :25].std()
    d = img.copy(); d[3:6, 4:9] = -1.0
    sm = dor.smooth_depth(d)
    assert np.all(sm[3:6, 4:9] == -1.0) and np.all(sm[d >= 0] >= 0)


def test_preprocess_conic_matches_the_reference_tree_ewa_statement():
    """The only statement of the projection arithmetic INSIDE the reference tree is the (dead) GL splat renderer:
    /root/reference/visualization/gl_render/shaders/gau_vert.glsl:60-107 (computeCov3D, computeCov2D: Sigma = (SR)^T(SR),
    J with x/z, y/z clamped to 1.3 tan(fov/2), cov = T^T Sigma^T T with T = W J, +0.3 on the diagonal), :148-160 (conic =
    inverse) and gau_frag.glsl:20-26 (alpha = min(0.99, a exp(power)), discard power > 0 and alpha < 1/255).  Restated
    here literally - GLSL mat3(...) fills COLUMNS - and compared with the oracle's per-Gaussian stage: decisions D2 / D4
    of oracle/surfel_oracle.py are the reference tree's own constants, and a surfel (z-scale 0, gaussian_map.py:373)
    is that 3-D Gaussian's flat limit."""
    import numpy as np
    from oracle.surfel_oracle import ALPHA_MAX, ALPHA_MIN, FRUSTUM_CLAMP, LOWPASS, preprocess

    def glsl_mat3(*cols9):                       # mat3(a,b,c, d,e,f, g,h,i): columns (a,b,c), (d,e,f), (g,h,i)
        return np.array(cols9, dtype=np.float64).reshape(3, 3).T

    def compute_cov3d(scale, q):                 # gau_vert.glsl:60-80
        S = np.diag(scale.astype(np.float64))
        r, x, y, z = [float(v) for v in q]
        R = glsl_mat3(1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                      2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                      2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y))
        M = S @ R
        return M.T @ M

    def compute_cov2d(t, fx, fy, tanx, tany, cov3d, view3):   # gau_vert.glsl:82-107; view3 = mat3(viewmatrix)
        t = t.astype(np.float64).copy()
        limx, limy = 1.3 * tanx, 1.3 * tany
        t[0] = min(limx, max(-limx, t[0] / t[2])) * t[2]
        t[1] = min(limy, max(-limy, t[1] / t[2])) * t[2]
        J = glsl_mat3(fx / t[2], 0, -(fx * t[0]) / (t[2] * t[2]), 0, fy / t[2], -(fy * t[1]) / (t[2] * t[2]), 0, 0, 0)
        W = view3.T
        T = W @ J
        cov = T.T @ cov3d.T @ T
        return cov[0, 0] + 0.3, cov[0, 1], cov[1, 1] + 0.3

    assert (LOWPASS, FRUSTUM_CLAMP, ALPHA_MAX) == (0.3, 1.3, 0.99) and abs(ALPHA_MIN - 1.0 / 255.0) < 1e-12
    for z_scale in (0.0, 1e-6, 0.4):             # the surfel limit, nearly flat, a genuinely 3-D Gaussian
        a, S = room_case(300, 96, 128, view=2, seed=5, scale_mult=3.0)
        a["scales"][:, 2] = z_scale * a["scales"][:, 0]
        ins = [t.double() for t in oracle_inputs(a, requires_grad=False)]
        G = preprocess(*ins, S)
        vis = G["vis"]
        assert vis.numel() > 40
        V = S.viewmatrix.double().numpy()        # row-vector convention: t = p @ V[:3, :3] + V[3, :3]
        A = V[:3, :3].T                          # column-vector rotation (what GLSL's mat3(viewmatrix) holds)
        fx, fy = S.image_width / (2 * S.tanfovx), S.image_height / (2 * S.tanfovy)
        worst = 0.0
        for k, i in enumerate(vis.tolist()):
            p = ins[0][i].numpy()
            t = A @ p + V[3, :3]
            cov3d = compute_cov3d(ins[5][i].numpy(), ins[6][i].numpy())
            ca, cb, cc = compute_cov2d(t, fx, fy, S.tanfovx, S.tanfovy, cov3d, A)
            det = ca * cc - cb * cb
            conic = np.array([cc / det, -cb / det, ca / det])              # gau_vert.glsl:153-155
            got = G["conic"][k].numpy()
            worst = max(worst, float(np.abs(got - conic).max() / np.abs(conic).max()))
        assert worst < 1e-9, (z_scale, worst)
    # gau_frag.glsl:20-26 against the oracle's blend of ONE surfel over an empty background: opacity image = alpha
    a, S = room_case(1, 48, 64, view=0, seed=3, scale_mult=8.0)
    a["means"][0] = torch.tensor([0.0, 0.0, 0.0])
    from active_gs_amd.synthetic import make_camera
    from active_gs_amd.camera import camera_matrices
    ins = [t.double() for t in oracle_inputs(a, requires_grad=False)]
    c2w = torch.eye(4); c2w[:3, 3] = torch.tensor([0.0, 0.0, -1.0])          # camera 1 m in front of the surfel, looking at it
    _, K = make_camera(0, 48, 64)
    cm = camera_matrices(c2w[None], K[None], 0.001, 10.0)
    S2 = OracleSettings(48, 64, cm["tanfov"][0, 0].item(), cm["tanfov"][0, 1].item(), torch.zeros(4), 1.0,
                        cm["viewmatrix"][0], cm["projmatrix"][0])
    out = rasterize(*ins, S2)
    G = preprocess(*ins, S2)
    assert G["vis"].numel() == 1
    m, con, o = G["mean2D"][0].numpy(), G["conic"][0].numpy(), float(ins[2][0])
    ys, xs = np.mgrid[0:48, 0:64].astype(np.float64)
    dx, dy = xs - m[0], ys - m[1]
    power = -0.5 * (con[0] * dx * dx + con[2] * dy * dy) - con[1] * dx * dy
    alpha = np.minimum(0.99, o * np.exp(power))
    alpha[(power > 0) | (alpha < 1.0 / 255.0)] = 0.0
    assert alpha.max() > 0.05 and np.abs(out[3][0].numpy() - alpha).max() < 1e-12


@pytest.mark.parametrize("tag", ["small", "c1"])
def test_oracle_against_the_real_extension_when_present(tag):
    """The only route from "parity unpinned" to pinned: tests/golden/extension_<tag>.pt, written by
    tests/tools/pin_against_extension.py on a machine that has the CUDA extension ActiveGS really runs
    (/root/reference/envs/requirements.txt:15), holds that extension's outputs and gradients for this repository's
    committed oracle scenes, one call per config variant, each variant isolating one of the oracle's decisions
    (DESIGN.md section 2, D5-D11).  Absent (as in the build container: no CUDA, no wheel, no source) -> skipped; present
    -> the oracle must agree within BASELINE.json's tolerances, and every decision that differs is named."""
    path = os.path.join(GOLD, f"extension_{tag}.pt")
    if not os.path.exists(path):
        pytest.skip("no extension_*.pt: run tests/tools/pin_against_extension.py where diff_gaussian_rasterization_2d (the CUDA "
                    "wheel) is installed; until then the oracle is pinned to the published algorithm only")
    d = torch.load(path)
    names = ("rgb", "normal", "depth", "opacity", "confidence", "importance", "count", "radii")
    differing = []
    for vname, v in d["variants"].items():
        a, S = room_case(d["n"], d["h"], d["w"], view=d["view"], seed=d["seed"], scale_mult=d["mult"], config=tuple(v["config"]),
                         mask=d["mask"] if v["masked"] else None)
        ins = oracle_inputs(a)
        for x, y in zip(ins, d["inputs"]):
            assert torch.equal(x.detach(), y)
        outs = rasterize(*ins, S)
        bad = []
        for k, o, r in zip(names, outs, v["outputs"]):
            if o.dtype.is_floating_point:
                tol = 1e-3 if k == "depth" else 1e-4                     # BASELINE.json: 1e-4 mean L1 (depth in metres)
                scale = max(1.0, float(r.abs().mean())) if k == "importance" else 1.0
                if float((o.detach() - r).abs().mean()) > tol * scale:
                    bad.append(f"{k}: mean L1 {float((o.detach() - r).abs().mean()):.3g}")
            elif int((o != r.to(o.dtype)).sum()) > max(2, o.numel() // 10_000):   # (rows on a ceil boundary may differ)
                bad.append(f"{k}: {int((o != r.to(o.dtype)).sum())} of {o.numel()} entries")
        if v["grads"]:
            sum((o * g).sum() for o, g in zip(outs[:5], d["image_grads"])).backward()
            for i, r in v["grads"].items():
                if i == 1:
                    continue                                               # means2D: returned, never read (operations.py:676)
                rel = float((ins[i].grad - r).abs().sum() / r.abs().sum().clamp_min(1e-12))
                if rel > 1e-3:
                    bad.append(f"gradient of input {i}: relative L1 {rel:.3g}")
        if bad:
            differing.append(f"{vname} {v['config']} -> {'; '.join(bad)}   [decides: {', '.join(v['decisions'])}]")
    assert not differing, "the oracle differs from the extension (" + str(d.get("extension")) + "):\n" + "\n".join(differing)
