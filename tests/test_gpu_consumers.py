"""What the reference's CONSUMERS of the rasterizer's outputs rely on, stated as tests on constructed scenes (HIP path
through the drop-in module, next to the oracle).  The CUDA extension's source is absent (parity unpinned, DESIGN.md
section 2); these are the reference-held statements that constrain its semantics:

  * planning/confidence.py:47-101   the confidence image is used as 1 - confidence (so it lies in [0, 1]) and a pixel
                                    with depth < 0.001 is an "unseen surface" (so an empty pixel renders depth 0);
  * mapping/gaussian_map.py:195     counts[-1] >= 1  <=> the surfel is visible in the newest view;
    mapping/gaussian_map.py:229-232 sum(counts) >= 1 <=> keep it (prune otherwise) - render_view_all(require_importance=
                                    True, front_only=True, render_masks=(depth_gt > 0)), :183-192;
  * mapping/gaussian_map.py:482-485 cal_mask: opacity < 0.5 marks a hole (new surfels are spawned there) and
                                    depth_gt - depth < -0.05 depth_gt marks a surface in front of the measurement.
"""
import pytest
import torch

from _scenes import oracle_inputs, product_settings

pytestmark = pytest.mark.gpu

FRONT = torch.tensor([0.0, 1.0, 0.0, 0.0])   # normal (0, 0, -1): faces a camera at the origin looking along +z
BACK = torch.tensor([1.0, 0.0, 0.0, 0.0])    # normal (0, 0, +1): faces away


def _settings(h, w, config, mask=None, t=0.8, bg=(0.0, 0.0, 0.0, 0.0)):
    from oracle.surfel_oracle import OracleSettings
    near, far = 0.001, 10.0
    P = torch.zeros(4, 4)
    P[0, 0] = 1 / t; P[1, 1] = 1 / t; P[3, 2] = 1; P[2, 2] = far / (far - near); P[2, 3] = -far * near / (far - near)
    return OracleSettings(h, w, t, t, torch.tensor(bg), 1.0, torch.eye(4), (torch.eye(4) @ P.t()).contiguous(),
                          campos=torch.zeros(3), render_mask=mask, config=torch.tensor([float(c) for c in config]))


def _both(a, S):
    """(oracle outputs, HIP outputs on the CPU)"""
    from diff_gaussian_rasterization_2d import GaussianRasterizer, check_overflow
    from oracle.surfel_oracle import rasterize
    dev = torch.device("cuda:0")
    ins = oracle_inputs(a, requires_grad=False)
    with torch.no_grad():
        ref = rasterize(*ins, S)
        gin = [t.to(dev) for t in ins]
        out = GaussianRasterizer(product_settings(S, dev))(gin[0], gin[1], gin[2], gin[3], None, gin[4], gin[5], gin[6], None)
    torch.cuda.synchronize()
    check_overflow()
    return ref, [o.cpu() for o in out]


def _wall(z, half, step, quat, opacity, size=0.03, conf=None, gen=None):
    """fronto-parallel grid of surfels on the plane z: (-half..half)^2, spacing `step`"""
    xs = torch.arange(-half, half + 1e-6, step)
    gx, gy = torch.meshgrid(xs, xs, indexing="ij")
    n = gx.numel()
    means = torch.stack([gx.reshape(-1), gy.reshape(-1), torch.full((n,), float(z))], -1)
    scales = torch.cat([torch.full((n, 2), size), torch.zeros(n, 1)], 1)
    return dict(means=means, scales=scales, rotations=quat[None].repeat(n, 1), opacities=torch.full((n,), float(opacity)),
                colors=torch.rand(n, 3, generator=gen) if gen is not None else torch.full((n, 3), 0.5),
                confidences=torch.rand(n, generator=gen) if conf is None else torch.full((n,), float(conf)))


def _cat(*parts):
    return {k: torch.cat([p[k] for p in parts], 0) for k in parts[0]}


def test_confidence_image_in_unit_interval_and_empty_pixels_read_as_unseen(agslib):
    """confidence.py:47-101: `uncertainty = 1 - confidences`; `depth < 0.001` -> unseen surface.  Left half of the
    image: a dense wall of surfels with per-surfel confidences in [0, 1] (several deep, so the blend saturates);
    right half: nothing."""
    gen = torch.Generator().manual_seed(3)
    h, w = 96, 128
    layers = [_wall(1.5 + 0.2 * k, 0.9, 0.03, FRONT, 0.9, gen=gen) for k in range(4)]
    a = _cat(*layers)
    keep = a["means"][:, 0] < -0.1                                    # left half only
    a = {k: v[keep] for k, v in a.items()}
    a["confidences"][::3] = 1.0                                        # the extremes of the interval are reached
    a["confidences"][1::3] = 0.0
    S = _settings(h, w, (1, 1, 1, 0, 0))
    ref, out = _both(a, S)
    for name, o in (("oracle", ref), ("hip", out)):
        conf, depth, opac = o[4][0], o[2][0], o[3][0]
        assert float(conf.min()) >= 0.0 and float(conf.max()) <= 1.0 + 1e-6, (name, float(conf.min()), float(conf.max()))
        empty = opac == 0
        assert empty[:, w // 2 + 8:].all() and not empty[24:72, 24:48].any(), name   # (the wall ends short of the image's edges)
        assert float(depth[empty].abs().max()) == 0.0 and float(conf[empty].abs().max()) == 0.0, name   # < 0.001: "unseen"
        seen = opac > 0.5
        assert seen.any() and float(depth[seen].min()) > 1.0, name       # a blended pixel reads a real depth, never "unseen"
    assert float((out[4] - ref[4]).abs().max()) < 1e-5


def test_count_says_visible_in_the_newest_view(agslib):
    """gaussian_map.py:183-195,229-232: post-processing renders with require_importance, front_only and
    render_masks = (depth_gt > 0) and reads ONLY `counts`: counts[-1] >= 1 <=> the surfel is visible in that view.
    Constructed: [0] a large opaque surfel facing the camera; [1] a small one hidden behind it (on the same ray); [2] a small one beside
    it, in the open; [3] one in the open that faces AWAY (front_only culls it); [4] one in the open but where the
    render mask is 0 (no depth measurement there); [5] one behind the camera."""
    h, w = 96, 96
    means = torch.tensor([[-0.5, 0.0, 1.0], [-1.0, 0.0, 2.0], [0.5, -0.4, 2.0], [0.5, 0.4, 2.0], [0.0, 0.9, 2.0], [0.0, 0.0, -1.0]])
    scales = torch.tensor([[0.3, 0.3, 0], [0.02, 0.02, 0], [0.05, 0.05, 0], [0.05, 0.05, 0], [0.05, 0.05, 0], [0.05, 0.05, 0]])
    rot = torch.stack([FRONT, FRONT, FRONT, BACK, FRONT, FRONT])
    a = dict(means=means, scales=scales, rotations=rot, opacities=torch.ones(6), colors=torch.full((6, 3), 0.5),
             confidences=torch.ones(6))
    mask = torch.ones(1, h, w)
    # surfel 4 projects to (x, y) = (0, 0.9) / 2 / 0.8 -> ndc y = 0.5625 -> row ~ 74: mask out the bottom rows
    mask[:, 64:, :] = 0.0
    S = _settings(h, w, (1, 1, 1, 1, 1), mask=mask)
    ref, out = _both(a, S)
    for name, o in (("oracle", ref), ("hip", out)):
        count, radii = o[6], o[7]
        visible = count >= 1
        assert visible.tolist() == [True, False, True, False, False, False], (name, count.tolist())
        assert int(radii[3]) == 0 and int(radii[5]) == 0 and int(radii[4]) > 0, (name, radii.tolist())   # culled vs merely masked
        assert int(count[0]) > 500, name                                   # the wall fills a good part of the image
    assert out[6].tolist() == ref[6].tolist()
    # without front_only the back-facing surfel is rendered (flipped, D11) and counted; without the mask surfel 4 too
    S2 = _settings(h, w, (1, 1, 1, 1, 0))
    ref2, out2 = _both(a, S2)
    assert (out2[6] >= 1).tolist() == (ref2[6] >= 1).tolist() == [True, False, True, True, True, False]
    # config[3] == 0: no statistics at all
    S3 = _settings(h, w, (1, 1, 1, 0, 1), mask=mask)
    ref3, out3 = _both(a, S3)
    assert int(out3[6].abs().sum()) == 0 and float(out3[5].abs().sum()) == 0.0


def test_opacity_and_depth_rules_of_the_growth_mask(agslib):
    """gaussian_map.py:482-485 (cal_mask): a pixel spawns new surfels if the rendered opacity is < 0.5 or the
    rendered depth lies more than 5 % in FRONT of the measurement.  A dense opaque wall at z = 2 must therefore render
    opacity >= 0.5 and depth within 5 % of 2 over its interior (no spawning on a mapped surface), and the uncovered part
    of the image opacity 0 (spawn)."""
    h, w = 96, 128
    a = _wall(2.0, 0.8, 0.025, FRONT, 0.95, size=0.03, conf=0.7)
    S = _settings(h, w, (1, 1, 1, 0, 0))
    ref, out = _both(a, S)
    fx = w / (2 * 0.8); fy = h / (2 * 0.8)
    ys, xs = torch.meshgrid(torch.arange(h).float(), torch.arange(w).float(), indexing="ij")
    X = (xs + 0.5 - w / 2) / fx * 2.0; Y = (ys + 0.5 - h / 2) / fy * 2.0          # where the pixel's ray meets z = 2
    inside = (X.abs() < 0.7) & (Y.abs() < 0.7)
    outside = (X.abs() > 0.95) | (Y.abs() > 0.95)
    depth_gt = torch.full((h, w), 2.0)
    for name, o in (("oracle", ref), ("hip", out)):
        depth, opac = o[2][0], o[3][0]
        assert float(opac[inside].min()) >= 0.5, (name, float(opac[inside].min()))
        in_front = (depth_gt - depth) < -0.05 * depth_gt
        assert not in_front[inside].any(), name
        assert float((depth[inside] - 2.0).abs().max()) < 0.05 * 2.0, name
        assert float(opac[outside].max()) < 0.5 and float(opac[outside].max()) == 0.0, name
    assert float((out[3] - ref[3]).abs().max()) < 1e-5 and float((out[2] - ref[2]).abs().max()) < 1e-4
