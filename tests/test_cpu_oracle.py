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
