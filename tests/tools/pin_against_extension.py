#!/usr/bin/env python3
"""Pin kit: feed this repository's committed oracle scenes through the REAL CUDA extension.

ActiveGS runs ``git+https://github.com/liren-jin/diff-gaussian-rasterization_2d`` (envs/requirements.txt:15 of the
reference, no pin, CUDA only).  Neither its source nor a wheel exists in the build container, so the oracle
(oracle/surfel_oracle.py) restates the PUBLISHED algorithm and its decisions D5-D11 are tied to
nothing but that.  Whoever has a machine with the extension installed (any CUDA GPU) closes the gap with this script:

    python tests/tools/pin_against_extension.py            # writes tests/golden/extension_small.pt, extension_c1.pt

It rebuilds the seeded scenes of tests/golden/oracle_small.pt / oracle_c1.pt (tensors + the seeded camera; checked
bit for bit against the committed inputs), calls the extension exactly as /root/reference/utils/operations.py:682-713
does - once per CONFIG VARIANT below, each of which isolates one oracle decision - and stores inputs, outputs and
gradients.  ``tests/test_cpu_oracle.py::test_oracle_against_the_real_extension_when_present`` then compares the oracle
with those files and names every decision that differs.  Nothing of the reference travels: the files hold tensors the
extension computed from this repository's own scenes.

The repository ships a module of the SAME name (the drop-in boundary): this script refuses to "pin" against it."""
import argparse
import importlib
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
GOLD = os.path.join(ROOT, "tests", "golden")

# variant -> (config [c0, normalize_depth?, per-pixel depth?, importance?, front_only?], render mask?, decisions it isolates)
VARIANTS = {
    "base":         ((1, 1, 1, 0, 0), False, ["D1-D4 geometry/alpha rule", "D7 rgb + background", "D8 confidence image", "D11 normal", "D12 radii"]),
    "config0_off":  ((0, 1, 1, 0, 0), False, ["config[0] (meaning unknown to the oracle: it ignores it)"]),
    "no_normalise": ((1, 0, 1, 0, 0), False, ["D6 depth / normal normalisation by accumulated opacity (config[1])"]),
    "centre_depth": ((1, 1, 0, 0, 0), False, ["D5 per-pixel ray-plane depth vs centre depth (config[2])"]),
    "stats":        ((1, 1, 1, 1, 0), False, ["D9 importance = sum alpha*T, count = #pixels with alpha*T > weight_thres"]),
    "stats_masked": ((1, 1, 1, 1, 0), True,  ["D9 render_mask restricts importance / count"]),
    "front_only":   ((1, 1, 1, 1, 1), False, ["D10 front_only skips surfels facing away"]),
}


def import_real_extension():
    """the installed wheel, not this repository's drop-in module of the same name"""
    clean = [p for p in sys.path if os.path.abspath(p or os.getcwd()) != ROOT]
    saved, sys.path[:] = list(sys.path), clean
    try:
        sys.modules.pop("diff_gaussian_rasterization_2d", None)
        mod = importlib.import_module("diff_gaussian_rasterization_2d")
    except ModuleNotFoundError:
        raise SystemExit("diff_gaussian_rasterization_2d (the CUDA extension) is not installed here: "
                         "pip install git+https://github.com/liren-jin/diff-gaussian-rasterization_2d on a CUDA machine")
    finally:
        sys.path[:] = saved
    where = os.path.abspath(getattr(mod, "__file__", "") or "")
    if where.startswith(ROOT) or hasattr(mod, "check_overflow"):
        raise SystemExit(f"{where} is this repository's drop-in module: install the CUDA extension "
                         "(pip install git+https://github.com/liren-jin/diff-gaussian-rasterization_2d) and run again")
    return mod


def run_variant(ext, ins, S, config, mask, image_grads, dev):
    """one call of the extension, shaped like operations.py:682-713 -> (8 outputs, 6 gradients or None)"""
    cfg = torch.tensor([float(c) for c in config]).to(dev)
    settings = ext.GaussianRasterizationSettings(
        image_height=S.image_height, image_width=S.image_width, tanfovx=S.tanfovx, tanfovy=S.tanfovy, bg=S.bg.to(dev),
        scale_modifier=1.0, viewmatrix=S.viewmatrix.to(dev), projmatrix=S.projmatrix.to(dev), sh_degree=0,
        campos=S.campos.to(dev), prefiltered=False, render_mask=(mask if mask is not None else torch.tensor([])).to(dev),
        weight_thres=0.03, debug=False, config=cfg)
    g = [t.detach().clone().to(dev) for t in ins]
    for i in (0, 1, 2, 4, 5, 6):
        g[i].requires_grad_(True)
    out = ext.GaussianRasterizer(settings)(means3D=g[0], means2D=g[1], opacities=g[2], confidences=g[3], shs=None,
                                           colors_precomp=g[4], scales=g[5], rotations=g[6], cov3D_precomp=None)
    grads = None
    if image_grads is not None:
        sum((o * w.to(dev)).sum() for o, w in zip(out[:5], image_grads)).backward()
        grads = {i: g[i].grad.detach().cpu().clone() for i in (0, 1, 2, 4, 5, 6) if g[i].grad is not None}
    return [o.detach().cpu().clone() for o in out], grads


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--out", default=GOLD)
    args = ap.parse_args()
    ext = import_real_extension()
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.append(p)                 # (behind site-packages; the extension is already in sys.modules)
    from _scenes import oracle_inputs, room_case
    dev = torch.device(args.device)
    for tag in ("small", "c1"):
        d = torch.load(os.path.join(GOLD, f"oracle_{tag}.pt"))
        a, S = room_case(d["n"], d["h"], d["w"], view=d["view"], seed=d["seed"], scale_mult=d["mult"], config=d["config"])
        ins = oracle_inputs(a, requires_grad=False)
        for x, y in zip(ins, d["inputs"]):
            assert torch.equal(x, y), "the seeded scene generator drifted: regenerate tests/golden first"
        gen = torch.Generator().manual_seed(17)
        mask = (torch.rand(1, d["h"], d["w"], generator=gen) > 0.4).float()
        record = dict(n=d["n"], h=d["h"], w=d["w"], view=d["view"], seed=d["seed"], mult=d["mult"], inputs=d["inputs"],
                      image_grads=d["image_grads"], mask=mask, variants={},
                      extension=dict(file=str(getattr(ext, "__file__", "?")), version=str(getattr(ext, "__version__", "?")),
                                     torch=str(torch.__version__), device=torch.cuda.get_device_name(dev) if dev.type == "cuda" else "cpu"))
        for name, (config, use_mask, decisions) in VARIANTS.items():
            outs, grads = run_variant(ext, ins, S, config, mask if use_mask else None, d["image_grads"], dev)
            record["variants"][name] = dict(config=list(config), masked=use_mask, decisions=decisions, outputs=outs, grads=grads)
            print(f"{tag}/{name}: rgb mean {float(outs[0].mean()):.5f}, visible {int((outs[7] > 0).sum())}, "
                  f"count sum {int(outs[6].sum())}")
        path = os.path.join(args.out, f"extension_{tag}.pt")
        torch.save(record, path)
        print("wrote", path)
    print("now run: python -m pytest tests/test_cpu_oracle.py -k real_extension -q")


if __name__ == "__main__":
    main()
