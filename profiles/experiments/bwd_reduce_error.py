"""The blend backward's reduction forms (AgsTuning.bwd_reduce: f32 | bf16 | bf16x3 | valu) against fp64 gradients.

For every scene: the CPU oracle is run in float64 (same decisions D1-D12, double arithmetic throughout) over a set of
tiles with seeded image gradients restricted to those tiles; the HIP path then runs forward + backward on the same
inputs once per form (the exact-f32 form twice: its atomics reorder the sums from run to run - that spread is the
yardstick).  Per gradient array: relative L1 distance to the fp64 gradients, and relative L1 distance to the first f32
run.  Writes one JSON object per scene to stdout and a markdown table to the path given as argv[1] (optional).

usage: python profiles/experiments/bwd_reduce_error.py [out.md] [--quick]"""
import copy
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from _scenes import oracle_inputs, oracle_on_tiles, room_case  # noqa: E402
from active_gs_amd import _lib, raster_api as api  # noqa: E402

dev = torch.device("cuda:0")
FIELDS = ("means3D", "scales", "rotations", "opacities", "colors")
ORACLE_IDX = {"means3D": 0, "opacities": 2, "colors": 4, "scales": 5, "rotations": 6}
MODES = ("f32", "f32", "bf16x3", "bf16", "valu")


def to64(S):
    S64 = copy.copy(S)
    for k, v in vars(S64).items():
        if torch.is_tensor(v) and v.is_floating_point():
            setattr(S64, k, v.double())
    return S64


def run_case(label, n, h, w, view, seed, scale_mult, max_tiles, fullest, focal=None):
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    a, S = room_case(n, h, w, view=view, seed=seed, scale_mult=scale_mult, focal_px=focal)
    ins = oracle_inputs(a)
    gen = torch.Generator().manual_seed(11)
    d_img = [torch.randn(c, h, w, generator=gen) / (h * w) for c in (3, 3, 1, 1, 1)]
    ins64 = [t.detach().double().requires_grad_(t.requires_grad) for t in ins]
    t0 = time.time()
    _, covered, aux = oracle_on_tiles(ins64, to64(S), [g.double() for g in d_img], max_tiles=max_tiles, fullest=fullest)
    oracle_s = time.time() - t0
    ref = {k: ins64[i].grad.reshape(n, -1) for k, i in ORACLE_IDX.items()}
    gin = [t.detach().to(dev) for t in ins]
    g = api.Gaussians(gin[0], gin[5].contiguous(), gin[6], gin[2].reshape(-1).contiguous(), gin[4], gin[3])
    cam = api.Camera(h, w, S.tanfovx, S.tanfovy, S.viewmatrix.to(dev), S.projmatrix.to(dev), S.bg.to(dev))
    m = covered.to(dev)
    d_dev = [(t.to(dev) * m).contiguous() for t in d_img]
    runs = []
    for mode in MODES:
        st = api.alloc_state(n, h, w, 1 << 24, dev, tuning=_lib.make_tuning(bwd_reduce=mode))
        api.forward(cam, g, st)
        gr = api.backward(cam, g, st, *d_dev)
        torch.cuda.synchronize()
        assert not api.read_status(st)["overflow"]
        runs.append({k: getattr(gr, k).double().cpu().reshape(n, -1) for k in FIELDS})
        del st
    rec = dict(scene=label, surfels=n, image=[h, w], tiles_compared=len(aux["tiles"]), tiles_nonempty=aux["nonempty"],
               instances=aux["instances"], oracle_fp64_seconds=round(oracle_s, 1), forms={})
    names = ("f32", "f32 (second run)", "bf16x3", "bf16", "valu")
    for name, r in zip(names, runs):
        rec["forms"][name] = dict(
            vs_fp64={k: float((r[k] - ref[k]).abs().sum() / ref[k].abs().sum()) for k in FIELDS},
            vs_f32_run1={k: float((r[k] - runs[0][k]).abs().sum() / runs[0][k].abs().sum()) for k in FIELDS})
    print(json.dumps(rec), flush=True)
    return rec


def main():
    out_md = next((a for a in sys.argv[1:] if not a.startswith("--")), None)
    quick = "--quick" in sys.argv
    cases = [("fixture c1: 5 k surfels 300x170", 5000, 170, 300, 1, 1, 2.0, None, 0, None),
             ("fixture: 2 k surfels 128x96 x3 scales", 2000, 96, 128, 0, 3, 3.0, None, 0, None),
             ("fixture: 20 k surfels 240x136", 20000, 136, 240, 2, 8, 2.0, None, 0, None)]
    if not quick:
        cases += [("C2: 200 k surfels 1200x680, 400 spread tiles + 12 fullest", 200_000, 680, 1200, 0, 0, 1.0, 400, 12, None),
                  ("C4 share: 1.5 M surfels 1200x680 x1.5, 120 tiles + 12 fullest", 1_500_000, 680, 1200, 1, 0, 1.5, 120, 12, None),
                  ("C5: 5 M surfels 2048x2048, 100 tiles + 10 fullest", 5_000_000, 2048, 2048, 0, 0, 1.0, 100, 10, None)]
    recs = [run_case(*c) for c in cases]
    if out_md:
        with open(out_md, "w") as f:
            f.write("| scene | form | " + " | ".join(f"{k} vs fp64" for k in FIELDS) + " | " + " | ".join(f"{k} vs f32 run 1" for k in FIELDS) + " |\n")
            f.write("|---|---|" + "---:|" * (2 * len(FIELDS)) + "\n")
            for r in recs:
                for name, e in r["forms"].items():
                    f.write(f"| {r['scene']} ({r['tiles_compared']} of {r['tiles_nonempty']} tiles) | {name} | "
                            + " | ".join(f"{e['vs_fp64'][k]:.2e}" for k in FIELDS) + " | "
                            + " | ".join(f"{e['vs_f32_run1'][k]:.2e}" for k in FIELDS) + " |\n")


if __name__ == "__main__":
    main()
