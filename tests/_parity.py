"""Parity gates shared by the GPU tests (and mirrored by bench.py's CPU leg): HIP path vs the oracle.

Three layers, all asserted:
  * the CONTRACT tolerances of BASELINE.json: mean L1 over the compared pixels <= 1e-4 per image (depth, in metres:
    1e-3), relative L1 <= 1e-3 per gradient;
  * SHAPE-OF-ERROR gates a mean cannot see: the largest error of any compared pixel and the mean L1 of the WORST 16x16
    tile (one wholly wrong tile in an 816 k-pixel image moves the image mean by 3e-5 - it must not pass);
  * REGRESSION gates at ~10x what the kernels measure today (see profiles/r03_parity_margins.json), so that a change
    that costs an order of magnitude of accuracy fails although it is still inside the contract.
Integer outputs are held EXACT except where a float sits on a rounding boundary, and every such row / pixel is
counted, explained (``radii_report``) and logged.

``AGS_PARITY_LOG=<file>`` appends one JSON line per comparison (what the margins file is made from)."""
import json
import os

import torch

IMAGES = ("rgb", "normal", "depth", "opacity", "confidence")
MEAN_L1 = {"rgb": 1e-4, "normal": 1e-4, "depth": 1e-3, "opacity": 1e-4, "confidence": 1e-4}       # contract
GRAD_REL = 1e-3                                                                                    # contract
# Any one pixel: the blend rule is discontinuous - a surfel whose alpha sits at the 1/255 cut (D4) is taken by one side
# and skipped by the other, which moves a unit-range channel by up to alpha * T <= 1/255 = 3.9e-3 (measured on every
# scene of the suite: <= 2.5e-3; depth, in metres: <= 5.8e-3).  The gate sits just above that bound; how MANY pixels
# may be off by more than rounding is gated separately (OUTLIER_*).
MAX_ABS = {"rgb": 4.5e-3, "normal": 4.5e-3, "depth": 3e-2, "opacity": 4.5e-3, "confidence": 4.5e-3}
OUTLIER_ABS = {"rgb": 2.5e-4, "normal": 2.5e-4, "depth": 2.5e-3, "opacity": 2.5e-4, "confidence": 2.5e-4}
OUTLIER_FRAC = 1e-3            # fraction of the compared pixels that may be off by more than OUTLIER_ABS
# the worst 16x16 tile (measured <= 8.1e-6, depth 2.3e-5)
TILE_L1 = {"rgb": 5e-5, "normal": 5e-5, "depth": 2.5e-4, "opacity": 5e-5, "confidence": 5e-5}
# ~10x what the kernels measure (profiles/r03_parity_margins.json: mean L1 <= 8.8e-7, depth 1.9e-6; gradients <= 4.7e-5)
REG_MEAN_L1 = {"rgb": 1e-5, "normal": 1e-5, "depth": 2e-5, "opacity": 1e-5, "confidence": 1e-5}
REG_GRAD_REL = 3e-4


def _log(kind, payload):
    path = os.environ.get("AGS_PARITY_LOG")
    if path:
        rec = {"test": os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0], "kind": kind, **payload}
        with open(path, "a") as f:
            f.write(json.dumps(rec) + "\n")


def image_stats(ref, out, covered=None):
    """ref / out: (C,H,W) CPU tensors; covered: (H,W) {0,1} mask of the pixels that were compared (None = all).
    -> dict(mean, max, tile): mean L1, largest error, mean L1 of the worst 16x16 tile (over its compared pixels)."""
    C, H, W = ref.shape
    d = (out.detach().cpu().double() - ref.detach().cpu().double()).abs()
    m = torch.ones(H, W, dtype=torch.float64) if covered is None else covered.cpu().double()
    d = d * m
    npx = float(m.sum())
    Hp, Wp = (H + 15) // 16 * 16, (W + 15) // 16 * 16
    dp = torch.zeros(C, Hp, Wp, dtype=torch.float64); dp[:, :H, :W] = d
    mp = torch.zeros(Hp, Wp, dtype=torch.float64); mp[:H, :W] = m
    ts = dp.reshape(C, Hp // 16, 16, Wp // 16, 16).sum((0, 2, 4))
    tc = mp.reshape(Hp // 16, 16, Wp // 16, 16).sum((1, 3)) * C
    tile = torch.where(tc > 0, ts / tc.clamp_min(1), torch.zeros_like(ts))
    return dict(mean=float(d.sum() / max(npx * C, 1)), max=float(d.max()), tile=float(tile.max()), npx=npx,
                per_pixel_max=d.amax(0))


def outlier_frac(stats, k):
    return float((stats["per_pixel_max"] > OUTLIER_ABS[k]).sum()) / max(stats["npx"], 1.0)


def check_images(ref, out, covered=None, names=IMAGES, regression=True, what=""):
    """ref / out: dicts or sequences of the five images in the order of ``names``.  Asserts all three layers."""
    stats = {}
    for i, k in enumerate(names):
        r = ref[k] if isinstance(ref, dict) else ref[i]
        o = out[k] if isinstance(out, dict) else out[i]
        assert tuple(o.shape) == tuple(r.shape) and o.dtype == torch.float32, (k, o.shape, r.shape, o.dtype)
        stats[k] = image_stats(r, o, covered)
        stats[k]["outliers"] = outlier_frac(stats[k], k)
        del stats[k]["per_pixel_max"]
    _log("images", {"what": what, "stats": stats})
    for k, s in stats.items():
        assert s["mean"] < MEAN_L1[k], f"{k}: mean L1 {s['mean']:.3g} exceeds the contract tolerance {MEAN_L1[k]}"
        assert s["max"] < MAX_ABS[k], f"{k}: a pixel is off by {s['max']:.3g} (gate {MAX_ABS[k]})"
        assert s["outliers"] < OUTLIER_FRAC, f"{k}: {s['outliers']:.3g} of the pixels are off by more than {OUTLIER_ABS[k]} (gate {OUTLIER_FRAC})"
        assert s["tile"] < TILE_L1[k], f"{k}: the worst 16x16 tile has mean L1 {s['tile']:.3g} (gate {TILE_L1[k]})"
        if regression:
            assert s["mean"] < REG_MEAN_L1[k], f"{k}: mean L1 {s['mean']:.3g} is >10x what the kernels measured ({REG_MEAN_L1[k]})"
    return stats


def grad_stats(ref_grads, out_grads):
    """dicts name -> tensor; -> name -> relative L1."""
    rel = {}
    for k, r in ref_grads.items():
        o = out_grads[k].detach().cpu().double().reshape(-1)
        r = r.detach().cpu().double().reshape(-1)
        rel[k] = float((o - r).abs().sum() / r.abs().sum().clamp_min(1e-30))
    return rel


def check_grads(ref_grads, out_grads, regression=True, what=""):
    rel = grad_stats(ref_grads, out_grads)
    _log("grads", {"what": what, "rel_L1": rel})
    for k, v in rel.items():
        assert v < GRAD_REL, f"d_{k}: relative L1 {v:.3g} exceeds the contract tolerance {GRAD_REL}"
        if regression:
            assert v < REG_GRAD_REL, f"d_{k}: relative L1 {v:.3g} is >10x what the kernels measured ({REG_GRAD_REL})"
    return rel


def radii_report(radii_hip, G, ins, S, what=""):
    """``radii`` is an integer output: exact, except rows whose float sits on a rounding boundary.  Every mismatching row
    is re-evaluated with the oracle's per-Gaussian stage in fp64 and must be EXPLAINED by one of
      * ceil: 3 sqrt(lambda_max) lies within the fp32 error bound of an integer and the two radii differ by 1
        (lambda_max = mid + sqrt(max(0.1, mid^2 - det)): the cancellation in mid^2 - det is what the bound carries);
      * rect: the surfel is visible on one side only and its tile rect flips between empty and non-empty when mean /
        radius move by their fp32 error (the truncations of D3), or det / the facing test sit at zero.
    Returns the counts; asserts that nothing is unexplained."""
    from oracle.surfel_oracle import OracleSettings, preprocess
    ref = G["radii"]
    hip = radii_hip.detach().cpu().to(torch.int32)
    rows = torch.nonzero(hip != ref).flatten()
    rep = dict(rows=int(ref.numel()), mismatch=int(rows.numel()), ceil=0, rect=0, unexplained=0, examples=[])
    if rows.numel():
        ins64 = [t.detach().double()[rows] if t.dim() and t.shape[0] == ref.numel() else t.detach().double() for t in ins]
        S64 = OracleSettings(S.image_height, S.image_width, S.tanfovx, S.tanfovy, S.bg.double(), S.scale_modifier,
                             S.viewmatrix.double(), S.projmatrix.double(), campos=S.campos, render_mask=S.render_mask,
                             weight_thres=S.weight_thres, config=S.config)
        with torch.no_grad():
            G64 = preprocess(*ins64, S64)
        dg = G64.get("diag")
        kept = {int(k): j for j, k in enumerate(dg["keep"].tolist())} if dg is not None else {}
        gx, gy = G64["grid"]
        eps = 2.0 ** -23
        for j, row in enumerate(rows.tolist()):
            a, b = int(hip[row]), int(ref[row])
            why = None
            if j in kept:
                q = kept[j]
                x, mx, my = float(dg["rad_raw"][q]), float(dg["mx"][q]), float(dg["my"][q])
                lam, mid = (x / 3.0) ** 2, float(dg["mid"][q])
                # fp32 error bound of x = 3 sqrt(lam), lam = mid + sq, sq = sqrt(max(0.1, mid^2 - det)) >= 0.316: the terms
                # of mid and det carry ~16 eps relative (a dozen roundings, the kernels' 2.5-ulp sqrt / divide); the
                # difference mid^2 - det has absolute error ~3 mid^2 of that, amplified by 1 / (2 sq) in the root
                sq = max(lam - mid, 0.316)
                tol = x / (2.0 * lam) * 16 * eps * (mid + 1.5 * mid * mid / sq) + 4 * eps * x
                if a > 0 and b > 0 and abs(a - b) == 1 and abs(x - round(x)) <= tol:
                    why = "ceil"
                elif (a == 0) != (b == 0):
                    # visible on one side only: does the tile rect flip under the floats' error?
                    dm = 64 * eps * (abs(mx) + abs(my) + x + 16.0)
                    cnt = set()
                    for r in (max(a, b), max(a, b) - 1, max(a, b) + 1):
                        for sx in (-dm, 0.0, dm):
                            for sy in (-dm, 0.0, dm):
                                x0 = min(max(int((mx + sx - r) / 16), 0), gx); x1 = min(max(int((mx + sx + r + 15) / 16), 0), gx)
                                y0 = min(max(int((my + sy - r) / 16), 0), gy); y1 = min(max(int((my + sy + r + 15) / 16), 0), gy)
                                cnt.add((x1 - x0) * (y1 - y0) > 0)
                    if len(cnt) == 2 or abs(float(dg["det"][q])) < 1e-3 or abs(float(dg["dotnc"][q])) < 1e-6:
                        why = "rect"
            if why is None:
                rep["unexplained"] += 1
                if len(rep["examples"]) < 8:
                    rep["examples"].append(dict(row=row, hip=a, oracle=b))
            else:
                rep[why] += 1
    _log("radii", {"what": what, **rep})
    assert rep["unexplained"] == 0, f"radii differ on rows no rounding boundary explains: {rep}"
    assert rep["mismatch"] <= max(2, 1e-4 * rep["rows"]), rep
    return rep


def count_report(count_hip, count_ref, what="", max_row_diff=2, max_rows_frac=2e-3):
    """``count`` (pixels whose blend weight exceeds weight_thres, per surfel) is an integer output: a row differs only
    where a pixel's weight sits within rounding of the threshold.  Logged; gated on the per-row difference and on the
    fraction of rows that differ."""
    a, b = count_hip.detach().cpu().long(), count_ref.detach().cpu().long()
    d = (a - b).abs()
    rep = dict(rows=int(a.numel()), rows_differ=int((d > 0).sum()), max_row_diff=int(d.max()) if d.numel() else 0,
               total=int(b.sum()), total_diff=int(d.sum()))
    _log("count", {"what": what, **rep})
    assert rep["max_row_diff"] <= max_row_diff, rep
    assert rep["rows_differ"] <= max(2, max_rows_frac * rep["rows"]), rep
    return rep


def oracle_last_contributor(aux, n_contrib, H, W):
    """(H,W) int64 surfel id of the last surfel each pixel blended in the oracle (-1: none), from its n_contrib image
    (1-based list positions), tile ranges and sorted list (oracle_on_tiles / rasterize(return_aux=True))."""
    G, so, ranges = aux["G"], aux["sorted_owner"], aux["ranges"]
    tx = (W + 15) // 16
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    tile = (ys // 16) * tx + xs // 16
    last = n_contrib.long()
    pos = (ranges[tile, 0] + last - 1).clamp(0, max(so.numel() - 1, 0))
    if so.numel() == 0:
        return torch.full((H, W), -1, dtype=torch.long)
    return torch.where(last > 0, G["vis"][so[pos]], torch.full_like(last, -1))


def last_contributor_report(hip_last, ref_last, covered=None, what="", max_frac=2e-5):
    """The id of the last surfel every pixel blended is an integer output: exact, except pixels whose transmittance
    sits within rounding of the 1e-4 stop (or whose last alpha sits at 1/255).  Counted, logged, gated."""
    a, b = hip_last.detach().cpu().long(), ref_last.detach().cpu().long()
    m = torch.ones_like(a, dtype=torch.bool) if covered is None else covered.cpu() > 0
    diff = (a != b) & m
    rep = dict(pixels=int(m.sum()), differ=int(diff.sum()))
    _log("last_contributor", {"what": what, **rep})
    assert rep["differ"] <= max(2, max_frac * rep["pixels"]), rep
    return rep
