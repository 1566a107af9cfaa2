"""Shared seeded scenes for the parity tests (oracle side on CPU, product side on GPU)."""
import torch

from active_gs_amd.camera import camera_matrices
from active_gs_amd.synthetic import activate, make_camera, make_room_scene
from oracle.surfel_oracle import OracleSettings


def room_case(n, h, w, view=0, seed=0, scale_mult=1.0, bg=(0.1, 0.2, 0.3, 0.0), config=(1, 1, 1, 0, 0),
              mask=None, focal_px=None):
    a = activate(make_room_scene(n, seed=seed))
    a["scales"] = a["scales"] * scale_mult
    c2w, K = make_camera(view, h, w, focal_px=focal_px)
    cm = camera_matrices(c2w[None], K[None], 0.001, 10.0)
    S = OracleSettings(h, w, cm["tanfov"][0, 0].item(), cm["tanfov"][0, 1].item(), torch.tensor(bg), 1.0,
                       cm["viewmatrix"][0].contiguous(), cm["projmatrix"][0].contiguous(),
                       campos=cm["campos"][0], render_mask=mask,
                       config=torch.tensor([float(c) for c in config]))
    return a, S


def oracle_inputs(a, requires_grad=True):
    n = a["means"].shape[0]
    ins = [a["means"].clone(), torch.zeros(n, 3), a["opacities"][:, None].clone(), a["confidences"].clone(),
           a["colors"].clone(), a["scales"].clone(), a["rotations"].clone()]
    if requires_grad:
        for i in (0, 1, 2, 4, 5, 6):
            ins[i].requires_grad_(True)
    return ins


def product_settings(S, dev):
    from diff_gaussian_rasterization_2d import GaussianRasterizationSettings
    mask = S.render_mask if S.render_mask is not None else torch.tensor([])
    return GaussianRasterizationSettings(
        image_height=S.image_height, image_width=S.image_width, tanfovx=S.tanfovx, tanfovy=S.tanfovy,
        bg=S.bg.to(dev), scale_modifier=S.scale_modifier, viewmatrix=S.viewmatrix.to(dev),
        projmatrix=S.projmatrix.to(dev), sh_degree=int(S.sh_degree), campos=(S.campos if S.campos is not None else torch.zeros(3)).to(dev),
        prefiltered=False, render_mask=mask.to(dev), weight_thres=S.weight_thres, debug=False,
        config=S.config.to(dev))


def oracle_on_tiles(ins, S, image_grads, tiles=None, max_tiles=None, batch=32, fullest=0):
    """Oracle forward + autograd backward over a SET of tiles (full-size scenes: the per-Gaussian stage and the
    binning run in full, the per-tile blend only where asked), with the given image gradients (rgb, normal,
    depth, opacity, confidence; None = zero) restricted to those tiles.

    tiles=None: every non-empty tile, or - with ``max_tiles`` - a strided, spatially spread subset of them plus the
    ``fullest`` longest tile lists.  Returns (images dict masked to the covered tiles, pixel mask (H,W), aux dict);
    the gradients are left in ``ins[i].grad``."""
    from oracle.surfel_oracle import bin_instances, preprocess, render_tiles
    G = preprocess(*ins, S)
    so, ranges = bin_instances(G)
    H, W = S.image_height, S.image_width
    tiles_x = (W + 15) // 16
    lens = ranges[:, 1] - ranges[:, 0]
    nonempty = torch.nonzero(lens > 0).flatten().tolist()
    if tiles is None:
        tiles = nonempty
        if max_tiles is not None and len(nonempty) > max_tiles:
            keep = set(torch.topk(lens, min(fullest, len(nonempty))).indices.tolist()) if fullest else set()
            stride = max(1, len(nonempty) // max(1, max_tiles - len(keep)))
            keep.update(nonempty[::stride])
            tiles = sorted(keep)
    covered = torch.zeros(H, W)
    names = ("rgb", "normal", "depth", "opacity", "confidence")
    images = {k: torch.zeros(3 if k in ("rgb", "normal") else 1, H, W) for k in names}
    n_contrib = torch.zeros(H, W, dtype=torch.int32)
    for b0 in range(0, len(tiles), batch):
        sample = tiles[b0:b0 + batch]
        m = torch.zeros(H, W)
        for t in sample:
            ty, tx = divmod(t, tiles_x)
            m[ty * 16:(ty + 1) * 16, tx * 16:(tx + 1) * 16] = 1.0
        R = render_tiles(G, so, ranges, S, tiles=sample)
        terms = [(R[k] * (g * m)).sum() for k, g in zip(names, image_grads) if g is not None]
        if terms:
            sum(terms).backward(retain_graph=True)
        covered += m
        for k in names:
            images[k] += R[k].detach() * m
        n_contrib += R["n_contrib"] * m.to(torch.int32)
    aux = dict(G=G, ranges=ranges, sorted_owner=so, n_contrib=n_contrib, tiles=tiles, nonempty=len(nonempty), max_list=int(lens.max()), instances=int(lens.sum()))
    return images, covered, aux
