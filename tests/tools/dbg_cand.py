import sys, os, math, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
from active_gs_amd import densify
from oracle import densify_oracle as dor
DEV = torch.device("cuda:0")
g = torch.load("tests/golden/densify.pt")
pred_ref = g["second"]["pred"]
preds = [None, dict(rgb=pred_ref["rgb"][0], depth=pred_ref["depth"][0], opacity=pred_ref["opacity"][0])]
for fi, (frame, pred) in enumerate(zip(g["frames"], preds)):
    ds = torch.from_numpy(dor.smooth_depth(frame["depth"][0].numpy()))[None]
    ref = dor.candidates(frame["rgb"], frame["depth"], frame["intrinsic"], frame["extrinsic"], ds, pred, g["error_thres"])
    todev = lambda d: {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in d.items()}
    out = densify.candidates(todev(frame), ds.to(DEV), None if pred is None else todev(pred), g["error_thres"])
    sel = out["select"].cpu().bool()
    bad = torch.nonzero(sel != ref["select"]).flatten()
    H, W = frame["rgb"].shape[-2:]
    print("frame", fi, "mismatch", bad.numel())
    R = frame["extrinsic"][:3, :3]
    # recompute oracle intermediates
    P = H * W
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    uv1 = torch.stack([(xs.float() + 0.5) / W, (ys.float() + 0.5) / H, torch.ones(H, W)], -1).view(P, 3)
    dir_w = (uv1 @ frame["intrinsic"].inverse().T) @ R.T
    cos = (torch.nn.functional.normalize(dir_w, dim=1) * ref["normals"]).sum(-1)
    for i in bad.tolist():
        y, x = divmod(i, W)
        extra = ""
        if pred is not None:
            err = ((frame["rgb"] - pred["rgb"]) ** 2).mean(0).view(-1)[i]
            extra = f" err={float(err):.4f} op={float(pred['opacity'].view(-1)[i]):.4f} dd={float(frame['depth'].view(-1)[i]-pred['depth'].view(-1)[i]):.4f}"
        print(f"  px({x},{y}) gpu={int(sel[i])} ref={int(ref['select'][i])} depth={float(frame['depth'].view(-1)[i]):.4f} cos={float(cos[i]):.5f} n={ref['normals'][i].tolist()}{extra}")
