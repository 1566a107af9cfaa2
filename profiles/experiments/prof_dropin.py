import sys, os, cProfile, pstats, time
sys.path.insert(0, os.getcwd())
import torch
from active_gs_amd.camera import camera_matrices
from active_gs_amd.synthetic import activate, make_camera, make_room_scene
from diff_gaussian_rasterization_2d import GaussianRasterizationSettings, GaussianRasterizer
dev = torch.device("cuda:0"); n, h, w, views = 200_000, 512, 512, 8
raw = {k: v.to(dev) for k, v in make_room_scene(n, seed=0).items()}
c2w, K = zip(*[make_camera(v, h, w, focal_px=0.5 * 512 / 0.57735) for v in range(views)])
cm0 = camera_matrices(torch.stack(c2w), torch.stack(K), 0.001, 10.0)
a = activate(raw)
leaves = [a["means"].clone().requires_grad_(True), torch.zeros(n, 3, device=dev, requires_grad=True),
          a["opacities"][:, None].clone().requires_grad_(True), a["confidences"], a["colors"].clone().requires_grad_(True),
          a["scales"].clone().requires_grad_(True), a["rotations"].clone().requires_grad_(True)]
gen = torch.Generator().manual_seed(0)
gimg = [torch.randn(c, h, w, generator=gen).to(dev) / (h * w) for c in (3, 3, 1, 1, 1)]
settings = [GaussianRasterizationSettings(image_height=h, image_width=w, tanfovx=float(cm0["tanfov"][v, 0]), tanfovy=float(cm0["tanfov"][v, 1]),
    bg=torch.zeros(4, device=dev), scale_modifier=1.0, viewmatrix=cm0["viewmatrix"][v].to(dev), projmatrix=cm0["projmatrix"][v].to(dev),
    sh_degree=0, campos=cm0["campos"][v].to(dev), prefiltered=False, render_mask=torch.tensor([], device=dev), weight_thres=0.03,
    debug=False, config=torch.tensor([1.0, 1, 1, 0, 0]).to(dev)) for v in range(views)]
def it():
    outs = [GaussianRasterizer(s)(leaves[0], leaves[1], leaves[2], leaves[3], None, leaves[4], leaves[5], leaves[6], None) for s in settings]
    torch.autograd.backward([o[k] for o in outs for k in range(5)], [gimg[k] for _ in outs for k in range(5)])
    for t in leaves: t.grad = None
for _ in range(5): it()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(20): it()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
