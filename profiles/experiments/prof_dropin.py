"""Where the drop-in module's per-call host time goes (cProfile + split timings); run on the GPU box:
python profiles/experiments/prof_dropin.py [h w views]   (also the target of the rocprofv3 --hip-trace pass)"""
import sys, os, cProfile, pstats, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from active_gs_amd import env_config
env_config.apply_env(os.environ)   # the package itself reads no environment variable
from active_gs_amd.camera import camera_matrices
from active_gs_amd.synthetic import activate, make_camera, make_room_scene
from diff_gaussian_rasterization_2d import GaussianRasterizationSettings, GaussianRasterizer
import active_gs_amd.rasterizer as R
dev = torch.device("cuda:0"); n = 200_000
h, w, views = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (512, 512, 8)
raw = {k: v.to(dev) for k, v in make_room_scene(n, seed=0).items()}
c2w, K = zip(*[make_camera(v, h, w, focal_px=(0.5 * 512 / 0.57735 if h == 512 else None)) for v in range(views)])
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
def fwd():
    return [GaussianRasterizer(s)(leaves[0], leaves[1], leaves[2], leaves[3], None, leaves[4], leaves[5], leaves[6], None) for s in settings]
def it():
    outs = fwd()
    torch.autograd.backward([o[k] for o in outs for k in range(3)], [gimg[k] for _ in outs for k in range(3)])
    for t in leaves: t.grad = None
for _ in range(5): it()
torch.cuda.synchronize()
N = 40
t0 = time.perf_counter()
with torch.no_grad():
    for _ in range(N): fwd()
t_ng = time.perf_counter() - t0; torch.cuda.synchronize(); t_ng_gpu = time.perf_counter() - t0
t0 = time.perf_counter(); tf = tb = 0.0
for _ in range(N):
    a0 = time.perf_counter(); outs = fwd(); a1 = time.perf_counter()
    torch.autograd.backward([o[k] for o in outs for k in range(3)], [gimg[k] for _ in outs for k in range(3)]); a2 = time.perf_counter()
    for t in leaves: t.grad = None
    tf += a1 - a0; tb += a2 - a1
t_all = time.perf_counter() - t0; torch.cuda.synchronize(); t_all_gpu = time.perf_counter() - t0
per = 1e6 / (N * views)
print(f"{w}x{h} x{views}: forward under no_grad: host {t_ng*per:.1f} us/view (with GPU drain {t_ng_gpu*per:.1f}); "
      f"with grad: forward host {tf*per:.1f}, backward() host {tb*per:.1f}, iteration host {t_all*per:.1f} (with GPU drain {t_all_gpu*per:.1f}) us/view")
torch.autograd.set_multithreading_enabled(False)
for _ in range(3): it()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(N): it()
t1 = time.perf_counter() - t0; torch.cuda.synchronize()
print(f"  autograd multithreading off: iteration host {t1*per:.1f} us/view (with GPU drain {(time.perf_counter()-t0)*per:.1f})")
torch.autograd.set_multithreading_enabled(True)
pr = cProfile.Profile(); pr.enable()
for _ in range(20): it()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
print("module calls:", R.counters())
