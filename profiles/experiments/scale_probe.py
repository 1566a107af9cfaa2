"""One 2048x2048 view of an office0 stand-in with N surfels stepped eagerly (forward + backward + Adam): does the path hold
at map sizes far beyond BASELINE.json's 5 M?   python profiles/experiments/scale_probe.py 60000000"""
import sys, os, torch, time
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."); sys.path.insert(0, R)
from active_gs_amd import env_config
env_config.apply_env(os.environ)   # the package itself reads no environment variable
from active_gs_amd import raster_api as api
from active_gs_amd.camera import camera_matrices
from active_gs_amd.synthetic import make_camera, make_room_scene
from active_gs_amd.trainer import SurfelTrainer
dev = torch.device("cuda:0"); n = int(sys.argv[1]); h, w = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (2048, 2048)
t0 = time.time()
raw = {k: v.to(dev) for k, v in make_room_scene(n, "office0", seed=0).items()}
print("scene made in %.1f s" % (time.time() - t0), flush=True)
c2w, K = make_camera(0, h, w, room="office0")
cm = camera_matrices(c2w[None], K[None], 0.001, 10.0)
cam = api.Camera(h, w, cm["tanfov"][0, 0].item(), cm["tanfov"][0, 1].item(), cm["viewmatrix"][0].to(dev), cm["projmatrix"][0].to(dev), torch.zeros(4, device=dev))
tr = SurfelTrainer(raw)
P = h * w
gen = torch.Generator().manual_seed(1234)
d_img = [(torch.randn(c, h, w, generator=gen) / P).to(dev) for c in (3, 3, 1)]
fn = lambda v, st: (d_img[0], d_img[1], d_img[2], None, None)
cap = 1 << 24
while True:
    tr.step([cam], fn, cap)
    st = tr.state_for(h, w, cap)
    info = api.read_status(st)
    if not info["overflow"] and info["needed"] <= cap: break
    cap = int(info["needed"] * 1.25)
print("status", info, "workspace GB %.2f" % (st.workspace.numel() / 2**30), flush=True)
for _ in range(5): tr.step([cam], fn, cap)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): tr.step([cam], fn, cap)
torch.cuda.synchronize()
tr.check_overflow()
print("n %d: %.3f ms/step, %.2f G surfels/s; rgb finite %s; mem GB %.1f" % (n, (time.perf_counter() - t0) / 20 * 1e3, n / ((time.perf_counter() - t0) / 20) / 1e9, bool(torch.isfinite(st.rgb).all()), torch.cuda.max_memory_allocated() / 2**30))
