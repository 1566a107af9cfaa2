"""Config 5 (5 M surfels @2048x2048, one view) stepped eagerly, 3 x 20 timed steps (AGS_FREEZE=1: learning rates 0).
Used for A/B runs of library builds: AGS_LIB_PATH=scratch/libags_<tag>.so python profiles/experiments/c5_eager_steps.py"""
import sys, os, torch
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."); sys.path.insert(0, R)
from active_gs_amd import env_config
env_config.apply_env(os.environ)   # the package itself reads no environment variable
from active_gs_amd import raster_api as api
from active_gs_amd.camera import camera_matrices
from active_gs_amd.synthetic import make_camera, make_room_scene
from active_gs_amd.trainer import SurfelTrainer
dev = torch.device("cuda:0"); n, h, w = 5_000_000, 2048, 2048
raw = {k: v.to(dev) for k, v in make_room_scene(n, "office0", seed=0).items()}
c2w, K = make_camera(0, h, w, room="office0")
cm = camera_matrices(c2w[None], K[None], 0.001, 10.0)
cam = api.Camera(h, w, cm["tanfov"][0, 0].item(), cm["tanfov"][0, 1].item(), cm["viewmatrix"][0].to(dev), cm["projmatrix"][0].to(dev), torch.zeros(4, device=dev))
tr = SurfelTrainer(raw, lrs=dict(mean=0.0, scale=0.0, rotation=0.0, opacity=0.0, harmonic=0.0) if os.environ.get("AGS_FREEZE") == "1" else None)
P = h * w
gen = torch.Generator().manual_seed(1234)
d_img = [(torch.randn(c, h, w, generator=gen) / P).to(dev) for c in (3, 3, 1)]
fn = lambda v, st: (d_img[0], d_img[1], d_img[2], None, None)
for _ in range(30):
    tr.step([cam], fn, 12_000_000)
torch.cuda.synchronize()
import time
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        tr.step([cam], fn, 12_000_000)
    e1.record(); torch.cuda.synchronize()
    print("ms/step %.4f" % (e0.elapsed_time(e1) / 20), api.read_status(tr.state_for(h, w, 12_000_000))["num_instances"])
