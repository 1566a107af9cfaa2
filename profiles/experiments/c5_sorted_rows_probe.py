"""Probe (round 4): does walking the optimiser's member list in ROW order speed up the per-Gaussian backward + Adam at
config 5?  The list is in insertion order (waves append their new rows as they run: roughly, not exactly, row order).
Steps config 5 eagerly, times ags_k_preprocess_bwd_rows with the library's stage events, sorts the member list in place
(torch.sort of rows[:count]: membership and results are unchanged, only the order the kernel walks it in) and times again."""
import sys, os, ctypes as C, torch
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."); sys.path.insert(0, R)
from active_gs_amd import env_config
env_config.apply_env(os.environ)   # the package itself reads no environment variable
from active_gs_amd import _lib, raster_api as api
from active_gs_amd.camera import camera_matrices
from active_gs_amd.synthetic import make_camera, make_room_scene
from active_gs_amd.trainer import SurfelTrainer
dev = torch.device("cuda:0"); n, h, w = 5_000_000, 2048, 2048
raw = {k: v.to(dev) for k, v in make_room_scene(n, "office0", seed=0).items()}
c2w, K = make_camera(0, h, w, room="office0")
cm = camera_matrices(c2w[None], K[None], 0.001, 10.0)
cam = api.Camera(h, w, cm["tanfov"][0, 0].item(), cm["tanfov"][0, 1].item(), cm["viewmatrix"][0].to(dev), cm["projmatrix"][0].to(dev), torch.zeros(4, device=dev))
tr = SurfelTrainer(raw, lrs=dict(mean=0.0, scale=0.0, rotation=0.0, opacity=0.0, harmonic=0.0))
gen = torch.Generator().manual_seed(1234)
d_img = [(torch.randn(c, h, w, generator=gen) / (h * w)).to(dev) for c in (3, 3, 1)]
fn = lambda v, st: (d_img[0], d_img[1], d_img[2], None, None)
lib = _lib.load()
def stage_times(tag):
    for _ in range(10):
        tr.step([cam], fn, 12_000_000)
    torch.cuda.synchronize()
    lib.ags_profile_enable(30)
    for _ in range(30):
        tr.step([cam], fn, 12_000_000)
    torch.cuda.synchronize()
    out = {}
    for name, sid in (("preprocess", 0), ("binning", 1), ("render_fwd", 2), ("render_bwd", 3), ("preprocess_bwd", 4)):
        a, m, c = C.c_float(), C.c_float(), C.c_int32()
        lib.ags_profile_read(sid, C.byref(a), C.byref(m), C.byref(c)); out[name] = round(m.value * 1e3, 1)
    lib.ags_profile_enable(0)
    print(tag, out)
stage_times("insertion order:")
cnt = int(tr.rows.count.item())
rows = tr.rows.rows[:cnt]
inv = int((rows[1:] < rows[:-1]).sum())
print("members", cnt, "adjacent inversions", inv, "median |step|", float((rows[1:] - rows[:-1]).abs().float().median()))
rows.copy_(torch.sort(rows).values)
stage_times("row order:      ")
