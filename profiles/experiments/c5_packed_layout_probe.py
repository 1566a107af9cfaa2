"""Probe: config 5 with scales / rotation / opacity / colour as strided views of ONE (N,16) buffer
(PACK=1, needs a probe build whose cull-first and member-row kernels index with stride 16) against the separate arrays
(PACK=0, product build); prints the per-stage medians.  The probe build is not kept: the layout was not adopted."""
import sys, os, torch, time
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."); sys.path.insert(0, R)
from active_gs_amd import env_config
env_config.apply_env(os.environ)   # the package itself reads no environment variable
from active_gs_amd import raster_api as api, _lib
from active_gs_amd.camera import camera_matrices
from active_gs_amd.synthetic import make_camera, make_room_scene
from active_gs_amd.trainer import SurfelTrainer
PACK = os.environ.get("PACK") == "1"
dev = torch.device("cuda:0"); n, h, w = 5_000_000, 2048, 2048
raw = {k: v.to(dev) for k, v in make_room_scene(n, "office0", seed=0).items()}
c2w, K = make_camera(0, h, w, room="office0")
cm = camera_matrices(c2w[None], K[None], 0.001, 10.0)
cam = api.Camera(h, w, cm["tanfov"][0, 0].item(), cm["tanfov"][0, 1].item(), cm["viewmatrix"][0].to(dev), cm["projmatrix"][0].to(dev), torch.zeros(4, device=dev))
tr = SurfelTrainer(raw, lrs=dict(mean=0.0, scale=0.0, rotation=0.0, opacity=0.0, harmonic=0.0))
if PACK:
    api._require_cuda = lambda t, name: None
    attrs = torch.zeros(n, 16, device=dev)
    attrs[:, 0:3] = tr.raw["scales"]; attrs[:, 4:8] = tr.raw["rotations"]; attrs[:, 8] = tr.raw["opacities"]; attrs[:, 12:15] = tr.raw["harmonics"].view(n, 3)
    tr.raw["scales"] = attrs[:, 0:3]; tr.raw["rotations"] = attrs[:, 4:8]; tr.raw["opacities"] = attrs[:, 8]; tr.raw["harmonics"] = attrs[:, 12:15].unsqueeze(1)
    tr.params = [tr.raw["means"], tr.raw["scales"], tr.raw["rotations"], tr.raw["opacities"], tr.raw["harmonics"]]
    tr.optim.params = tr.params
    if hasattr(tr.optim, "_struct"): tr.optim._struct = None
P = h * w
gen = torch.Generator().manual_seed(1234)
d_img = [(torch.randn(c, h, w, generator=gen) / P).to(dev) for c in (3, 3, 1)]
fn = lambda v, st: (d_img[0], d_img[1], d_img[2], None, None)
lib = _lib.load()
for _ in range(10):
    tr.step([cam], fn, 12_000_000)
torch.cuda.synchronize()
st = tr.state_for(h, w, 12_000_000)
print("status", api.read_status(st)["num_instances"], "rgb sum %.3f" % float(st.rgb.sum()))
_lib.check(lib.ags_profile_enable(30), "x")
for _ in range(30):
    tr.step([cam], fn, 12_000_000)
torch.cuda.synchronize()
import ctypes as C
names = ["preprocess", "binning", "render_fwd", "render_bwd", "preprocess_bwd"]
out = {}
for s_, nm in enumerate(names):
    a, m, c = C.c_float(), C.c_float(), C.c_int32()
    lib.ags_profile_read(s_, C.byref(a), C.byref(m), C.byref(c)); out[nm] = round(m.value * 1e3, 1)
print("PACK" if PACK else "SEPARATE", out)
