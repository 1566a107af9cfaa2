import sys, os, json
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from active_gs_amd import env_config
env_config.apply_env(os.environ)   # the package itself reads no environment variable
from active_gs_amd import raster_api as api
from active_gs_amd.camera import camera_matrices
from active_gs_amd.fused_map_trainer import FusedMapTrainer
from active_gs_amd.synthetic import activate, make_camera, make_room_scene
dev = torch.device("cuda:0"); h = w = 512; NGT = 400_000; KF = 50
gt = {k: v.to(dev) for k, v in make_room_scene(NGT, seed=0).items()}
gt["scales"][:, :2] += 0.6; gt["opacities"] += 4.0
a = activate(gt)
g = api.Gaussians(a["means"], a["scales"], a["rotations"], a["opacities"], gt["harmonics"].view(-1, 3).contiguous(), a["confidences"])
st = api.alloc_state(NGT, h, w, 1 << 24, dev)
frames, cams = [], []
for v in range(KF):
    c2w, K = make_camera(v, h, w)
    cm = camera_matrices(c2w[None].to(dev), K[None].to(dev), 0.001, 10.0)
    tan = cm["tanfov"][0].cpu()
    cam = api.Camera(h, w, float(tan[0]), float(tan[1]), cm["viewmatrix"][0].contiguous(), cm["projmatrix"][0].contiguous(), torch.zeros(4, device=dev))
    api.forward(cam, g, st); cams.append(cam)
    depth = torch.where(st.opacity > 0.5, st.depth, torch.zeros_like(st.depth))
    frames.append(dict(rgb=st.rgb.clone().clamp(0, 1), depth=depth.clone(), extrinsic=c2w.to(dev), intrinsic=K.to(dev), depth_range=torch.tensor([0.001, 10.0], device=dev)))
z = lambda *s: torch.zeros(*s, device=dev)
raw = dict(means=z(0, 3), scales=z(0, 3), rotations=z(0, 4), opacities=z(0), harmonics=z(0, 1, 3))
np.random.seed(0)
tr = FusedMapTrainer(raw, [], dict(optimization_steps=10, sampler="device"), use_graph=False, num_streams=4, batched=True)
for f in frames:
    tr.add_gaussians(f); tr.train(); tr.is_init = True
n = tr.means.shape[0]
gm = tr._gaussians()
st2 = api.alloc_state(n, h, w, 1 << 23, dev)
T = (h // 16) * (w // 16)
allL = []
for cam in cams[-11:]:
    api.forward(cam, gm, st2)
    info = api.read_status(st2)
    rg = st2.workspace[8448:8448 + T * 8].view(torch.int32).view(T, 2).cpu().numpy()
    L = rg[:, 1] - rg[:, 0]
    allL.append(L)
    print("view: instances", info["num_instances"], "visible", info["num_visible"], "tile len mean %.1f p50 %d p90 %d p99 %d max %d" % (L.mean(), *np.percentile(L, [50, 90, 99]), L.max()))
L = np.concatenate(allL)
print("batch: tiles", L.size, "mean %.1f max %d  sum/max = %.0f tiles-equivalents; top-1%% share of work %.3f" % (L.mean(), L.max(), L.sum() / L.max(), np.sort(L)[-L.size // 100:].sum() / L.sum()))
print("hist", np.histogram(L, bins=[0, 1, 32, 64, 128, 256, 512, 1024, 2048, 8192])[0].tolist())
