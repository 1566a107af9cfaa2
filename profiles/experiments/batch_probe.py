"""C4's per-GPU share (1.5 M surfels, 4 views @1200x680): the rank's views as ONE set of launches (ags_forward_batch /
ags_backward_batch, blockIdx.y = view; row-set Adam behind them) against per-view launches (SurfelTrainer.step today).
Both captured into a hipGraph and replayed.  usage: python profiles/experiments/batch_probe.py  (GPU box)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from active_gs_amd import env_config
env_config.apply_env(os.environ)   # the package itself reads no environment variable
from active_gs_amd import raster_api as api
from active_gs_amd.camera import camera_matrices
from active_gs_amd.optimizer import FusedAdam
from active_gs_amd.synthetic import make_camera, make_room_scene
from active_gs_amd.trainer import SurfelTrainer, GradSlab, DEFAULT_LRS
import bench

dev = torch.device("cuda:0")
n, h, w, views = 1_500_000, 680, 1200, 4
raw = {k: v.to(dev) for k, v in make_room_scene(n, room="room0", seed=0).items()}
cams_m = [make_camera(v, h, w) for v in range(views)]
cm = camera_matrices(torch.stack([c for c, _ in cams_m]), torch.stack([k for _, k in cams_m]), 0.001, 10.0)
gen = torch.Generator().manual_seed(4)
d_img = [(torch.randn(views, c, h, w, generator=gen) / (h * w * views)).to(dev) for c in (3, 3, 1)]

# ---- batched
g = api.Gaussians(raw["means"], raw["scales"], raw["rotations"], raw["opacities"], raw["harmonics"].view(n, 3), raw["confidences"],
                  raw_params=True)
cap = 1 << 22
while True:
    vb = api.ViewBatch(g, views, h, w, float(cm["tanfov"][0, 0]), float(cm["tanfov"][0, 1]), torch.zeros(4, device=dev), cap)
    vb.viewmats.copy_(cm["viewmatrix"].to(dev)); vb.projmats.copy_(cm["projmatrix"].to(dev))
    vb.forward(views)
    need = int(vb.statuses()[:, 7].max())
    if need <= cap: break
    cap = int(need * 1.25)
params = [raw["means"], raw["scales"], raw["rotations"], raw["opacities"], raw["harmonics"]]
lr = DEFAULT_LRS
opt = FusedAdam(params, [lr["mean"], lr["scale"], lr["rotation"], lr["opacity"], lr["harmonic"]], eps=1e-15)
rows = api.RowSet(n, dev)
opt.touched, opt.zero_grad = rows, True
slab = GradSlab(n, dev)
opt.use_clock(True)
def batched_step():
    vb.forward(views, touched=rows)
    vb.backward(views, d_img[0], d_img[1], d_img[2], slab.grads, touched=rows, adam_tick=opt.tick_args())
    opt.step(slab.as_list(), device_clock=True, pre_ticked=True)
for _ in range(3): batched_step()
torch.cuda.synchronize()
assert not vb.overflowed(views)
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
gr = torch.cuda.CUDAGraph()
with torch.cuda.stream(side):
    with torch.cuda.graph(gr, stream=side, capture_error_mode="thread_local"):
        batched_step()
torch.cuda.current_stream().wait_stream(side)
for _ in range(5): gr.replay()
s = bench.summarise(bench.time_samples(lambda: [gr.replay() for _ in range(20)], 5, False, dev), 20)
print("batched (one set of launches for 4 views):", round(s["median"], 4), "ms/step", s)
out = {"batched_ms": s["median"]}
# ---- per view (what SurfelTrainer.step does)
raw2 = {k: v.to(dev) for k, v in make_room_scene(n, room="room0", seed=0).items()}
r = bench.measure_config("c4 per-view launches", n, h, w, views, "room0", 20, dev)
print("per-view launches:", r["ms_per_step"], r["stage_ms_per_view"])
out["per_view_ms"] = r["ms_per_step"]
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r03_batch_probe.json"), "w"))
