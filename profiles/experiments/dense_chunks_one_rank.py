#!/usr/bin/env python3
"""What cutting the dense exchange into row chunks costs WITHOUT a wire (one-rank RCCL group, configuration 4's per-rank
share at 8 ranks: 1.5 M surfels, 4 views @1200x680): the step and its tail for DENSE_CHUNKS = 1, 2, 4, 8."""
import os, sys, json, statistics
os.environ.setdefault("AGS_DP_FORCE", "1")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29579")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch, torch.distributed as dist
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from active_gs_amd import env_config
env_config.apply_env(os.environ)   # the package itself reads no environment variable
from active_gs_amd import raster_api as api
from active_gs_amd.camera import camera_matrices
from active_gs_amd.synthetic import make_camera, make_room_scene
from active_gs_amd.trainer import SurfelTrainer
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev)
n, h, w, views = 1_500_000, 680, 1200, 4
cams = []
for v in range(views):
    c2w, K = make_camera(v, h, w)
    cm = camera_matrices(c2w[None].to(dev), K[None].to(dev), 0.001, 10.0)
    tan = cm["tanfov"][0].cpu()
    cams.append(api.Camera(h, w, float(tan[0]), float(tan[1]), cm["viewmatrix"][0].contiguous(), cm["projmatrix"][0].contiguous(), torch.zeros(4, device=dev)))
gen = torch.Generator().manual_seed(4)
d_img = [(torch.randn(c, h, w, generator=gen) / (h * w * 32)).to(dev) for c in (3, 3, 1)]
fn = lambda v, st: (d_img[0], d_img[1], d_img[2], None, None)
# (the same scene, views, capacity and zero learning rates for every form of the step)
for K_ in ("fused", "rows", 1, 2, 4, 8):
    # "fused": the single-rank step (row-set Adam fused into the per-Gaussian backward, no exchange); "rows": the
    # data-parallel step with the row all-gather; 1..8: the data-parallel step with the dense slab in that many chunks
    os.environ["AGS_DP_FORCE"] = "0" if K_ == "fused" else "1"
    if isinstance(K_, int):
        SurfelTrainer.DENSE_CHUNKS = K_
    raw = {k: v.to(dev) for k, v in make_room_scene(n, room="room0", seed=0).items()}
    tr = SurfelTrainer(raw, sparse_rows=not isinstance(K_, int), view_streams=4,
                       lrs=dict(mean=0.0, scale=0.0, rotation=0.0, opacity=0.0, harmonic=0.0))
    cap = 6_000_000
    for _ in range(3):
        tr.step(cams, fn, cap)
    tr.check_overflow()
    replay = tr.capture(cams, fn, cap)
    for _ in range(5):
        replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        replay()
    e1.record(); torch.cuda.synchronize()
    tr.tail_probe = []
    for _ in range(8):
        tr.step(cams, fn, cap)
    tl = [SurfelTrainer.tail_timeline(r) for r in tr.tail_probe]
    med = lambda k: round(statistics.median(t[k] for t in tl), 4) if tl else None
    print(json.dumps(dict(form=K_ if isinstance(K_, str) else f"dense, {K_} chunk(s)", ms_per_step_graph=round(e0.elapsed_time(e1) / 50, 4),
                          chain_rule_ms=med("rows_ms"), all_reduce_sum_ms=med("all_reduce_sum_ms"),
                          adam_ms=med("adam_ms"), tail_ms=med("tail_ms"), exposed_ms=med("exposed_ms"))), flush=True)
    del tr, replay, raw
    torch.cuda.empty_cache()
dist.destroy_process_group()
