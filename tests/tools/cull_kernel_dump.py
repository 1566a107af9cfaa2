"""Helper of tests/test_gpu_cull_kernel.py: render raw-parameter scenes through the C ABI and dump everything the
per-Gaussian stage decides (radii, visible count, instance count, the projected records of the visible rows, the row
set) plus the images; run once with AGS_PRE_CULL_MIN_N=0 (cull-first kernel everywhere) and once with a huge value
(plain kernel everywhere).  usage: cull_kernel_dump.py out.pt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from active_gs_amd import env_config, raster_api as api  # noqa: E402
from active_gs_amd.camera import camera_matrices  # noqa: E402
from active_gs_amd.synthetic import make_camera, make_room_scene  # noqa: E402

env_config.apply_env(os.environ)
dev = torch.device("cuda:0")
out = {}
cases = [("small", 6000, 136, 240, 3, 1.5, 1.7), ("ragged", 5037, 100, 150, 1, 0.5, 1.0), ("tiny", 300, 64, 64, 2, 2.0, 1.0),
         ("large", 1_200_000, 680, 1200, 0, 0.0, 1.0)]
for tag, n, h, w, view, ds, qs in cases:
    raw = {k: v.to(dev) for k, v in make_room_scene(n, seed=view).items()}
    raw["scales"][:, :2] += ds                      # some scales hit the 0.05 clamp
    raw["rotations"] *= qs                          # un-normalised raw quaternions
    c2w, K = make_camera(view, h, w)
    cm = camera_matrices(c2w[None], K[None], 0.001, 10.0)
    cam = api.Camera(h, w, cm["tanfov"][0, 0].item(), cm["tanfov"][0, 1].item(), cm["viewmatrix"][0].to(dev),
                     cm["projmatrix"][0].to(dev), torch.tensor([0.1, 0.2, 0.3, 0.0], device=dev))
    g = api.Gaussians(raw["means"], raw["scales"], raw["rotations"], raw["opacities"], raw["harmonics"].view(n, 3).contiguous(),
                      raw["confidences"], raw_params=True)
    rows = api.RowSet(n, dev)
    st = api.alloc_state(n, h, w, 1 << 22, dev)
    api.forward(cam, g, st, touched=rows)
    info = api.read_status(st)
    assert not info["overflow"], info
    geom = api.workspace_region(st, n, h, w, api.REGION_GEOM, torch.float32).view(n, 16)
    vis = st.radii > 0
    out[tag] = dict(radii=st.radii.cpu(), visible=info["num_visible"], instances=info["num_instances"],
                    geom_visible=geom[vis].cpu(), rgb=st.rgb.cpu(), depth=st.depth.cpu(), opacity=st.opacity.cpu(),
                    members=torch.sort(rows.rows[: int(rows.count.item())]).values.cpu())
    # a batch of views (blockIdx.y) with statistics, front_only and a render mask
    if n <= 10000:
        V = 3
        cms = [camera_matrices(make_camera(v, h, w)[0][None], make_camera(v, h, w)[1][None], 0.001, 10.0) for v in range(V)]
        masks = (torch.rand(V, h, w, generator=torch.Generator().manual_seed(1)) > 0.3).float().to(dev)
        vb = api.ViewBatch(g, V, h, w, cam.tanfovx, cam.tanfovy, cam.bg, 1 << 20, want_stats=True, front_only=True, render_masks=masks)
        vb.render(torch.stack([c["viewmatrix"][0] for c in cms]).to(dev), torch.stack([c["projmatrix"][0] for c in cms]).to(dev))
        torch.cuda.synchronize()
        assert not vb.overflowed()
        out[tag + "_batch"] = dict(radii=vb.radii.cpu(), count=vb.count.cpu(), rgb=vb.rgb.cpu())
torch.save(out, sys.argv[1])
