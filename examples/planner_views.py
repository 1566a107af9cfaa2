#!/usr/bin/env python3
"""The planners' utility render (/root/reference/planning/confidence.py:24-46,
config/planner/confidence.yaml: ~100 candidate views at 128x128) four ways: through the facade +
one by one through the C ABI, concurrently on HIP streams (forward_many), the same replayed from one
hipGraph, and as ONE batched set of launches (ViewBatch -> ags_forward_batch).  Prints one JSON line."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from active_gs_amd import env_config
    env_config.apply_env(os.environ)   # the package itself reads no environment variable
    from active_gs_amd import raster_api as api
    from active_gs_amd.camera import camera_matrices
    from active_gs_amd.synthetic import activate, make_camera, make_room_scene
    dev = torch.device("cuda:0")
    n, h, w, V = 200_000, 128, 128, 100
    a = activate(make_room_scene(n, seed=0))
    g = api.Gaussians(*(a[k].to(dev).contiguous() for k in ("means", "scales", "rotations", "opacities", "colors",
                                                              "confidences")))
    c2w, K = zip(*[make_camera(v, h, w) for v in range(V)])
    cm = camera_matrices(torch.stack(c2w).to(dev), torch.stack(K).to(dev), 0.001, 10.0)
    tan = cm["tanfov"][0].cpu()
    bg = torch.zeros(4, device=dev)
    vm, pm = cm["viewmatrix"].contiguous(), cm["projmatrix"].contiguous()
    cams = [api.Camera(h, w, float(tan[0]), float(tan[1]), vm[v], pm[v], bg) for v in range(V)]
    cap = 1 << 19
    states = [api.alloc_state(n, h, w, cap, dev) for _ in range(V)]

    def timed(fn, reps=5):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    t_seq = timed(lambda: api.forward_many(cams, g, states, None))
    t_streams = timed(lambda: api.forward_many(cams, g, states, api.StreamPool(8)))
    batch = api.ViewBatch(g, V, h, w, float(tan[0]), float(tan[1]), bg, cap, num_streams=8, mode="streams")
    t_graph = round(timed(lambda: batch.render(vm, pm)), 2)
    batch = api.ViewBatch(g, V, h, w, float(tan[0]), float(tan[1]), bg, cap)
    t_batched = round(timed(lambda: batch.render(vm, pm)), 2)
    assert not batch.overflowed()
    diff = max(float((a_.rgb - b_.rgb).abs().max()) for a_, b_ in zip(states, batch.states))
    # what an UNTOUCHED planner gets (planning/confidence.py:24-46: one renderer for the candidates, render_view(i) per
    # candidate under no_grad, confidence[0] and depth[0] consumed) once utils/operations.py's GaussianRenderer is
    # facade.SurfelRenderer: the first request renders all views as one batch, the others are served from it
    from active_gs_amd.facade import SurfelRenderer
    attr = (g.means3D, g.colors[:, None, :].contiguous(), g.opacities, g.confidences, g.scales, g.rotations)
    extr, intr = torch.stack(c2w).to(dev), torch.stack(K).to(dev)

    def planner_loop():
        r = SurfelRenderer(extr, intr, attr, bg, (0.001, 10.0), (h, w), dev)
        acc = 0.0
        with torch.no_grad():
            for i in range(V):
                rgb, depth, normal, opacity, d2n, confidence, importance, count, _ = r.render_view(i)
                acc = acc + confidence[0].sum() + depth[0].sum()
        return acc

    t_facade = round(timed(planner_loop, reps=3), 2)
    print(json.dumps(dict(workload=f"{V} candidate views @{h}x{w}, {n} surfels, forward only",
                          ms_one_by_one=round(t_seq, 2), ms_streams8=round(t_streams, 2), ms_streams8_graph_replay=t_graph,
                          ms_one_batched_launch=t_batched, ms_facade_render_view_loop=t_facade,
                          facade_note="GaussianRenderer-shaped: a new renderer per planning step, render_view(i) for every candidate "
                                      "(incl. the per-view normal / depth2normal post-processing and the consumer's two reductions)",
                          max_abs_rgb_difference_batched_vs_single=diff)))


if __name__ == "__main__":
    main()
