"""View-parallel SurfelTrainer on the GPU: two ranks (sharing cuda:0, gloo transport) each render
one view, all-reduce the gradient slab, step a replicated Adam — eagerly and as
graph | collective | graph — and must land on the single-process two-view result."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, H, W, CAP, STEPS = 8000, 136, 240, 1 << 20, 3


def _setup(views):
    import sys
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from _scenes import room_case
    from active_gs_amd import raster_api as api
    from active_gs_amd.synthetic import make_room_scene
    dev = torch.device("cuda:0")
    raw = {k: v.to(dev) for k, v in make_room_scene(N, seed=8).items()}
    raw["scales"][:, :2] += 1.0
    cams, grads = [], []
    for v in views:
        _, S = room_case(16, H, W, view=v, seed=0)
        cams.append(api.Camera(H, W, S.tanfovx, S.tanfovy, S.viewmatrix.to(dev), S.projmatrix.to(dev), S.bg.to(dev)))
        gen = torch.Generator().manual_seed(100 + v)
        grads.append([(torch.randn(c, H, W, generator=gen) / (H * W * 2)).to(dev) for c in (3, 3, 1)])
    return raw, cams, grads


def _worker(rank, world, port, use_graph, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from active_gs_amd.trainer import SurfelTrainer
        raw, cams, grads = _setup([rank])          # rank r renders view r
        tr = SurfelTrainer(raw)
        fn = lambda v, st: (grads[v][0], grads[v][1], grads[v][2], None, None)
        tr.step(cams, fn, CAP, device_clock=True)
        if use_graph:
            replay = tr.capture(cams, fn, CAP)
            for _ in range(STEPS - 1):
                replay()
        else:
            for _ in range(STEPS - 1):
                tr.step(cams, fn, CAP, device_clock=True)
        torch.cuda.synchronize()
        ret[rank] = [p.cpu() for p in tr.params]
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("use_graph", [False, True])
def test_two_ranks_equal_single_process_two_views(agslib, use_graph):
    from active_gs_amd.trainer import SurfelTrainer
    raw, cams, grads = _setup([0, 1])
    tr = SurfelTrainer(raw)
    fn = lambda v, st: (grads[v][0], grads[v][1], grads[v][2], None, None)
    for _ in range(STEPS):
        tr.step(cams, fn, CAP, device_clock=True)
    torch.cuda.synchronize()
    ref = [p.cpu() for p in tr.params]
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(2, port, use_graph, ret), nprocs=2, join=True)
        for a, b in zip(ret[0], ret[1]):
            assert torch.equal(a, b)                               # replicas stay identical
        for a, r, init in zip(ret[0], ref, [raw[k].cpu() for k in ("means", "scales", "rotations", "opacities", "harmonics")]):
            travel = (r - init).abs().mean()
            assert (a - r).abs().mean() < 5e-3 * travel + 1e-9     # Adam eps=1e-15: sign flips on ~0 gradients
