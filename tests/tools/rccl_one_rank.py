"""Run by tests/test_gpu_distributed.py in a subprocess: a ONE-rank RCCL process group with the
data-parallel path forced on (AGS_DP_FORCE=1) - the only way to put torch's RCCL collectives, and their
capture into the step's hipGraph, under test on a single-GPU box.  Prints OK when the captured
steps land on the plain single-process trainer's parameters."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["AGS_DP_FORCE"] = "1"
import torch
import torch.distributed as dist

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
import test_gpu_distributed as T
from active_gs_amd.trainer import RowExchange, SurfelTrainer

RowExchange.GROWTH, RowExchange.SLACK = 1.25, 64
for sparse in (True, False):
    raw, cams, grads = T._setup([0, 1])
    fn = lambda v, st: (grads[v][0], grads[v][1], grads[v][2], None, None)
    tr = SurfelTrainer(raw, sparse_rows=sparse)
    tr.step(cams, fn, T.CAP, device_clock=True)
    assert (tr.exchange is not None and tr.exchange.capacity > 0) == sparse
    replay = tr.capture(cams, fn, T.CAP, repeat=3)
    assert replay.collective_in_graph and replay.steps == 3, (replay.collective_in_graph, replay.steps)
    for _ in range(5):
        replay()
    torch.cuda.synchronize()
    assert int(tr.optim.device_clock.view(torch.int32)[0]) == 16
    if sparse:
        assert not tr.exchange.overflowed() and float(tr.slab.flat.abs().max()) == 0.0
    os.environ["AGS_DP_FORCE"] = "0"                     # reference: the plain single-process trainer
    raw2, cams2, grads2 = T._setup([0, 1])
    init = [raw2[k].clone() for k in ("means", "scales", "rotations", "opacities", "harmonics")]
    tr2 = SurfelTrainer(raw2)
    for _ in range(16):
        tr2.step(cams2, fn, T.CAP, device_clock=True)
    os.environ["AGS_DP_FORCE"] = "1"
    torch.cuda.synchronize()
    for a, b, i0 in zip(tr.params, tr2.params, init):
        travel = (b - i0).abs().mean()
        assert (a - b).abs().mean() < 5e-3 * travel + 1e-9, (float((a - b).abs().mean()), float(travel))
dist.destroy_process_group()
print("OK")
