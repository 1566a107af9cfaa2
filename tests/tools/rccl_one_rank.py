"""Run by tests/test_gpu_distributed.py in a subprocess: a ONE-rank RCCL process group with the
data-parallel path forced on (SurfelTrainer.DP_FORCE) - the only way to put torch's RCCL collectives, and their
capture into the step's hipGraph, under test on a single-GPU box.  Prints OK when the captured
steps land on the plain single-process trainer's parameters."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.distributed as dist

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")       # one node: RCCL's bootstrap never needs a NIC picked by host name
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
_t = torch.ones(4, device=dev)
dist.all_reduce(_t)
torch.cuda.synchronize()
assert float(_t.sum()) == 4.0
print("RCCL-UP", flush=True)     # from here on a failure is the product's; before it, the box's (the test tells them apart)
import test_gpu_distributed as T
from active_gs_amd.trainer import RowExchange, SurfelTrainer

RowExchange.GROWTH, RowExchange.SLACK = 1.25, 64
SurfelTrainer.DP_FORCE = True
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
    SurfelTrainer.DP_FORCE = False                     # reference: the plain single-process trainer
    raw2, cams2, grads2 = T._setup([0, 1])
    init = [raw2[k].clone() for k in ("means", "scales", "rotations", "opacities", "harmonics")]
    tr2 = SurfelTrainer(raw2)
    for _ in range(16):
        tr2.step(cams2, fn, T.CAP, device_clock=True)
    SurfelTrainer.DP_FORCE = True
    torch.cuda.synchronize()
    for a, b, i0 in zip(tr.params, tr2.params, init):
        travel = (b - i0).abs().mean()
        assert (a - b).abs().mean() < 5e-3 * travel + 1e-9, (float((a - b).abs().mean()), float(travel))
# the dense slab in ROW CHUNKS (SurfelTrainer.DENSE_CHUNKS): chain rule | all-reduce on the communication stream | Adam per
# chunk.  (1) from IDENTICAL gradient records the chunked tail gives the same bits as one all-reduce of the whole slab
# (two runs of the blend backward differ in the order of its float atomics, so the records are snapshotted);
# (2) the forks and joins of the chunks are recorded inside the captured step graph.
SurfelTrainer.DENSE_CHUNK_MIN_ROWS = 1024
raw, cams, grads = T._setup([0, 1])
fn = lambda v, st: (grads[v][0], grads[v][1], grads[v][2], None, None)
tr = SurfelTrainer(raw, sparse_rows=False)
SurfelTrainer.DENSE_CHUNKS = 1
tr.step(cams, fn, T.CAP, device_clock=True)
assert tr.rows is None and tr._dense_chunked(cams)
done, ticked = tr._dense_blend(cams, fn, T.CAP, True)
torch.cuda.synchronize()
snap = dict(ws=[st.workspace.clone() for _, st in done], params=[p.clone() for p in tr.params], state=tr.optim.state_rows.clone(),
            clock=tr.optim.device_clock.clone(), slab=tr.slab.flat.clone())
out = []
for chunks in (1, 4, 3):
    SurfelTrainer.DENSE_CHUNKS = chunks
    for (_, st), w in zip(done, snap["ws"]):
        st.workspace.copy_(w)
    for p, q in zip(tr.params, snap["params"]):
        p.copy_(q)
    tr.optim.state_rows.copy_(snap["state"]); tr.optim.device_clock.copy_(snap["clock"]); tr.slab.flat.copy_(snap["slab"])
    tr._dense_tail(done, ticked, True)
    torch.cuda.synchronize()
    out.append([p.clone() for p in tr.params] + [tr.optim.state_rows.clone(), tr.slab.flat.clone(), tr.optim.device_clock.clone()])
for other in out[1:]:
    for a, b in zip(out[0], other):
        assert torch.equal(a, b)
assert float(out[0][-2].abs().max()) > 0
SurfelTrainer.DENSE_CHUNKS = 4
replay = tr.capture(cams, fn, T.CAP, repeat=2)
assert replay.collective_in_graph and replay.steps == 2
step0 = int(tr.optim.device_clock.view(torch.int32)[0])
for _ in range(3):
    replay()
torch.cuda.synchronize()
tr.check_overflow()
assert int(tr.optim.device_clock.view(torch.int32)[0]) == step0 + 6
assert all(bool(torch.isfinite(p).all()) for p in tr.params)
dist.destroy_process_group()
print("OK")
