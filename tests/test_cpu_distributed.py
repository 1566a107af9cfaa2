"""View-parallel data parallelism on CPU: world_size 2 over gloo reproduces the reference's
single-process GaussianMap.train() capture (gradients all-reduced before a replicated Adam)."""
import os

import torch

from _spawn import spawn_ranks

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def _worker(rank, world):
    torch.set_num_threads(2)
    from test_cpu_host_logic import _train_from_fixture
    d = torch.load(os.path.join(GOLD, "train.pt"))
    t = _train_from_fixture(d)
    assert t.world == world and t.rank == rank
    return dict(raw_final={k: getattr(t, k).clone() for k in d["raw_final"]},
                training_performance=t.training_performance.clone(), view_supports=t.view_supports.clone(),
                view_scores=t.view_scores.clone(), view_means=t.view_means.clone(), losses=list(t.last_losses))


def test_two_rank_view_parallel_training_matches_reference_capture():
    ret = spawn_ranks(_worker, world=2, watchdog_s=600)
    d = torch.load(os.path.join(GOLD, "train.pt"))
    r0, r1 = ret[0], ret[1]
    for k, ref in d["raw_final"].items():
        assert torch.equal(r0["raw_final"][k], r1["raw_final"][k]), f"replicas diverged on {k}"
        assert torch.allclose(r0["raw_final"][k], ref, rtol=2e-4, atol=2e-4), k
    assert torch.allclose(r0["training_performance"], d["training_performance"], rtol=1e-3, atol=1e-5)
    assert torch.equal(r0["view_supports"], d["view_supports"])
    assert torch.allclose(r0["view_scores"], d["view_scores"], atol=1e-4)
    assert r0["losses"] == r1["losses"]


def _gather_worker(rank, world):
    from active_gs_amd.trainer import RowExchange
    n = 1000
    grads = [torch.zeros(n, w) for w in (3, 3, 4, 1, 3)]
    x = RowExchange(n, grads, torch.device("cpu"), None)
    cap = x.agree(local_rows=40 + 10 * rank, slab_floats=14 * n)          # largest rank: 50 rows
    assert cap == 0 and x.capacity == 0
    small = RowExchange(100000, grads, torch.device("cpu"), None)
    assert small.agree(local_rows=100 + rank, slab_floats=14 * 100000) == int(RowExchange.GROWTH * 101) + RowExchange.SLACK
    gen = torch.Generator().manual_seed(rank)
    small.send.copy_(torch.randn(small.send.shape, generator=gen))
    ids = torch.randint(0, 100000, (small.capacity,), generator=gen, dtype=torch.int32)   # denormal bit patterns
    small.send.view(-1, 16)[1:, 14] = ids.view(torch.float32)
    small.send[:2] = torch.tensor([small.capacity, small.capacity + rank], dtype=torch.int32).view(torch.float32)
    small.gather()
    return dict(cap=cap, send=small.send.view(torch.int32).clone(), recv=small.recv.view(torch.int32).clone(),
                overflow=small.overflowed())


def test_row_exchange_host_logic_two_ranks():
    """RowExchange (trainer.py) over gloo: the ranks agree on one segment size, and the gather moves
    every rank's segment to every rank bit for bit (row ids travel as raw int32 bits)."""
    ret = spawn_ranks(_gather_worker, world=2, watchdog_s=600)
    assert ret[0]["cap"] == ret[1]["cap"] == 0            # two 1000-row segments against a 1000-row slab: stay dense
    for r in (0, 1):
        assert torch.equal(ret[r]["recv"][0], ret[0]["send"]) and torch.equal(ret[r]["recv"][1], ret[1]["send"])
        assert ret[r]["overflow"]                         # rank 1 announced capacity + 1 rows
