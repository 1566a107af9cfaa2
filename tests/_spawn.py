"""Multi-process test plumbing: N fresh interpreters (start method "spawn" - nothing is ever forked from a
process that has touched the GPU), a FILE rendezvous (no TCP store, so no port to lose a race for) and results
returned as files (``torch.save`` to ``<dir>/rank<r>.pt``, renamed into place; the parent loads them after the
workers have been joined).  No Manager, no proxies, no descriptor passing between the processes."""
import os
import shutil
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _entry(rank, world, outdir, worker, args, watchdog_s):
    import sys
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    if watchdog_s:
        # a worker still running after `watchdog_s` dumps every thread's stack and exits with code 1
        import faulthandler
        faulthandler.dump_traceback_later(watchdog_s, exit=True)
    import torch.distributed as dist
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")       # every rank is on this host: never pick a NIC by hostname
    dist.init_process_group("gloo", init_method="file://" + os.path.join(outdir, "rendezvous"), rank=rank, world_size=world)
    try:
        result = worker(rank, world, *args)
        tmp = os.path.join(outdir, f"rank{rank}.pt.part")
        torch.save(result, tmp)
        os.replace(tmp, os.path.join(outdir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


def spawn_ranks(worker, args=(), world=2, watchdog_s=0, stall_retries=0):
    """Runs ``worker(rank, world, *args)`` in `world` spawned processes joined in one gloo group; returns
    ``{rank: what the worker returned}`` (CPU tensors / plain Python only).  An exception in a worker propagates
    with the worker's traceback.  Only the watchdog's own signature (exit code 1, no signal, no exception) is a
    stall, and it is tried again at most `stall_retries` times; the last stall raises."""
    import torch.multiprocessing as mp
    from torch.multiprocessing.spawn import ProcessExitedException
    last = None
    for _ in range(stall_retries + 1):
        outdir = tempfile.mkdtemp(prefix="ags_ranks_")
        try:
            try:
                mp.spawn(_entry, args=(world, outdir, worker, tuple(args), watchdog_s), nprocs=world, join=True)
            except ProcessExitedException as e:
                if getattr(e, "signal_name", None) or getattr(e, "exit_code", 1) != 1:
                    raise                                   # killed by a signal / odd exit code: a crash, not a stall
                last = e
                continue
            return {r: torch.load(os.path.join(outdir, f"rank{r}.pt"), weights_only=False) for r in range(world)}
        finally:
            shutil.rmtree(outdir, ignore_errors=True)
    raise RuntimeError(f"{world} ranks stalled until the watchdog ({stall_retries + 1} attempt(s)): {last}")
