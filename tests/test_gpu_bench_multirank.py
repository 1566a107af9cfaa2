"""bench.py's N>1 path end to end (``python bench.py --gpus 2`` launching its own two ranks, process group,
graph | collective | graph step, barriers, max over ranks, one JSON line relayed from rank 0) with the two ranks
sharing the single GPU over gloo."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


STRONG_REDUCED = "c4=150000,8,680,1200;c5=400000,4,1024,1024"   # configurations 4 and 5 at a tenth of the surfels, 8 / 4 views


def _stall_report(r, name):
    """a rank that hung: keep the job's whole output for a post-mortem and fail with the watchdog's stacks"""
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, name), "w") as f:
            f.write(r.stderr + "\n==== stdout ====\n" + r.stdout)
    trace = [l for l in r.stderr.splitlines() if " in " in l and "line " in l and ", line" not in l]   # faulthandler frames
    pytest.fail("bench ranks sharing the GPU over gloo stalled until the watchdog (a deadlock regression looks exactly "
                "like this: no second attempt); stacks:\n" + "\n".join(trace[-60:]))


def test_bench_two_ranks_on_one_gpu(agslib):
    """The driver's multi-GPU command shape (bench.py finds no RANK in the environment and launches the ranks itself), two
    ranks sharing the GPU over gloo: the weak-scaled headline AND the strong-scaled configurations 4 and 5
    (config.secondary.c4 / .c5) at reduced sizes - views dealt out over the ranks, the exchange agree() picks, the dense
    slab all-reduced in row chunks for configuration 4, replicas bit-identical."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(AGS_BENCH_SHARE_GPU="1", AGS_BENCH_BACKEND="gloo", AGS_BENCH_WATCHDOG="300", AGS_BENCH_STRONG=STRONG_REDUCED)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "3",
           "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    if r.returncode != 0 and "Timeout" in r.stderr:
        _stall_report(r, "bench_multirank_stall.log")
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                  # exactly one JSON line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["parallelism"] == "view-parallel dp2" and not d["config"]["overflow"]
    assert d["config"]["launch"].startswith("hipGraph") and "replay(s) per sample" in d["config"]["launch"]
    assert d["samples"] >= 21 and d["ms_per_step_min"] <= d["ms_per_step"] <= d["ms_per_step_max"]
    x = d["config"]["exchange"]
    assert x["world_size"] == 2 and len(x["ranks"]) == 2 and {e["rank"] for e in x["ranks"]} == {0, 1}
    assert set(d["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source"}
    assert d["config"]["exchange"]["refused_steps"] == 0 and d["config"]["derived_rates"]["tile_instances_per_s"] > 0
    # -- configurations 4 and 5, strong-scaled over the two ranks
    sec = d["config"]["secondary"]
    c4, c5 = sec["c4"], sec["c5"]
    assert isinstance(c4, dict) and isinstance(c5, dict), sec
    for c, total in ((c4, 8), (c5, 4)):
        assert c["scaling"] == "strong" and c["world_size"] == 2 and c["views_total"] == total and c["views_this_rank"] == total // 2
        assert c["ms_per_step"] > 0 and c["gaussians_per_s"] > 0 and c["reduced_size"]
        assert c["replicas_identical"] is True
        assert c["all_reduce_exposed_ms"] >= 0 and c["exchange_bytes_per_rank"] > 0
    # four views from inside the room list most of the map: agree() leaves the row exchange for the dense slab, which goes
    # out in row chunks (56 bytes per surfel)
    assert c4["exchange_path"].startswith("dense") and "row chunks" in c4["exchange_path"] and c4["dense_chunks"] == 4
    assert c4["exchange_bytes_per_rank"] >= 56 * c4["surfels"]
    tl = c4["exchange_timeline_ms"]
    assert len(tl["all_reduce_per_chunk"]) == 4 and tl["chain_rule"] > 0 and tl["adam"] > 0
    assert c5["exchange_path"].startswith(("dense", "rows"))


def test_bench_single_rank_contract_fields(agslib):
    """The N=1 line the driver records: every field describes the timed region (whole replays of one graph), the
    derived rates, both roofline objects, and the parity of the HIP path against the oracle on the bench view."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5"], env=env,
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    # the JSON line is ALL there is on stdout (whatever libraries print - RCCL's banner, the mapper's "delete n gaussians" -
    # goes to stderr: bench.py hands file descriptor 1 to stderr and keeps the real stdout for this one line)
    assert [l for l in r.stdout.splitlines() if l.strip()] == lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["vs_baseline"] is None
    assert "20 step(s) per graph, 1 replay(s) per sample" in d["config"]["launch"]
    assert abs(d["value"] - 200_000 / (d["ms_per_step"] * 1e-3)) < 1e-3 * d["value"]
    # the line is one program, sampled: median of >= 21 samples of exactly K steps, min / max beside it; the other forms of
    # the same step and the other workloads are separate, labelled fields measured after the timed region
    assert d["samples"] >= 21 and d["ms_per_step_min"] <= d["ms_per_step"] <= d["ms_per_step_max"]
    # the headline computes in the reference's arithmetic (fp32 throughout); the bf16 hi/lo-split form of the blend backward
    # is an opt-in, timed as a labelled secondary field
    assert 0.7 * d["ms_per_step"] < d["ms_per_step_pipelined"] < 1.05 * d["ms_per_step"]
    assert 0.8 * d["ms_per_step"] < d["ms_per_step_bf16_split"] < 1.02 * d["ms_per_step"]
    assert "5 launches per step" in d["config"]["launch"]
    assert d["dtype"] == "f32"
    # the drop-in module: safe by default (every call checked before it returns), no synchronisation in the opt-in form
    dr = d["config"]["dropin"]
    assert 0 < d["config"]["dropin_ms_per_view"] < 1.0 and dr["c2_1200x680_1_view"]["module_event_waits_per_view"] == 1
    assert dr["c2_1200x680_1_view"]["module_syncs_per_view"] == 0 and dr["c2_1200x680_1_view_deferred"]["module_syncs_per_view"] == 0
    assert dr["c2_1200x680_1_view_deferred"]["module_event_waits_per_view"] == 0
    # configuration 3 (the mapper loop through the GaussianMap class) is in the driver's line
    c3 = d["config"]["c3"]
    assert c3["512x512"]["iterations"] == 500 and 0 < c3["seconds"] < 5 and c3["final_surfels"] > 50_000
    assert c3["1200x680"]["seconds"] > 0 and c3["512x512"]["overflow_retries"] == 0
    # where the loop's wall time goes comes from ONE run (event marks at the phase boundaries), not from a profiled run's
    # kernel time over this run's clock
    kb = c3["kernels_busy"]
    assert 0.4 < kb["gpu_bound_frac"] <= 1.0 and kb["phases"]["iterations"]["times"] == 50 and "kernels_busy_frac" not in kb
    # the drop-in figures are sampled like the headline
    for v in dr.values():
        if isinstance(v, dict):
            assert v["samples"] >= 9 and v["ms_per_view_min"] <= v["ms_per_view"] <= v["ms_per_view_max"]
    sec = d["config"]["secondary"]
    assert sec["c4_share_ms"] > 0 and sec["c5_ms"] > 0 and set(sec["c5"]["stage_hbm_frac"]) >= {"preprocess", "render_bwd"}
    # the one-rank end of the strong-scaled configurations 4 and 5 (what --gpus N reports as config.secondary.c4 / .c5)
    s4, s5 = sec["strong"]["c4"], sec["strong"]["c5"]
    assert s4["views_total"] == s4["views_this_rank"] == 32 and s4["surfels"] == 1_500_000 and s4["ms_per_step"] > 0
    assert s5["views_total"] == 8 and s5["surfels"] == 5_000_000 and s5["image"] == [2048, 2048] and s5["scaling"] == "strong"
    assert d["roofline"]["traffic_read"] is not None and d["roofline"]["traffic_write"] is not None
    assert d["roofline"]["c5"]["kernel"] == "render_bwd" and 0.05 < d["roofline"]["c5"]["frac"] < 0.5
    dr = d["config"]["derived_rates"]
    assert 0 < dr["visible_gaussians_per_s"] < dr["tile_instances_per_s"] < d["value"]
    assert d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1 and not d["config"]["overflow"]
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0
    p = d["parity"]
    assert p["ok"] and p["parity_rgb_L1"] < 1e-5 and p["parity_grad_rel"] < 3e-4 and p["tiles_compared"] > 100
    assert max(p["image_max_abs"].values()) < 3e-2 and max(p["image_worst_tile_L1"].values()) < 2.5e-4
    # the other workloads' headline figures are repeated as SCALAR keys of `config` (the driver's record keeps scalars only)
    cf = d["config"]
    assert cf["c3_seconds_512"] == c3["512x512"]["seconds"] and cf["c3_seconds_1200x680"] == c3["1200x680"]["seconds"]
    assert cf["c3_gpu_bound_frac"] == kb["gpu_bound_frac"] and cf["c4_strong_ms"] == s4["ms_per_step"] and cf["c5_strong_ms"] == s5["ms_per_step"]
    assert cf["dropin_ms_per_view_512"] > 0 and cf["c5_ms"] == sec["c5_ms"] and cf["ms_per_step_bf16x3"] == d["ms_per_step_bf16x3"]
    assert 0.8 * d["ms_per_step"] < d["ms_per_step_bf16x3"] < 1.1 * d["ms_per_step"]


def test_torch_free_cabi_demo(agslib):
    """examples/ags_cabi_demo.cpp: the C ABI used from plain C++/HIP (hipMalloc'd buffers, the whole
    step captured in a hipGraph and replayed) - no torch in the process."""
    from active_gs_amd import build
    exe = build.build_demo()
    r = subprocess.run([exe, "30000", "50"], capture_output=True, text=True, timeout=480)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.strip().endswith("OK") and "adam_steps=71" in r.stdout


def test_bench_check_mode_reports_the_exchange_path(agslib):
    """``bench.py --gpus 2 --check``: the readiness probe a multi-GPU node is asked before the timed run - process group
    up, ranks identified, one eager data-parallel step done, the captured-collective probe taken, every rank agreeing
    on the exchange path; exits in seconds with one JSON line (here: two ranks sharing the GPU over gloo, whose
    collectives cannot be recorded into a graph)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(AGS_BENCH_SHARE_GPU="1", AGS_BENCH_BACKEND="gloo", AGS_BENCH_WATCHDOG="150")
    env.update(AGS_BENCH_STRONG=STRONG_REDUCED)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--check"], env=env, capture_output=True,
                       text=True, timeout=400)
    if r.returncode != 0 and "Timeout" in r.stderr:
        _stall_report(r, "bench_check_stall.log")
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["check"] == "ok" and d["n_gpus"] == 2 and d["backend"] == "gloo" and d["refused_steps"] == 0
    assert d["exchange_path"] in ("rows: graph | collective | graph", "dense: graph | collective | graph")
    assert [e["rank"] for e in d["ranks"]] == [0, 1]
    # the path per strong-scaled configuration (one sizing pass and two eager steps each)
    sc = d["strong_configs"]
    assert sc["c4"]["exchange_path"].startswith("dense") and sc["c4"]["views_this_rank"] == 4 and sc["c4"]["replicas_identical"]
    assert sc["c5"]["exchange_path"].startswith(("dense", "rows")) and sc["c5"]["replicas_identical"]


def test_bench_check_mode_four_ranks(agslib):
    """The same readiness probe with FOUR ranks sharing the GPU over gloo (the dry run of profiles/r05_q_gloo_ranks.md kept
    in the suite): four identities, one exchange path agreed by all, the strong-scaled configurations dealt out four ways
    (configuration 4's 8 reduced views -> 2 per rank) with bit-identical replicas."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(AGS_BENCH_SHARE_GPU="1", AGS_BENCH_BACKEND="gloo", AGS_BENCH_WATCHDOG="200", AGS_BENCH_STRONG=STRONG_REDUCED)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--check"], env=env, capture_output=True,
                       text=True, timeout=500)
    if r.returncode != 0 and "Timeout" in r.stderr:
        _stall_report(r, "bench_check4_stall.log")
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["check"] == "ok" and d["n_gpus"] == 4 and d["backend"] == "gloo" and d["refused_steps"] == 0
    assert [e["rank"] for e in d["ranks"]] == [0, 1, 2, 3]
    sc = d["strong_configs"]
    assert sc["c4"]["views_this_rank"] == 2 and sc["c4"]["replicas_identical"] and sc["c5"]["views_this_rank"] == 1
    assert sc["c5"]["replicas_identical"]


def test_a_hung_rank_ends_the_job_with_stacks_not_a_silent_timeout(agslib):
    """The default watchdog of multi-rank runs: a rank that stalls dumps every thread's stack and exits non-zero, the
    launcher relays the tail of the job's stderr and returns non-zero itself.  Stall injected with AGS_BENCH_TEST_HANG
    (rank 1 sleeps before its first collective)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(AGS_BENCH_SHARE_GPU="1", AGS_BENCH_BACKEND="gloo", AGS_BENCH_WATCHDOG="25", AGS_BENCH_TEST_HANG="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--check"], env=env, capture_output=True,
                       text=True, timeout=400)
    assert r.returncode != 0
    assert "Timeout" in r.stderr and "exited with code" in r.stderr, r.stderr[-2000:]
