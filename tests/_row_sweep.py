"""One oracle / fixture comparison per row of SURVEY.md section 8, run by ``__graft_entry__.smoke()`` after its own
forward+backward check: row-level evidence that survives a trip of the pytest run.  Every entry is one of the suite's own
GPU tests (called directly, on cuda:0, through the C ABI), chosen to finish in a few seconds; the sweep prints one line
per row and stops taking new rows when its time budget is spent (those rows read "not run")."""
import importlib
import time
import traceback

# (row, what is compared with what, test module, test function, keyword arguments)
ROWS = [
    ("a2/a3", "HIP forward + backward vs the committed oracle vectors, configuration 1 (5 k surfels, 300x170)",
     "test_gpu_golden", "test_hip_matches_committed_oracle_vectors", {"tag": "c1"}),
    ("a4/a7", "facade post-processing (normalise * mask, depth2normal incl. the fov/H pairing) vs the torch statements, 97x51",
     "test_gpu_fused_loss", "test_facade_post_kernel_matches_the_torch_statements", {"h": 97, "w": 51}),
    ("a5", "camera set-up vs the reference's GaussianRenderer.__init__ capture (camera.pt)",
     "test_cpu_host_logic", "test_camera_conventions_match_reference_renderer", None),
    ("a6", "render_view / render_view_all vs the reference renderer's capture (facade.pt), served from one batch",
     "test_gpu_gaussian_map", "test_batched_render_view_serves_the_reference_renderer_capture", {}),
    ("a8", "activations in registers vs the separate activation kernels",
     "test_gpu_parity", "test_fused_activations_match_separate_kernels", {}),
    ("a9/f1", "fused loss head (4 losses + depth->normal) vs torch autograd of the reference's statements",
     "test_gpu_fused_loss", "test_fused_loss_matches_torch_autograd", {}),
    ("a10", "fused Adam vs the torch.optim.Adam vector (adam.pt: eps 1e-15, zero-gradient rows)",
     "test_gpu_golden", "test_fused_adam_matches_torch_vector", {}),
    ("a11", "weighted frame draw kernel vs the torch statement",
     "test_gpu_densify", "test_weighted_frame_draw_kernel_equals_the_torch_statement", {}),
    ("a12", "post-processing counts: visible in the newest view (count >= 1) vs the oracle's counts",
     "test_gpu_consumers", "test_count_says_visible_in_the_newest_view", {}),
    ("a9/a10 loop", "train(): HIP rasterizer + fused loss + Adam vs the reference's GaussianMap.train() capture (train.pt)",
     "test_gpu_fused_loss", "test_fused_train_loop_matches_reference_train_capture", {}),
    ("f2", "add_gaussians + prune vs the reference's own outputs (densify.pt)",
     "test_gpu_densify", "test_add_gaussians_and_prune_match_reference_fixture", {}),
    ("f3", "planner-shaped batch of candidate views (ags_forward_batch) vs per-view renders",
     "test_gpu_gaussian_map", "test_planner_shaped_batch_of_candidate_views", {}),
    ("f4", "the reference's own checkpoint (map_ref.th) rendered at 1024x1024 vs the oracle",
     "test_gpu_fullsize", "test_reference_checkpoint_renders_at_mesh_resolution", {}),
    ("e", "row exchange tail (index + gathered Adam) vs unpack-per-rank + Adam, five simulated ranks, bit for bit",
     "test_gpu_distributed", "test_indexed_exchange_tail_equals_unpack_then_adam", {}),
]


def run(lib, budget_s: float = 55.0) -> list:
    """-> [(row, status, seconds, what)]; status "ok" | "FAIL: ..." | "not run (time budget)"."""
    out, t0 = [], time.perf_counter()
    for row, what, mod, fn, kw in ROWS:
        if time.perf_counter() - t0 > budget_s:
            out.append((row, "not run (time budget)", 0.0, what))
            continue
        t = time.perf_counter()
        try:
            f = getattr(importlib.import_module(mod), fn)
            f() if kw is None else f(lib, **kw)
            status = "ok"
        except Exception as e:  # noqa: BLE001 - the sweep reports every row
            status = f"FAIL: {type(e).__name__}: {str(e)[:200]}"
            traceback.print_exc()
        out.append((row, status, time.perf_counter() - t, what))
    return out


def report(rows) -> str:
    lines = ["smoke sweep: one comparison per SURVEY section-8 row", "| row | result | s | compared |", "|---|---|---:|---|"]
    lines += [f"| {r} | {s} | {t:.1f} | {w} |" for r, s, t, w in rows]
    return "\n".join(lines)
