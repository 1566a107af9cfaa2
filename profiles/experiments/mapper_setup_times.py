"""Host time (no synchronisation) of the pieces FusedMapTrainer._train_batched runs before its first iteration, per
train() call of the mapper loop:  python profiles/experiments/mapper_setup_times.py"""
import sys, os, time, collections
sys.path.insert(0, os.getcwd())
import torch
sys.argv = [sys.argv[0]]
import importlib.util
spec = importlib.util.spec_from_file_location("ml", "examples/mapper_loop.py"); ml = importlib.util.module_from_spec(spec); spec.loader.exec_module(ml)
from active_gs_amd import env_config
env_config.apply_env(os.environ)   # the package itself reads no environment variable
from active_gs_amd import fused_map_trainer as fmt, optimizer, trainer, raster_api as api, map_trainer
acc = collections.defaultdict(float); cnt = collections.defaultdict(int)
def wrap(obj, name, label):
    fn = getattr(obj, name)
    def w(*a, **k):
        t0 = time.perf_counter(); r = fn(*a, **k); acc[label] += time.perf_counter() - t0; cnt[label] += 1
        return r
    setattr(obj, name, w)
wrap(optimizer.FusedAdam, "__init__", "FusedAdam()"); wrap(trainer.GradSlab, "__init__", "GradSlab()"); wrap(api.RowSet, "__init__", "RowSet()")
wrap(fmt, "make_frame_sampler", "make_frame_sampler"); wrap(fmt.FusedMapTrainer, "_frame_store", "_frame_store"); wrap(fmt.FusedMapTrainer, "_gaussians", "_gaussians")
wrap(api.ViewBatch, "bind", "ViewBatch.bind"); wrap(api.ViewBatch, "forward", "batch.forward (per iteration)"); wrap(api.ViewBatch, "backward", "batch.backward (per iteration)"); wrap(api, "backward_rows", "backward_rows (per iteration)")
from active_gs_amd import fused_loss
for nm in ("stage_frames", "stage1_batch", "stage2_batch", "finish", "set_batch_total"):
    wrap(fused_loss.FusedLoss, nm, "loss." + nm + " (per iteration)"); wrap(fmt.FusedMapTrainer, "_uniform_frames", "_uniform_frames"); wrap(fmt.FusedMapTrainer, "_snapshot", "_snapshot")
wrap(fmt, "weighted_choice_into", "sampler draw (per iteration)"); wrap(fmt.FusedMapTrainer, "_train_batched", "_train_batched (whole, incl. waits)")
wrap(fmt.FusedMapTrainer, "post_processing", "post_processing (incl. waits)"); wrap(fmt.FusedMapTrainer, "add_gaussians", "add_gaussians (incl. waits)")
wrap(fmt.FusedMapTrainer, "_make_camera", "_make_camera"); wrap(api.ViewBatch, "statuses", "ViewBatch.statuses (wait)")
ml.main()
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"{k:40s} {1e3 * v:8.1f} ms total {cnt[k]:5d} calls {1e3 * v / cnt[k]:7.3f} ms/call")
