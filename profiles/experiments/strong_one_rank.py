import os, sys, json, importlib.util, faulthandler
faulthandler.enable()
os.environ.setdefault("AGS_DP_FORCE", "1")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch, torch.distributed as dist
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
spec = importlib.util.spec_from_file_location("bench_module", os.path.join(root, "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev)
which = sys.argv[1:] or ["c4", "c5"]
for key in which:
    cfg = bench.strong_configs()[key]
    print("==", key, cfg, file=sys.stderr, flush=True)
    r = bench.measure_strong(key, cfg, 10, dev, 1, 0, True)
    print(json.dumps({k: r[k] for k in ("exchange_path", "ms_per_step", "launch", "all_reduce_exposed_ms", "replicas_identical")}), flush=True)
dist.destroy_process_group()
