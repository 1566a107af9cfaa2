import os, sys, json, importlib.util, faulthandler
faulthandler.enable()
os.environ.setdefault("AGS_DP_FORCE", "1")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29578")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch, torch.distributed as dist
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from active_gs_amd import env_config
env_config.apply_env(os.environ)   # the package itself reads no environment variable
from active_gs_amd.trainer import RowExchange
def agree_dense(self, local_rows, slab_floats):      # what agree() decides at 8 ranks for configuration 4: the dense slab
    self.agreements += 1
    self.capacity = 0
    return 0
RowExchange.agree = agree_dense
spec = importlib.util.spec_from_file_location("bench_module", os.path.join(root, "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev)
for key in (sys.argv[1:] or ["c4", "c5"]):
    cfg = bench.strong_configs()[key]
    r = bench.measure_strong(key, cfg, 10, dev, 1, 0, True)
    print(json.dumps({k: r.get(k) for k in ("exchange_path", "dense_chunks", "ms_per_step", "launch", "all_reduce_exposed_ms", "exchange_timeline_ms", "replicas_identical")}), flush=True)
dist.destroy_process_group()
