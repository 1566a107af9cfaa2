#!/bin/bash
# the driver's multi-GPU command shapes dry-run on ONE GPU: 8, 4 and 2 ranks sharing it over gloo, reduced strong sizes
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
export AGS_BENCH_SHARE_GPU=1 AGS_BENCH_BACKEND=gloo AGS_BENCH_WATCHDOG=400 AGS_BENCH_STRONG="c4=150000,32,680,1200;c5=400000,8,1024,1024"
for n in 8 4 2; do
  python bench.py --gpus $n --steps 20 --warmup 3 --no-cpu-baseline > $O/r06_gloo_$n.out 2> $O/r06_gloo_$n.err
  echo "== --gpus $n: rc=$? stdout lines $(wc -l < $O/r06_gloo_$n.out)"
  python - $O/r06_gloo_$n.out <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l); c = d["config"]; s = c.get("secondary", {})
        print(dict(n_gpus=d["n_gpus"], ms_per_step=round(d["ms_per_step"], 4), value=round(d["value"] / 1e6), launch=c["launch"][:60],
                   ranks=len(c["exchange"]["ranks"]), refused=c["exchange"]["refused_steps"]))
        for k in ("c4", "c5"):
            v = s.get(k)
            print(" ", k, v if not isinstance(v, dict) else {x: v.get(x) for x in ("views_this_rank", "ms_per_step", "exchange_path", "replicas_identical")})
PY
done
