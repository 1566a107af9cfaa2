#!/bin/bash
# work stealing between XCD bands: parity with it forced everywhere, then C5 / C4 A/B against the static split
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
AGS_STEAL_MIN_TILES=0 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_fullsize.py tests/test_gpu_pipeline.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4
export AGS_FREEZE=1
for v in 99999999 12288 99999999 12288; do
  echo "== min_tiles $v"; AGS_STEAL_MIN_TILES=$v python examples/large_configs.py --only c5 2>&1 | tail -1 | cut -c70-130
done
