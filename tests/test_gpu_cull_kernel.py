"""The cull-first per-Gaussian kernel (preprocess.hip: ags_k_preprocess_cull - large raw-parameter maps under one-pass
binning) against the plain kernel: it may only change WHO does the work, never a bit of the result.  Its conservative
cull (view depth, projected centre, a radius bound from max_scale) must keep every row the exact stage would keep; the
exact stage is the same code.  Both kernels are selected per process (AGS_PRE_CULL_MIN_N), hence two child processes
rendering the same scenes (tests/tools/cull_kernel_dump.py)."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_cull_first_kernel_is_bit_identical_to_the_plain_kernel(agslib, tmp_path):
    outs = {}
    for tag, thr in (("cull", "0"), ("plain", str(1 << 30))):
        f = str(tmp_path / f"{tag}.pt")
        env = dict(os.environ, AGS_PRE_CULL_MIN_N=thr)
        r = subprocess.run([sys.executable, os.path.join(HERE, "tools", "cull_kernel_dump.py"), f], env=env, capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        outs[tag] = torch.load(f)
    assert set(outs["cull"]) == set(outs["plain"]) and "large" in outs["cull"]
    for case, a in outs["cull"].items():
        b = outs["plain"][case]
        for k, va in a.items():
            vb = b[k]
            if torch.is_tensor(va):
                assert va.shape == vb.shape and torch.equal(va, vb), (case, k)
            else:
                assert va == vb, (case, k, va, vb)
    big = outs["cull"]["large"]
    assert 10_000 < big["visible"] < 600_000 and big["instances"] > big["visible"]      # the scene exercises both outcomes of the cull
