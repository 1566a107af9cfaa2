import os, subprocess, sys
sys.path.insert(0, '.')
from active_gs_amd import env_config
env_config.apply_env(os.environ)   # the package itself reads no environment variable
import active_gs_amd.build as b
def build(tag, defs):
    objdir = f"scratch/exp_{tag}"; os.makedirs(objdir, exist_ok=True)
    objs = []
    for src in b.SOURCES:
        obj = os.path.join(objdir, src.replace('.hip', '.o'))
        cmd = [b._hipcc(), *b.FLAGS, *b.EXTRA_FLAGS.get(src, []), *defs, '-c', os.path.join(b.CSRC, src), '-o', obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        objs.append(obj)
    out = f"scratch/libags_{tag}.so"
    r = subprocess.run([b._hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', *objs, '-o', out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    print(out)
for tag, defs in [a.split('=', 1) for a in sys.argv[1:]]:
    build(tag, [d for d in defs.split(',') if d])
