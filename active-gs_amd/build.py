"""In-tree build of libags_raster.so with hipcc for gfx950 (no torch types, no JIT cache).

The shared object lands in ``active-gs_amd/lib/`` next to the sources, so it travels with
a snapshot of the repository; hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libags_raster.so")
SOURCES = ["preprocess.hip", "binning.hip", "render.hip", "adam.hip", "loss.hip", "densify.hip", "capi.hip"]
HEADERS = ["ags_internal.h", "ags_experiments.h", "surfel_math.h", "loss_pixel.h", os.path.join("..", "..", "include", "ags_raster.h")]
# loss.hip must reproduce exact cancellations of the reference's un-fused torch ops (see ags_point)
# render.hip: -fno-signed-zeros lets the compiler fold the `0 + x` of freshly zeroed accumulators
# (-2.5 % step time); NaN / inf semantics are left alone.
# densify.hip: same reason as loss.hip - depth2normal relies on (p_neighbour - p_centre) being exactly
# 0 at replicated borders, which an fma-contracted difference of products is not.
# preprocess.hip / adam.hip: fp32 divides and square roots as v_rcp / v_sqrt sequences (<= 2.5 ulp) instead of
# the correctly rounded ~10-instruction expansions: 34 divisions in the per-Gaussian backward alone.
EXTRA_FLAGS = {"loss.hip": ["-ffp-contract=off"], "densify.hip": ["-ffp-contract=off"],
               "render.hip": ["-fno-signed-zeros"],
               # -ffp-contract=on: multiply-adds are fused where the SOURCE writes a * b + c in one expression and
               # nowhere else, so the per-Gaussian stage rounds the same in every kernel it is inlined into (the
               # pipelined step runs it from ags_k_rows_adam_preprocess: with the default cross-statement fusion the
               # two kernels' records differed in the last bit of a few fields)
               "preprocess.hip": ["-fno-hip-fp32-correctly-rounded-divide-sqrt", "-ffp-contract=on"],
               "adam.hip": ["-fno-hip-fp32-correctly-rounded-divide-sqrt"]}
# -fno-slp-vectorize everywhere: hipcc's SLP pass packs pairs of scalar fp32 ops into v_pk_* at the
# price of register-pair shuffles (25 % v_mov in render_bwd) and VGPRs; scalar code measured
# faster in every kernel of this library (step -7.5 %).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-fno-slp-vectorize", "-Wall",
         "-Wno-unused-function", "-Wno-unused-value"]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build libags_raster.so")
    return exe


STAMP = LIB + ".srchash"


def source_digest() -> str:
    """SHA-256 over the contents of every source, header and this file (the compile flags live here): what the
    library was built from, independent of file times (a checkout can give a stale .so a newer mtime than its sources)."""
    import hashlib
    h = hashlib.sha256()
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.normpath(os.path.join(CSRC, x)) for x in HEADERS]
    deps.append(os.path.abspath(__file__))
    for d in deps:
        h.update(os.path.basename(d).encode())
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def is_stale() -> bool:
    """True unless the library exists and was built from exactly the current sources and flags (content digest kept
    next to it; the digest file travels with the .so)."""
    if not os.path.exists(LIB) or not os.path.exists(STAMP):
        return True
    with open(STAMP) as f:
        return f.read().strip() != source_digest()


# (Until round 3 a second library, libags_raster_f32exact.so, carried the blend backward with exact f32 matrix
# instructions.  Both forms now live in the ONE library as two instantiations of ags_k_render_bwd_mfma, selected per
# workspace with AgsTuning.bwd_reduce - exact f32 is the default.)
VARIANT_FLAGS = {}   # tag -> {source: extra flags}: experiment builds only (profiles/experiments)


def _link(hipcc, objs, out):
    tmp = out + ".tmp"
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", tmp],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr}")
    os.replace(tmp, out)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 and link libags_raster.so.  Returns the library's path."""
    if not force and not is_stale():
        return LIB
    hipcc = _hipcc()
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)

    def compile_one(job):
        src, extra, suffix = job
        obj = os.path.join(objdir, src.replace(".hip", suffix + ".o"))
        cmd = [hipcc, *FLAGS, *EXTRA_FLAGS.get(src, []), *extra, "-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr}")
        if verbose and r.stderr:
            print(r.stderr)
        return obj

    jobs = [(s, [], "") for s in SOURCES] + [(s, fl, "_" + tag) for tag, m in VARIANT_FLAGS.items() for s, fl in m.items()]
    with ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, jobs))
    main_objs = objs[:len(SOURCES)]
    _link(hipcc, main_objs, LIB)
    stale = os.path.join(LIBDIR, "libags_raster_f32exact.so")   # the round-3 variant library: gone
    if os.path.exists(stale):
        os.remove(stale)
    with open(STAMP, "w") as f:
        f.write(source_digest() + "\n")
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))


DEMO_SRC = os.path.normpath(os.path.join(HERE, "..", "examples", "ags_cabi_demo.cpp"))
DEMO_BIN = os.path.normpath(os.path.join(HERE, "..", "examples", "ags_cabi_demo"))


def build_demo() -> str:
    """Torch-free C++ user of the C ABI (examples/ags_cabi_demo.cpp), linked against the in-tree library."""
    build()
    if os.path.exists(DEMO_BIN) and os.path.getmtime(DEMO_BIN) >= max(os.path.getmtime(DEMO_SRC), os.path.getmtime(LIB)):
        return DEMO_BIN
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O2", "-std=c++17", "-I", os.path.normpath(os.path.join(HERE, "..", "include")),
           DEMO_SRC, "-L", LIBDIR, "-lags_raster", "-Wl,-rpath,$ORIGIN/../active-gs_amd/lib", "-o", DEMO_BIN]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on the C ABI demo:\n{r.stderr}")
    return DEMO_BIN


# ---- the drop-in module's native host side (csrc/torch_binding.cpp): a torch extension module built in-tree with g++
# (no device code in it: it calls the C ABI) against the torch of this image; it travels with the repository snapshot
# like the library does.
BINDING_SRC = os.path.join(CSRC, "torch_binding.cpp")
BINDING_NAME = "ags_torch_binding"
BINDING = os.path.join(LIBDIR, BINDING_NAME + ".so")
BINDING_STAMP = BINDING + ".srchash"


def binding_digest() -> str:
    import hashlib
    import torch
    h = hashlib.sha256()
    for d in (BINDING_SRC, os.path.normpath(os.path.join(CSRC, "..", "..", "include", "ags_raster.h"))):
        with open(d, "rb") as f:
            h.update(f.read())
    h.update(torch.__version__.encode())
    h.update(b"flags-v1")
    return h.hexdigest()


def binding_is_stale() -> bool:
    if not os.path.exists(BINDING) or not os.path.exists(BINDING_STAMP):
        return True
    with open(BINDING_STAMP) as f:
        return f.read().strip() != binding_digest()


def build_torch_binding(force: bool = False) -> str:
    """Compile csrc/torch_binding.cpp into lib/ags_torch_binding.so.  Returns its path."""
    if not force and not binding_is_stale():
        return BINDING
    import sysconfig
    import torch
    import torch.utils.cpp_extension as ce
    os.makedirs(LIBDIR, exist_ok=True)
    cxx = shutil.which("g++") or "g++"
    inc = [*ce.include_paths(), sysconfig.get_paths()["include"], "/opt/rocm/include",
           os.path.normpath(os.path.join(HERE, "..", "include"))]
    libdirs = ce.library_paths() + ["/opt/rocm/lib"]
    tmp = BINDING + ".tmp"
    cmd = [cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", f"-DTORCH_EXTENSION_NAME={BINDING_NAME}",
           "-DTORCH_API_INCLUDE_EXTENSION_H", "-Wno-deprecated-declarations", BINDING_SRC, "-o", tmp,
           *[f"-I{i}" for i in inc], *[f"-L{d}" for d in libdirs], *[f"-Wl,-rpath,{d}" for d in libdirs],
           "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch_hip", "-ltorch", "-ltorch_python", "-lamdhip64", "-ldl"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"g++ failed on torch_binding.cpp:\n{r.stderr[-6000:]}")
    os.replace(tmp, BINDING)
    with open(BINDING_STAMP, "w") as f:
        f.write(binding_digest() + "\n")
    return BINDING
