"""ctypes wrapper of tests/host_emu (test-only sequential driver of surfel_math.h)."""
import ctypes
import os
import subprocess

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "host_emu", "emu.cpp")
LIB = os.path.join(HERE, "host_emu", "libags_emu.so")
HDR = os.path.join(ROOT, "active-gs_amd", "csrc", "surfel_math.h")


def build():
    if (not os.path.exists(LIB)) or os.path.getmtime(LIB) < max(os.path.getmtime(SRC), os.path.getmtime(HDR)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off",
                               "-I", os.path.dirname(HDR), SRC, "-o", LIB])
    return ctypes.CDLL(LIB)


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def forward(S, means, scales, rots, opac, colors, conf):
    lib = build()
    lib.emu_forward.restype = ctypes.c_long
    H, W, N = S.image_height, S.image_width, means.shape[0]
    cfg = S.config.tolist()
    f = lambda *s: torch.zeros(*s, dtype=torch.float32)
    out = dict(rgb=f(3, H, W), normal=f(3, H, W), depth=f(1, H, W), opacity=f(1, H, W), confidence=f(1, H, W),
               final_T=f(H, W), n_contrib=torch.zeros(H, W, dtype=torch.int32), importance=f(N),
               count=torch.zeros(N, dtype=torch.int32), radii=torch.zeros(N, dtype=torch.int32), geom=f(N, 16))
    mask = None
    if S.render_mask is not None and S.render_mask.numel() > 0:
        mask = S.render_mask.float().contiguous()
    keep = [t.contiguous().float() for t in (S.viewmatrix, S.projmatrix, S.bg, means, scales, rots,
                                             opac.reshape(-1), colors, conf)]
    V, P, bg, means, scales, rots, opac, colors, conf = keep
    I = lib.emu_forward(H, W, ctypes.c_float(S.tanfovx), ctypes.c_float(S.tanfovy), ctypes.c_float(S.scale_modifier),
                        int(cfg[1] > 0), int(cfg[2] > 0), int(cfg[4] > 0), int(cfg[3] > 0),
                        ctypes.c_float(S.weight_thres), _p(mask), _p(V), _p(P), _p(bg), N, _p(means), _p(scales),
                        _p(rots), _p(opac), _p(colors), _p(conf), _p(out["rgb"]), _p(out["normal"]),
                        _p(out["depth"]), _p(out["opacity"]), _p(out["confidence"]), _p(out["final_T"]),
                        _p(out["n_contrib"]), _p(out["importance"]), _p(out["count"]), _p(out["radii"]),
                        _p(out["geom"]))
    out["num_rendered"] = I
    return out


def backward(S, means, scales, rots, opac, colors, conf, fwd, d_rgb, d_normal, d_depth, d_opacity, d_conf):
    lib = build()
    H, W, N = S.image_height, S.image_width, means.shape[0]
    cfg = S.config.tolist()
    f = lambda *s: torch.zeros(*s, dtype=torch.float32)
    g = dict(means=f(N, 3), scales=f(N, 3), rots=f(N, 4), opac=f(N), colors=f(N, 3), means2d=f(N, 3), dgeom=f(N, 16))
    keep = [t.contiguous().float() for t in (S.viewmatrix, S.projmatrix, S.bg, means, scales, rots, opac.reshape(-1),
                                             colors, conf, d_rgb, d_normal, d_depth, d_opacity, d_conf)]
    V, P, bg, means, scales, rots, opac, colors, conf, d_rgb, d_normal, d_depth, d_opacity, d_conf = keep
    lib.emu_backward(H, W, ctypes.c_float(S.tanfovx), ctypes.c_float(S.tanfovy), ctypes.c_float(S.scale_modifier),
                     int(cfg[1] > 0), int(cfg[2] > 0), int(cfg[4] > 0), _p(V), _p(P), _p(bg), N, _p(means),
                     _p(scales), _p(rots), _p(opac), _p(colors), _p(conf), _p(fwd["depth"]), _p(fwd["opacity"]),
                     _p(fwd["final_T"]), _p(fwd["n_contrib"]), _p(d_rgb), _p(d_normal), _p(d_depth), _p(d_opacity),
                     _p(d_conf), _p(g["means"]), _p(g["scales"]), _p(g["rots"]), _p(g["opac"]), _p(g["colors"]),
                     _p(g["means2d"]), _p(g["dgeom"]))
    return g
