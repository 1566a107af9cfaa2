#!/usr/bin/env python3
"""Per-wave phase timelines of one optimisation step (C2 bench workload by default) from a -DAGS_TIMELINE build:
    python profiles/experiments/build_exp.py "tl=-DAGS_TIMELINE"      (here: cross-compiles)
    AGS_LIB_PATH=scratch/libags_tl.so python profiles/experiments/timeline.py [--n 200000 --h 680 --w 1200]
Every wave notes s_memtime (shader clock) at phase boundaries; this prints, per kernel, when waves start and end
relative to the kernel's first wave and how long each phase takes (median / p90 / max, in cycles and us @2.4 GHz)."""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=200_000)
ap.add_argument("--h", type=int, default=680)
ap.add_argument("--w", type=int, default=1200)
ap.add_argument("--room", default="office0")
ap.add_argument("--mult", type=float, default=1.0)
ap.add_argument("--binning", default="direct")
ap.add_argument("--pipeline", action="store_true", help="software-pipelined steps (4 launches)")
ap.add_argument("--graph", type=int, default=0, help="stamp the LAST step of a hipGraph of this many steps (0: one eager step)")
args = ap.parse_args()

from active_gs_amd import env_config
env_config.apply_env(os.environ)   # the package itself reads no environment variable
from active_gs_amd import _lib, raster_api as api  # noqa: E402
from active_gs_amd.camera import camera_matrices  # noqa: E402
from active_gs_amd.synthetic import make_camera, make_room_scene  # noqa: E402
from active_gs_amd.trainer import SurfelTrainer  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
raw = {k: v.to(dev) for k, v in make_room_scene(args.n, args.room, seed=0).items()}
if args.mult != 1.0:
    raw["scales"][:, :2] += float(np.log(args.mult))
c2w, K = make_camera(0, args.h, args.w, room=args.room)
cm = camera_matrices(c2w[None], K[None], 0.001, 10.0)
cam = api.Camera(args.h, args.w, cm["tanfov"][0, 0].item(), cm["tanfov"][0, 1].item(), cm["viewmatrix"][0].to(dev),
                 cm["projmatrix"][0].to(dev), torch.zeros(4, device=dev))
MODE = {"direct": api.BIN_DIRECT, "tile_sort": api.BIN_TILE_SORT}[args.binning]
tr = SurfelTrainer(raw, binning_mode=MODE)
probe = api.alloc_state(args.n, args.h, args.w, 40_000_000, dev, MODE)
api.forward(cam, tr.gaussians(), probe)
info = api.read_status(probe)
del probe
cap = int(info["needed"] * 1.3) + 4096
P = args.h * args.w
gen = torch.Generator().manual_seed(1234)
d_img = [(torch.randn(c, args.h, args.w, generator=gen) / P).to(dev) for c in (3, 3, 1)]
fn = lambda v, st: (d_img[0], d_img[1], d_img[2], None, None)
nxt = cam if args.pipeline else None
for _ in range(30):
    tr.step([cam], fn, cap, next_cam=nxt)
torch.cuda.synchronize()
NW = int(os.environ.get("AGS_TL_WAVES", 16384))      # (-DAGS_TL_WAVES=65536 for 2048x2048: 4 waves x 16 384 tiles)
buf = torch.zeros(8 * NW * 8, dtype=torch.int64, device=dev)
lib.ags_debug_timeline.argtypes = [C.c_void_p]
if args.graph:
    replay = tr.capture([cam], fn, cap, repeat=args.graph, pipeline=args.pipeline)
    replay(); torch.cuda.synchronize()
lib.ags_debug_timeline(buf.data_ptr())
if args.graph:
    replay()            # every step of the graph stamps the same slots: the last one's stamps survive
else:
    tr.step([cam], fn, cap, next_cam=nxt)
torch.cuda.synchronize()
lib.ags_debug_timeline(None)
t = buf.cpu().numpy().reshape(8, NW, 8).astype(np.int64)
if os.environ.get("AGS_TL_DUMP"):
    np.savez_compressed(os.environ["AGS_TL_DUMP"], t=t)
print(f"workload: n={args.n} {args.h}x{args.w} visible={info['num_visible']} instances={info['num_instances']}")
GHZ = 2.4
KERNELS = {0: ("preprocess", ["rows loaded", "math+stores", "row set", "emit(count)", "block sums"]),
           1: ("bucket", ["tile scan", "loads", "emit(keys)"]),
           5: ("tile_sort", ["sort"]),
           2: ("render_fwd", ["pixel init", "first staging", "blend loop", "stores"]),
           3: ("render_bwd_mfma", ["pixel loads", "feature exchange", "first staging", "blend loop", "last flush"]),
           4: ("preprocess_bwd_rows", ["all"])}


def pct(x, q):
    return float(np.percentile(x, q)) if len(x) else 0.0


t0_all = None
for kid, (name, phases) in KERNELS.items():
    a = t[kid]
    ok = a[:, 0] > 0
    if not ok.any():
        continue
    a = a[ok]
    nph = len(phases)
    # waves that returned early leave later stamps at 0: keep complete ones for the phase statistics
    full = a[a[:, nph] > 0]
    start0 = a[:, 0].min()
    end = np.where(a[:, 1:nph + 1] > 0, a[:, 1:nph + 1], 0).max(axis=1)
    print(f"\n== {name}: {len(a)} waves ({len(full)} complete); kernel span first start -> last end "
          f"{(end.max() - start0) / GHZ / 1e3:.2f} us (clock assumed {GHZ} GHz; per-XCD counters may be offset)")
    st = a[:, 0] - start0
    print(f"   wave start  p50 {pct(st, 50) / GHZ / 1e3:.2f}  p90 {pct(st, 90) / GHZ / 1e3:.2f}  max {st.max() / GHZ / 1e3:.2f} us")
    life = end - a[:, 0]
    print(f"   wave life   p50 {pct(life, 50) / GHZ / 1e3:.2f}  p90 {pct(life, 90) / GHZ / 1e3:.2f}  max {life.max() / GHZ / 1e3:.2f} us"
          f"   sum of lives / 1024 SIMDs = {life.sum() / 1024 / GHZ / 1e3:.2f} us")
    for p, label in enumerate(phases):
        d = full[:, p + 1] - full[:, p]
        print(f"   {label:18s} p50 {pct(d, 50):8.0f} cyc  p90 {pct(d, 90):8.0f}  max {d.max() if len(d) else 0:8.0f}   (p50 {pct(d, 50) / GHZ / 1e3:.2f} us)")
    if kid in (0, 4, 5):
        # workgroups go to the XCDs round-robin and every XCD has its own clock: per XCD, when waves start and end
        # relative to the XCD's first wave (the launch ramp and the tail of the one-round kernels)
        xcd = (a[:, 7] >> 32) & 15          # XCC_ID noted by the wave
        rel_s, rel_e, spans = [], [], []
        for x in range(8):
            sel = xcd == x
            if not sel.any():
                continue
            s0 = a[sel, 0].min()
            rel_s.append(a[sel, 0] - s0); rel_e.append(end[sel] - s0); spans.append((end[sel].max() - s0) / GHZ / 1e3)
        rs, re = np.concatenate(rel_s) / GHZ / 1e3, np.concatenate(rel_e) / GHZ / 1e3
        print(f"   per XCD (own clock): span first start -> last end {min(spans):.2f} .. {max(spans):.2f} us; wave start after the XCD's first "
              f"p50 {pct(rs, 50):.2f} p90 {pct(rs, 90):.2f} max {rs.max():.2f} us; wave end p50 {pct(re, 50):.2f} p90 {pct(re, 90):.2f} max {re.max():.2f} us")
    if kid in (2, 3):
        # where each wave ran (HW_REG_HW_ID | XCC_ID << 32, noted by the wave): per SIMD, how many waves were resident
        # on average between the SIMD's first start and last end, and how long that span was (one clock per XCD)
        hw = a[:, 7]
        simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7; xcc = (hw >> 32) & 15
        key = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
        spans, occ, first = [], [], []
        x0 = {}
        for x in np.unique(xcc):
            x0[x] = a[xcc == x, 0].min()
        for k in np.unique(key):
            sel = key == k
            s0, e1 = a[sel, 0].min(), end[sel].max()
            spans.append((e1 - s0) / GHZ / 1e3); occ.append(life[sel].sum() / max(e1 - s0, 1))
            first.append((s0 - x0[xcc[sel][0]]) / GHZ / 1e3)
        xs = [(end[xcc == x].max() - x0[x]) / GHZ / 1e3 for x in np.unique(xcc)]
        print(f"   {len(spans)} SIMDs on {len(x0)} XCDs saw waves; per XCD first start -> last end: {min(xs):.2f} .. {max(xs):.2f} us")
        print(f"   per SIMD: busy span p50 {pct(spans, 50):.2f} p10 {pct(spans, 10):.2f} p90 {pct(spans, 90):.2f} us; first wave starts "
              f"p50 {pct(first, 50):.2f} p90 {pct(first, 90):.2f} us after the XCD's first; resident waves (time average) "
              f"p50 {pct(occ, 50):.2f} p10 {pct(occ, 10):.2f} p90 {pct(occ, 90):.2f}; waves per SIMD p50 "
              f"{pct([int((key == k).sum()) for k in np.unique(key)], 50):.0f}")
    if kid in (2, 3):
        it = full[:, 6] & 0xFFFFFFFF
        loop = full[:, 3 if kid == 2 else 4] - full[:, 2 if kid == 2 else 3]
        print(f"   blended (surfel, wave) pairs per wave: mean {it.mean():.1f} p90 {pct(it, 90):.0f} max {it.max()}; "
              f"cycles per pair (loop / pairs): {loop.sum() / max(it.sum(), 1):.0f}")
        if kid == 3:
            fl = full[:, 6] >> 32
            print(f"   flushes per wave: mean {fl.mean():.2f}")
        else:
            print(f"   list length mean {full[:, 5].mean():.1f} max {full[:, 5].max()}")
