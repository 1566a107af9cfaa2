"""Resident waves per SIMD over the blend launches and per-XCD end times / wave-time, from a timeline dump made with a
-DAGS_TIMELINE -DAGS_TL_REALTIME -DAGS_TL_WAVES=65536 build (AGS_TL_DUMP=x.npz AGS_TL_WAVES=65536 python timeline.py ...):
    python profiles/experiments/xcd_residency.py x.npz"""
import numpy as np, sys
t = np.load(sys.argv[1])["t"]
NS = 10.0
for kid, nph, name in ((2, 4, "fwd"), (3, 5, "bwd")):
    a = t[kid]; a = a[a[:, 0] > 0]
    if not len(a): continue
    end = np.where(a[:, 1:nph + 1] > 0, a[:, 1:nph + 1], 0).max(axis=1)
    s0 = a[:, 0].min(); e1 = end.max()
    span = (e1 - s0) * NS / 1e3
    life = (end - a[:, 0]) * NS / 1e3
    print(f"{name}: waves {len(a)} span {span:.1f} us; sum life/1024 = {life.sum()/1024:.1f} us -> mean resident per SIMD {life.sum()/1024/span:.2f}; life p50 {np.percentile(life,50):.1f} p90 {np.percentile(life,90):.1f}")
    # concurrency over time (chip-wide), 20 buckets
    ev = np.concatenate([np.stack([a[:, 0], np.ones(len(a))], 1), np.stack([end, -np.ones(len(a))], 1)])
    ev = ev[np.argsort(ev[:, 0], kind="stable")]
    conc = np.cumsum(ev[:, 1]); tt = (ev[:, 0] - s0) * NS / 1e3
    edges = np.linspace(0, span, 21)
    out = []
    for i in range(20):
        sel = (tt >= edges[i]) & (tt < edges[i + 1])
        out.append(round(float(conc[sel].mean() / 1024), 2) if sel.any() else 0)
    print("   resident waves per SIMD over the span (20 buckets):", out)
    hw = a[:, 7]
    simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7; xcc = (hw >> 32) & 15
    key = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
    u, cnt = np.unique(key, return_counts=True)
    print(f"   SIMDs seen {len(u)}; waves per SIMD min {cnt.min()} p50 {int(np.median(cnt))} max {cnt.max()}")
    # per XCD: last end
    for x in np.unique(xcc):
        sel = xcc == x
        print(f"   xcc {x}: waves {sel.sum()} first start {(a[sel,0].min()-s0)*NS/1e3:.1f} last end {(end[sel].max()-s0)*NS/1e3:.1f} sum life/128 {life[sel].sum()/128:.1f}")
