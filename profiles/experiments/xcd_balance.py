"""Per-XCD work of the blend backward under other band partitions, from the same kind of dump as xcd_residency.py (the
build must also stamp tile / list length / maxlast per wave - see r03_xcd_work_stealing.patch's session notes):
    python profiles/experiments/xcd_balance.py x.npz <number of tiles>"""
import numpy as np, sys
t = np.load(sys.argv[1])["t"]
a = t[3]; m = t[6]
ok = a[:, 0] > 0
end = np.where(a[:, 1:6] > 0, a[:, 1:6], 0).max(axis=1)
life = np.where(ok, (end - a[:, 0]) * 10.0 / 1e3, 0.0)
tile = (m[:, 0] & 0xFFFFFFFF).astype(np.int64); cnt = (m[:, 0] >> 32).astype(np.int64); ml = m[:, 1].astype(np.int64)
W = len(a)
slot = np.arange(W) // 4
ntile = int(sys.argv[2])
# per tile work (sum of its four waves' lives), count, maxlast-sum
work = np.zeros(ntile); c = np.zeros(ntile); mls = np.zeros(ntile)
valid = ok & (m[:, 0] > 0)
np.add.at(work, tile[valid], life[valid]); np.maximum.at(c, tile[valid], cnt[valid]); np.add.at(mls, tile[valid], ml[valid])
print("tiles with work", (work > 0).sum(), "total wave-time/1024 %.1f us" % (work.sum() / 1024))
def bands_equal(n):
    q, r = n >> 3, n & 7
    b = [0]
    for x in range(8): b.append(b[-1] + q + (1 if x < r else 0))
    return b
def bands_by(wgt):
    cs = np.cumsum(wgt); tot = cs[-1]
    b = [0] + [int(np.searchsorted(cs, tot * x / 8)) for x in range(1, 8)] + [len(wgt)]
    return b
for name, b in (("equal tiles", bands_equal(ntile)), ("equal counts", bands_by(c)), ("equal maxlast", bands_by(mls)), ("equal count+16", bands_by(c + 16)), ("oracle(equal work)", bands_by(work))):
    w = np.array([work[b[x]:b[x + 1]].sum() for x in range(8)]) / 128
    sizes = [b[x + 1] - b[x] for x in range(8)]
    print(f"{name:20s} per-XCD work (us of 128 SIMDs x resident): max/mean {w.max() / w.mean():.3f}  sizes max/mean {max(sizes) / (ntile / 8):.2f}  ", np.round(w, 0).tolist())
