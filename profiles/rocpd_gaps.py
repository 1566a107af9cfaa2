#!/usr/bin/env python3
"""Where the GPU idles: from a rocprofv3 rocpd database (--kernel-trace), the idle time between consecutive kernel
dispatches (all queues merged), attributed to the pair (kernel that ended, kernel that started).
Usage: rocpd_gaps.py results.db [min_gap_us=5] [top=25] [first_kernel_substring[@k]: analyse from its first launch on
(@k: from its k-th launch; negative k counts from the end, e.g. ags_k_bilateral@-50 = the last 50 keyframes of a mapper loop)]"""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
tables = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
view = "kernels" if "kernels" in tables else None
if view is None:
    print("tables:", tables)
    sys.exit(1)
cols = [r[1] for r in db.execute(f"pragma table_info({view})")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = list(db.execute(f"select {name_col}, start, end from {view} order by start"))
short = lambda n: n.split("(")[0].replace("void ", "")[:60]
if len(sys.argv) > 4:
    pat, _, kth = sys.argv[4].partition("@")
    hits = [i for i, r in enumerate(rows) if pat in r[0]]
    rows = rows[hits[int(kth) if kth else 0]:]
busy = sum(e - s for _, s, e in rows) / 1e6
busy_end = rows[0][2]
t0, t1 = rows[0][1], max(r[2] for r in rows)
gaps = collections.defaultdict(lambda: [0.0, 0])
prev = rows[0][0]
idle = 0.0
for n, s, e in rows[1:]:
    if s > busy_end:
        g = (s - busy_end) / 1e3
        idle += g
        if g >= min_gap:
            k = (short(prev), short(n))
            gaps[k][0] += g; gaps[k][1] += 1
    if e > busy_end:
        busy_end, prev = e, n
print(f"{len(rows)} launches, kernel time {busy:.2f} ms (sum); span {(t1 - t0) / 1e6:.2f} ms, idle {idle / 1e3:.2f} ms in gaps; gaps >= {min_gap} us by (ended -> started):")
print("| ended | started | gaps | total (ms) | mean (us) |")
print("|---|---|---:|---:|---:|")
for (a, b), (tot, n) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:top]:
    print(f"| `{a}` | `{b}` | {n} | {tot / 1e3:.2f} | {tot / n:.1f} |")
