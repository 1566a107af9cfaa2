#!/usr/bin/env python3
"""How much of a multi-view step's kernels overlap: from a rocprofv3 --kernel-trace rocpd database, the kernels of the
last few steps by stream/queue with start and end, the fraction of the busy time with >= 2 kernels in flight, and the sum
of kernel durations against the time from the first start to the last end.
    rocprofv3 --kernel-trace -d out -o x -- python3 examples/large_configs.py --only c4
    python3 profiles/experiments/stream_overlap.py out/x_results.db"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if "kernel_dispatch" in t and "rocpd" in t][0]
cols = [r[1] for r in db.execute(f"pragma table_info('{kd}')")]
names = {}
ks = [t for t in tabs if "kernel_symbol" in t or "info_kernel" in t]
for t in ks:
    c = [r[1] for r in db.execute(f"pragma table_info('{t}')")]
    if "kernel_name" in c and "id" in c:
        for i, n in db.execute(f"select id, kernel_name from '{t}'"):
            names[i] = n.split("(")[0].replace("void ", "")
sel = "kernel_id" if "kernel_id" in cols else "kernel_symbol_id"
qcol = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else None)
rows = list(db.execute(f'select {sel}, start, "end", {qcol or 0} from "{kd}" order by start'))
rows = [(names.get(k, str(k)), s, e, q) for k, s, e, q in rows]
# the last 3 steps: from the third-last ags_k_rows_multi / rows kernel's end
ends = [i for i, r in enumerate(rows) if "rows" in r[0]]
if len(ends) >= 4:
    rows = rows[ends[-4] + 1: ends[-1] + 1]
t0 = rows[0][1]
ev = []
for n, s, e, q in rows:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
busy = over = 0
live = 0; last = ev[0][0]
for t, d in ev:
    if live >= 1: busy += t - last
    if live >= 2: over += t - last
    live += d; last = t
span = max(r[2] for r in rows) - t0
tot = sum(r[2] - r[1] for r in rows)
print(f"kernels {len(rows)}; first start -> last end {span/1e3:.1f} us; sum of kernel durations {tot/1e3:.1f} us; "
      f"some kernel running {busy/1e3:.1f} us; two or more running {over/1e3:.1f} us ({100*over/max(busy,1):.0f} % of that)")
print("| start (us) | end (us) | stream/queue | kernel |"); print("|---:|---:|---:|---|")
for n, s, e, q in rows[: 3 * 17 + 2][:40]:
    print(f"| {(s-t0)/1e3:.1f} | {(e-t0)/1e3:.1f} | {q} | `{n[:50]}` |")
