#!/usr/bin/env python3
"""Fraction of the mapper loop's wall time during which at least one kernel runs, from a rocprofv3 kernel trace of
examples/mapper_loop.py: the TIMED loop is the last 50 keyframes' worth of launches - it starts at the last launch of
ags_k_bilateral (growth's first kernel) that is followed by 49 more, i.e. the 50th-from-last.  -> JSON on stdout.
usage: mapper_busy.py results.db [session tag]"""
import json
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = list(db.execute(f"select {name_col}, start, end from kernels order by start"))
marks = [i for i, r in enumerate(rows) if "ags_k_bilateral" in r[0]]
keyframes = 50
first = marks[-keyframes] if len(marks) >= keyframes else 0
rows = rows[first:]
t0, t1 = rows[0][1], max(r[2] for r in rows)
busy, cur_s, cur_e = 0, rows[0][1], rows[0][2]
for _, s, e in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(json.dumps(dict(kernels_busy_frac=round(busy / (t1 - t0), 4), span_ms=round((t1 - t0) / 1e6, 2), busy_ms=round(busy / 1e6, 2),
                      launches=len(rows), keyframes=min(keyframes, len(marks)), _session=sys.argv[2] if len(sys.argv) > 2 else "?",
                      _note="union of kernel intervals / span from the first growth kernel of the timed loop to the last kernel "
                            "(rocprofv3 --kernel-trace of examples/mapper_loop.py; the profiler itself slows the host side)")))
