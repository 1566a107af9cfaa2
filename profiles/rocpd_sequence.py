#!/usr/bin/env python3
"""The kernel sequence of one period of a loop from a rocprofv3 rocpd database (--kernel-trace): every dispatch between the
k-th and (k+1)-th launch of a marker kernel, with its start offset, duration and the idle gap in front of it.
Usage: rocpd_sequence.py results.db marker_substring k   (k may be negative: counted from the end)"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = list(db.execute(f"select {name_col}, start, end from kernels order by start"))
pat, k = sys.argv[2], int(sys.argv[3])
hits = [i for i, r in enumerate(rows) if pat in r[0]]
a = hits[k]
b = hits[k + 1] if (k + 1 < len(hits) and k + 1 != 0) else len(rows)
short = lambda n: n.split("(")[0].replace("void ", "")[:70]
t0 = rows[a][1]
end = rows[a][1]
print(f"{b - a} dispatches between launch {k} and {k + 1} of {pat}: span {(rows[b - 1][2] - t0) / 1e3:.1f} us, "
      f"kernel time {sum(e - s for _, s, e in rows[a:b]) / 1e3:.1f} us")
print("| # | start (us) | gap before (us) | duration (us) | kernel |")
print("|---:|---:|---:|---:|---|")
for i, (n, s, e) in enumerate(rows[a:b]):
    print(f"| {i} | {(s - t0) / 1e3:.1f} | {max(0, s - end) / 1e3:.1f} | {(e - s) / 1e3:.1f} | `{short(n)}` |")
    end = max(end, e)
