#!/usr/bin/env python3
"""Turn a rocprofv3 rocpd database (``*_results.db``) into the per-kernel stats table
(`--kernel-trace --stats` summary) as markdown.  Usage: rocpd_stats.py results.db > out.md"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
print("| kernel | calls | total (us) | avg (us) | % |")
print("|---|---:|---:|---:|---:|")
for name, calls, tot, avg, pct in rows:
    short = name.split("(")[0].replace("void ", "")
    print(f"| `{short}` | {calls} | {tot:.1f} | {avg:.3f} | {pct:.2f} |")
