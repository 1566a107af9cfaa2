"""Turns the calibration run's counters into per-pattern factors: algorithmic bytes / counter bytes.
usage: fetch_calibration_summary.py bytes.json fetch.csv write.csv kernel_stats.csv tag"""
import csv
import json
import sys
from collections import defaultdict

alg = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])


def load(path, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][0] += float(r["Counter_Value"]) * 1024.0      # KiB -> bytes
        acc[k][1] += 1
    return {k: v[0] / v[1] for k, v in acc.items()}


fetch, write = load(sys.argv[2], "FETCH_SIZE"), load(sys.argv[3], "WRITE_SIZE")
dur = {}
try:
    for r in csv.DictReader(open(sys.argv[4])):
        dur[r["Name"].split("(")[0].replace("void ", "")] = float(r["AverageNs"]) * 1e-3
except Exception as e:  # noqa: BLE001
    print(f"(no kernel durations: {e})")
print(f"# {sys.argv[5]}: FETCH_SIZE / WRITE_SIZE against known byte counts (1 GiB arrays, {alg['_touches']} random touches per gather / "
      "scatter kernel; profiles/experiments/fetch_calibration.hip)\n")
print("| kernel | algorithmic read MB | FETCH_SIZE MB | read factor (alg / counter) | algorithmic write MB | WRITE_SIZE MB | write factor | avg us | algorithmic GB/s |")
print("|---|---:|---:|---:|---:|---:|---:|---:|---:|")
for k, (r, w) in alg.items():
    if k.startswith("_"):
        continue
    f, ws, us = fetch.get(k, 0.0), write.get(k, 0.0), dur.get(k)
    rf = f"{r / f:.2f}" if r and f else "-"
    wf = f"{w / ws:.2f}" if w and ws else "-"
    gbs = f"{(r + w) / (us * 1e-6) / 1e9:.0f}" if us else "-"
    print(f"| `{k}` | {r / 1e6:.1f} | {f / 1e6:.1f} | {rf} | {w / 1e6:.1f} | {ws / 1e6:.1f} | {wf} | {us if us else '-'} | {gbs} |")
