#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, csv) per kernel.

Usage: pmc_summary.py fetch_counter_collection.csv write_counter_collection.csv [out.json [session-tag]]
FETCH_SIZE / WRITE_SIZE are in KiB (TCC_EA0 request counters); per
/opt/skills/guides/MI355X_MICROARCH.md §HBM, FETCH_SIZE counts 128-B requests of wide
(16 B/lane) coalesced streams at 64 B on gfx950, so the read side of such kernels is up to 2x
the raw figure — both the raw and the doubled read bytes are printed; WRITE_SIZE is
uncalibrated (it matches 12 B/element exactly on the Adam kernel, whose FETCH_SIZE x2 also
matches its 16 B/element of reads).  `out.json` maps bench stage -> {fetch_raw, write,
traffic = 2*fetch_raw + write} bytes per launch of its kernels."""
import csv
import json
import sys
from collections import defaultdict

STAGE = {"ags_k_preprocess<": "preprocess", "ags_k_scan_tiles": "binning", "ags_k_bucket": "binning",
         "ags_k_tile_sort": "binning", "ags_k_render_fwd": "render_fwd", "ags_k_render_bwd": "render_bwd",
         "ags_k_preprocess_bwd": "preprocess_bwd", "ags_k_adam": "adam"}


def load(path, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][0] += float(r["Counter_Value"])
        acc[k][1] += 1
    return {k: v[0] / v[1] * 1024.0 for k, v in acc.items()}


fetch = load(sys.argv[1], "FETCH_SIZE")
write = load(sys.argv[2], "WRITE_SIZE")
print("| kernel | FETCH raw (MB) | FETCH x2 (MB) | WRITE (MB) |")
print("|---|---:|---:|---:|")
stage_bytes = defaultdict(lambda: [0.0, 0.0])
for k in sorted(fetch, key=lambda k: -fetch[k]):
    if not k.startswith("ags_k"):
        continue
    w = write.get(k, 0.0)
    print(f"| `{k}` | {fetch[k] / 1e6:.3f} | {2 * fetch[k] / 1e6:.3f} | {w / 1e6:.3f} |")
    for pat, st in STAGE.items():
        if k.startswith(pat.rstrip("<")) and (pat != "ags_k_preprocess<" or "bwd" not in k):
            stage_bytes[st][0] += fetch[k]
            stage_bytes[st][1] += w
            break
if len(sys.argv) > 3:
    out = {k: {"fetch_raw": round(v[0]), "write": round(v[1]), "traffic": round(2 * v[0] + v[1])}
           for k, v in stage_bytes.items()}
    out["_session"] = sys.argv[4] if len(sys.argv) > 4 else "?"
    out["_note"] = ("bytes per launch, summed over the kernels of a bench stage (bench.py --eager under rocprofv3 --pmc, "
                    "FETCH_SIZE and WRITE_SIZE in separate passes); traffic = 2*fetch_raw + write (gfx950 FETCH_SIZE "
                    "counts 128-B requests of wide streaming reads at 64 B: MI355X_MICROARCH.md, HBM section)")
    json.dump(out, open(sys.argv[3], "w"), indent=1)
