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
    """kernel -> (bytes per launch, launches)"""
    acc = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][0] += float(r["Counter_Value"])
        acc[k][1] += 1
    return {k: (v[0] / v[1] * 1024.0, v[1]) for k, v in acc.items()}


def stage_of(k):
    for pat, st in STAGE.items():
        if k.startswith(pat.rstrip("<")) and (pat != "ags_k_preprocess<" or "bwd" not in k):
            return st
    return None


fetch = load(sys.argv[1], "FETCH_SIZE")
write = load(sys.argv[2], "WRITE_SIZE")
print("| kernel | launches | FETCH raw (MB) | FETCH x2 (MB) | WRITE (MB) |")
print("|---|---:|---:|---:|---:|")
# A stage's bytes per STEP: every kernel of the stage weighted by how often it ran per step (launches / the stage's
# most-launched kernel): the one-off launches outside the steps (the bench's workspace probe runs the general tile
# sort once) do not count as a second kernel of the stage.
steps = defaultdict(int)
for k, (_, n) in fetch.items():
    if stage_of(k):
        steps[stage_of(k)] = max(steps[stage_of(k)], n)
stage_bytes = defaultdict(lambda: [0.0, 0.0])
for k in sorted(fetch, key=lambda k: -fetch[k][0]):
    if not k.startswith("ags_k"):
        continue
    f, n = fetch[k]
    w = write.get(k, (0.0, 0))[0]
    print(f"| `{k}` | {n} | {f / 1e6:.3f} | {2 * f / 1e6:.3f} | {w / 1e6:.3f} |")
    st = stage_of(k)
    if st:
        stage_bytes[st][0] += f * n / steps[st]
        stage_bytes[st][1] += w * n / steps[st]
if len(sys.argv) > 3:
    out = {k: {"fetch_raw": round(v[0]), "write": round(v[1]), "traffic": round(2 * v[0] + v[1])}
           for k, v in stage_bytes.items()}
    out["_session"] = sys.argv[4] if len(sys.argv) > 4 else "?"
    out["_note"] = ("bytes per step, summed over the kernels of a bench stage weighted by their launches per step (bench.py --eager under rocprofv3 --pmc, "
                    "FETCH_SIZE and WRITE_SIZE in separate passes); traffic = 2*fetch_raw + write (gfx950 FETCH_SIZE "
                    "counts 128-B requests of wide streaming reads at 64 B: MI355X_MICROARCH.md, HBM section)")
    json.dump(out, open(sys.argv[3], "w"), indent=1)
