#!/usr/bin/env python3
"""Memory-side bytes per kernel from the L2's request-size counters (gfx950), not from FETCH_SIZE's fixed 64 bytes:
  read  = 128 * TCC_EA0_RDREQ_128B + 64 * TCC_EA0_RDREQ_64B + 32 * TCC_EA0_RDREQ_32B
  write = 64 * TCC_EA0_WRREQ_64B + 32 * (TCC_EA0_WRREQ - TCC_EA0_WRREQ_64B)          (atomics: TCC_EA0_ATOMIC, 32 B each by WRITE_SIZE)
Calibrated in profiles/r06_fetch_calibration.md: FETCH_SIZE = 64 B x TCC_EA0_RDREQ whatever a request moves, so it is exact
for isolated 64-byte sectors (random record gathers) and half the bytes wherever whole 128-byte lines are consumed (every
coalesced stream, 4 to 16 bytes per lane, and the blend kernels' 32-byte image row segments).

usage: pmc_exact_summary.py <dir with *_counter_collection.csv of the passes> [out.json [session-tag]] [--all-kernels]"""
import csv
import glob
import json
import sys
from collections import defaultdict

STAGE = {"ags_k_preprocess_bwd": "preprocess_bwd", "ags_k_rows_multi": "preprocess_bwd", "ags_k_preprocess": "preprocess",
         "ags_k_scan_tiles": "binning", "ags_k_bucket": "binning", "ags_k_tile_sort": "binning", "ags_k_render_fwd": "render_fwd",
         "ags_k_render_bwd": "render_bwd", "ags_k_adam": "adam"}


def stage_of(k):
    for pat, st in STAGE.items():
        if k.startswith(pat):
            return st
    return None


args = [a for a in sys.argv[1:] if not a.startswith("--")]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for p in sorted(glob.glob(args[0] + "/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        a = acc[k][r["Counter_Name"].replace("_sum", "")]
        a[0] += float(r["Counter_Value"]); a[1] += 1
rows = {}
for k, d in acc.items():
    g = lambda n: d[n][0] / d[n][1] if n in d and d[n][1] else 0.0    # noqa: E731
    launches = max(v[1] for v in d.values())
    rd = 128 * g("TCC_EA0_RDREQ_128B") + 64 * g("TCC_EA0_RDREQ_64B") + 32 * g("TCC_EA0_RDREQ_32B")
    other = g("TCC_EA0_RDREQ") - g("TCC_EA0_RDREQ_128B") - g("TCC_EA0_RDREQ_64B") - g("TCC_EA0_RDREQ_32B")
    wr = 64 * g("TCC_EA0_WRREQ_64B") + 32 * (g("TCC_EA0_WRREQ") - g("TCC_EA0_WRREQ_64B"))
    rows[k] = dict(launches=launches, read=rd, write=wr, rdreq=g("TCC_EA0_RDREQ"), r128=g("TCC_EA0_RDREQ_128B"), r64=g("TCC_EA0_RDREQ_64B"),
                   r32=g("TCC_EA0_RDREQ_32B"), unsized=other, wrreq=g("TCC_EA0_WRREQ"), w64=g("TCC_EA0_WRREQ_64B"), atomics=g("TCC_EA0_ATOMIC"),
                   fetch_size_equiv=64 * g("TCC_EA0_RDREQ"))
show_all = "--all-kernels" in sys.argv
print("| kernel | launches | read MB (exact) | 64 B x RDREQ (= FETCH_SIZE) MB | 128-B / 64-B / 32-B read requests | write MB | 64-B / 32-B write requests | memory-side atomics |")
print("|---|---:|---:|---:|---|---:|---|---:|")
for k in sorted(rows, key=lambda k: -(rows[k]["read"] + rows[k]["write"])):
    if not show_all and not k.startswith("ags_k"):
        continue
    r = rows[k]
    print(f"| `{k}` | {r['launches']} | {r['read'] / 1e6:.3f} | {r['fetch_size_equiv'] / 1e6:.3f} | {r['r128']:.4g} / {r['r64']:.4g} / {r['r32']:.4g} | "
          f"{r['write'] / 1e6:.3f} | {r['w64']:.4g} / {r['wrreq'] - r['w64']:.4g} | {r['atomics']:.4g} |")
if len(args) > 1:
    steps = defaultdict(int)
    for k, r in rows.items():
        if stage_of(k):
            steps[stage_of(k)] = max(steps[stage_of(k)], r["launches"])
    out = defaultdict(lambda: dict(read=0.0, write=0.0, fetch_raw=0.0))
    for k, r in rows.items():
        st = stage_of(k)
        if st:
            w = r["launches"] / steps[st]      # one-off launches outside the steps do not count as a kernel of the stage
            out[st]["read"] += r["read"] * w; out[st]["write"] += r["write"] * w; out[st]["fetch_raw"] += r["fetch_size_equiv"] * w
    res = {k: {"read": round(v["read"]), "write": round(v["write"]), "traffic": round(v["read"] + v["write"]), "fetch_raw": round(v["fetch_raw"])}
           for k, v in out.items()}
    res["_session"] = args[2] if len(args) > 2 else "?"
    res["_note"] = ("bytes per step, summed over the kernels of a bench stage weighted by their launches per step; rocprofv3 --pmc passes of "
                    "the L2's memory-side request counters BY SIZE: read = 128 x TCC_EA0_RDREQ_128B + 64 x _64B + 32 x _32B, write = 64 x "
                    "TCC_EA0_WRREQ_64B + 32 x the other write requests (profiles/r06_fetch_calibration.md: FETCH_SIZE counts every read "
                    "request at 64 bytes); traffic = read + write; fetch_raw = 64 x TCC_EA0_RDREQ = what FETCH_SIZE reports")
    json.dump(res, open(args[1], "w"), indent=1)
