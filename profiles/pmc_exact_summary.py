#!/usr/bin/env python3
"""Memory-side bytes per kernel from the L2's request counters (gfx950).
  write = 64 * TCC_EA0_WRREQ_64B + 32 * (TCC_EA0_WRREQ - TCC_EA0_WRREQ_64B)      exact (calibration: streams, 64-byte and 8-byte scatters)
  read  : BRACKETED.  FETCH_SIZE = 64 B x TCC_EA0_RDREQ whatever a request moves (profiles/r06_fetch_calibration.md): exact for
          isolated 64-byte sectors, half the bytes wherever whole 128-byte lines are consumed.  The size-split counters do
          not settle it: TCC_EA0_RDREQ_128B tallies EVERY read request, also those of a random 64-byte gather (whose request
          rate - 55 G/s - would mean 7.0 TB/s at 128 bytes each, more than a pure stream reaches on the same box).  So
          read_lo = 64 x RDREQ  <=  bytes read  <=  read_hi = 128 x RDREQ,
          and `read` is the end the stage's access pattern sits at: read_hi for the stages that consume whole lines (every
          coalesced stream and the blend kernels' image rows and id lists), read_lo for the one gather-only stage
          (the per-Gaussian backward over the member rows: 64-byte records and 12/16-byte fields by row id).

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
    rd = 128 * g("TCC_EA0_RDREQ")           # read_hi; read_lo = fetch_size_equiv below
    other = g("TCC_EA0_RDREQ") - g("TCC_EA0_RDREQ_128B") - g("TCC_EA0_RDREQ_64B") - g("TCC_EA0_RDREQ_32B")
    wr = 64 * g("TCC_EA0_WRREQ_64B") + 32 * (g("TCC_EA0_WRREQ") - g("TCC_EA0_WRREQ_64B"))
    rows[k] = dict(launches=launches, read=rd, write=wr, rdreq=g("TCC_EA0_RDREQ"), r128=g("TCC_EA0_RDREQ_128B"), r64=g("TCC_EA0_RDREQ_64B"),
                   r32=g("TCC_EA0_RDREQ_32B"), unsized=other, wrreq=g("TCC_EA0_WRREQ"), w64=g("TCC_EA0_WRREQ_64B"), atomics=g("TCC_EA0_ATOMIC"),
                   fetch_size_equiv=64 * g("TCC_EA0_RDREQ"))
show_all = "--all-kernels" in sys.argv
print("| kernel | launches | read_hi MB (128 B x RDREQ) | read_lo MB (64 B x RDREQ = FETCH_SIZE) | read requests tallied as 128-B / 64-B / 32-B | write MB (exact) | 64-B / 32-B write requests | memory-side atomics |")
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
    GATHER_ONLY = ("preprocess_bwd",)          # stages whose reads are isolated sectors by row id: read = read_lo
    out = defaultdict(lambda: dict(read_hi=0.0, write=0.0, read_lo=0.0, atomics=0.0))
    for k, r in rows.items():
        st = stage_of(k)
        if st:
            w = r["launches"] / steps[st]      # one-off launches outside the steps do not count as a kernel of the stage
            out[st]["read_hi"] += r["read"] * w; out[st]["write"] += r["write"] * w; out[st]["read_lo"] += r["fetch_size_equiv"] * w
            out[st]["atomics"] += r["atomics"] * w
    res = {}
    for k, v in out.items():
        rd = v["read_lo"] if k in GATHER_ONLY else v["read_hi"]
        res[k] = {"read": round(rd), "read_lo": round(v["read_lo"]), "read_hi": round(v["read_hi"]), "write": round(v["write"]),
                  "traffic": round(rd + v["write"]), "atomics": round(v["atomics"]), "fetch_raw": round(v["read_lo"])}
    res["_session"] = args[2] if len(args) > 2 else "?"
    res["_note"] = ("bytes per step, summed over the kernels of a bench stage weighted by their launches per step; rocprofv3 --pmc passes of "
                    "the L2's memory-side request counters: write = 64 x TCC_EA0_WRREQ_64B + 32 x the other write requests (exact); read "
                    "is bracketed by read_lo = 64 x TCC_EA0_RDREQ (= FETCH_SIZE: exact for isolated 64-byte sectors) and read_hi = 128 x "
                    "TCC_EA0_RDREQ (exact where whole 128-byte lines are consumed); `read` = read_hi, except for the gather-only stage "
                    "preprocess_bwd (read_lo) - profiles/r06_fetch_calibration.md; traffic = read + write")
    json.dump(res, open(args[1], "w"), indent=1)
