"""Condense an AGS_PARITY_LOG (one JSON line per comparison of the GPU suite, tests/_parity.py) into the worst margins:
python profiles/experiments/parity_summary.py gpurun_out/r03_parity_log.jsonl > profiles/r03_parity_margins.json"""
import collections, json, sys
rows = [json.loads(l) for l in open(sys.argv[1])]
img = collections.defaultdict(lambda: collections.defaultdict(lambda: (0.0, "")))
for r in rows:
    if r["kind"] == "images":
        for k, s in r["stats"].items():
            for m in ("mean", "max", "tile", "outliers"):
                if m in s and s[m] >= img[k][m][0]:
                    img[k][m] = (s[m], r["test"].split("::")[-1] + " | " + r["what"])
grads = collections.defaultdict(lambda: (0.0, ""))
for r in rows:
    if r["kind"] == "grads":
        for k, v in r["rel_L1"].items():
            if v >= grads[k][0]:
                grads[k] = (v, r["test"].split("::")[-1] + " | " + r["what"])
out = {"comparisons": len(rows),
       "images_worst": {k: {m: {"value": v[0], "where": v[1]} for m, v in d.items()} for k, d in img.items()},
       "grads_worst_rel_L1": {k: {"value": v[0], "where": v[1]} for k, v in grads.items()},
       "radii": [dict(test=r["test"].split("::")[-1], what=r["what"], rows=r["rows"], mismatch=r["mismatch"], ceil=r["ceil"],
                      rect=r["rect"], unexplained=r["unexplained"]) for r in rows if r["kind"] == "radii" and (r["mismatch"] or r["rows"] >= 100000)],
       "radii_comparisons": sum(r["kind"] == "radii" for r in rows), "radii_rows_total": sum(r["rows"] for r in rows if r["kind"] == "radii"),
       "radii_mismatch_total": sum(r["mismatch"] for r in rows if r["kind"] == "radii"),
       "count": [dict(test=r["test"].split("::")[-1], **{k: r[k] for k in ("rows", "rows_differ", "max_row_diff", "total", "total_diff")}) for r in rows if r["kind"] == "count"],
       "last_contributor": [dict(test=r["test"].split("::")[-1], what=r["what"], pixels=r["pixels"], differ=r["differ"]) for r in rows if r["kind"] == "last_contributor"]}
# the reference-capture replays (tests/test_gpu_densify.py, test_gpu_gaussian_map.py): worst margin per quantity over the
# keyframes, and the final row-by-row comparison (rows aligned by origin) - what CAPTURE_GATES are 10x of
cap = collections.defaultdict(float)
for r in rows:
    if r["kind"] == "capture":
        for k in ("rows", "perf_rel", "opacity_mean", "supports_rel", "scores_rel"):
            cap[k] = max(cap[k], abs(r[k]))
out["capture_worst"] = dict(cap)
out["capture_final"] = [dict(test=r["test"].split("::")[-1], rows_ref=r["rows_ref"], rows_mine=r["rows_mine"], common=r["common"],
                             common_frac=r["common_frac"], mean_abs_diff=r["mean_abs_diff"], pruned=r["pruned_mine"])
                        for r in rows if r["kind"] == "capture_final"]
out["densify_fixture"] = [dict(what=r["what"], rows_mine=r["rows_mine"], rows_ref=r["rows_ref"], paired=r["paired"],
                               max_abs_diff=r["max_abs_diff"]) for r in rows if r["kind"] == "densify_fixture"]
json.dump(out, sys.stdout, indent=1)
