#!/usr/bin/env python3
"""Condense tools/prof_shapes.sh's output: for every shape directory under <root> write <root>/<tag>_<shape>_kernel_stats.csv
(kernel rows of rocprofv3's stats) and <root>/<tag>_<shape>_pmc.md (FETCH_SIZE / WRITE_SIZE / SQ counters per kernel, mean per
dispatch; HBM bytes per launch = (2 FETCH_SIZE + WRITE_SIZE) 1024 with the gfx950 FETCH_SIZE correction of
MI355X_MICROARCH.md), plus <root>/traffic_rows.json: rows for profiles/traffic.json keyed by (kernel, batch, n_iter).

    python tools/prof_collect_shapes.py gpurun_out/<tag> <tag>      (then copy the files into profiles/)"""
import csv
import json
import os
import shutil
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from prof_collect import KEYS, counters, find, short  # noqa: E402

KEYS_EXTRA = ("dual_step_small_kernel", "dual_step_wide_kernel", "mark_unfinished_kernel", "ctx_bn_sums_kernel", "ctx_bn_apply_kernel")


def main():
    root, tag = sys.argv[1], sys.argv[2]
    rows_out = []
    for shape in sorted(os.listdir(root)):
        d = os.path.join(root, shape)
        if not os.path.isdir(d) or not os.path.exists(os.path.join(d, "run.json")):
            continue
        try:
            run = json.loads(open(os.path.join(d, "run.json")).read().strip().splitlines()[-1])
        except (ValueError, IndexError):
            run = {"shape": shape}
        stats = find(os.path.join(d, "stats"), "*kernel_stats.csv")
        if stats:
            rows = list(csv.reader(open(stats)))
            with open(os.path.join(root, "%s_%s_kernel_stats.csv" % (tag, shape)), "w") as fh:
                w = csv.writer(fh)
                w.writerow(rows[0])
                for r in rows[1:]:
                    if "icnn_be" in r[0]:
                        w.writerow([name_of(r[0])] + r[1:])
        fetch = counters(find(os.path.join(d, "pmc_FETCH_SIZE"), "*counter_collection.csv"))
        write = counters(find(os.path.join(d, "pmc_WRITE_SIZE"), "*counter_collection.csv"))
        sq = counters(find(os.path.join(d, "pmc_sq"), "*counter_collection.csv"))
        lines = ["# %s / %s: `python tools/prof_target.py %s` -- %s" % (tag, shape, shape, json.dumps(run)), "",
                 "Separate `rocprofv3 --kernel-trace --pmc` passes (tools/prof_shapes.sh): FETCH_SIZE alone, WRITE_SIZE alone, one SQ "
                 "pass.  HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024.", "",
                 "| kernel | dispatches | FETCH_SIZE KiB | WRITE_SIZE KiB | HBM bytes / launch |", "|---|---|---|---|---|"]
        for k in sorted(set(fetch) | set(write)):
            f, wv = fetch[k].get("FETCH_SIZE", []), write[k].get("WRITE_SIZE", [])
            if not f or not wv or not (k in KEYS or k in KEYS_EXTRA):
                continue
            fm, wm = sum(f) / len(f), sum(wv) / len(wv)
            hbm = (2 * fm + wm) * 1024
            lines.append("| %s | %d | %.0f | %.0f | %.4g |" % (k, len(f), fm, wm, hbm))
            rows_out.append({"kernel": k, "batch": run.get("batch"), "n_iter": run.get("n_iter"), "shape": shape,
                             "hbm_bytes_per_launch": hbm, "fetch_kib": fm, "write_kib": wm})
        names = sorted({c for k in sq for c in sq[k]})
        if names:
            lines += ["", "SQ pass, mean per dispatch:", "", "| kernel | " + " | ".join(names) + " |", "|---|" + "---|" * len(names)]
            for k in sorted(sq):
                if k in KEYS or k in KEYS_EXTRA:
                    lines.append("| %s | " % k + " | ".join("%.4g" % (sum(sq[k][c]) / len(sq[k][c])) if sq[k][c] else "" for c in names) + " |")
        open(os.path.join(root, "%s_%s_pmc.md" % (tag, shape)), "w").write("\n".join(lines) + "\n")
        for sub in ("stats", "pmc_FETCH_SIZE", "pmc_WRITE_SIZE", "pmc_sq"):       # raw traces stay on the box (size)
            shutil.rmtree(os.path.join(d, sub), ignore_errors=True)
        print("\n".join(lines[:3] + lines[5:]))
    json.dump(rows_out, open(os.path.join(root, "traffic_rows.json"), "w"), indent=1)


def name_of(full):
    for key in KEYS_EXTRA:
        if key in full:
            return key
    return short(full)


if __name__ == "__main__":
    main()
