#!/usr/bin/env python3
"""Condense the rocprofv3 output of tools/prof_round.sh (run on the GPU box) into the files that get committed:

    <dir>/kernel_stats.csv   the kernel rows of rocprofv3's *_kernel_stats.csv (name, calls, total/avg/min/max ns, %)
    <dir>/pmc.md             FETCH_SIZE / WRITE_SIZE / SQ counters, mean per dispatch, per kernel
    <dir>/traffic.json       HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (FETCH_SIZE counts 64-byte
                             units as 32 on gfx950: MI355X_MICROARCH.md) -- what bench.py reports as roofline.traffic

    python tools/prof_collect.py gpurun_out/<tag> <tag>
"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

KEYS = ("fused_fc_solve_kernel", "fused_rows_solve_kernel", "dual_step_small_kernel", "dual_step_wide_kernel", "dual_step_kernel", "fc_fg_rows_kernel", "fc_fg_kernel",
        "conv_fwd_kernel", "conv_fc_fwd_kernel", "conv_fc_bwd_kernel", "conv_bwd_kernel", "state_init_kernel",
        "adam_rows_kernel", "adam_fc_kernel", "ctx_gemm_kernel", "ctx_bn_kernel", "clamp_kernel", "implicit_feed_kernel")


def short(name):
    for key in KEYS:
        if key in name:
            return key
    return name.split("(")[0][:60]


def find(root, pattern):
    hits = sorted(glob.glob(os.path.join(root, "**", pattern), recursive=True))
    return hits[0] if hits else None


def counters(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    if not path:
        return agg
    for r in csv.DictReader(open(path)):
        agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def main():
    root, tag = sys.argv[1], sys.argv[2]
    bench = {}
    try:
        bench = json.loads(open(os.path.join(root, "bench.json")).read().strip().splitlines()[-1])
    except Exception as exc:   # noqa: BLE001
        print("no bench line:", exc)
    stats = find(os.path.join(root, "stats"), "*kernel_stats.csv")
    if stats:
        rows = list(csv.reader(open(stats)))
        with open(os.path.join(root, "kernel_stats.csv"), "w") as fh:
            w = csv.writer(fh)
            w.writerow(rows[0])
            for r in rows[1:]:
                w.writerow([short(r[0])] + r[1:])
    fetch = counters(find(os.path.join(root, "pmc_FETCH_SIZE"), "*counter_collection.csv"))
    write = counters(find(os.path.join(root, "pmc_WRITE_SIZE"), "*counter_collection.csv"))
    sq = counters(find(os.path.join(root, "pmc_sq"), "*counter_collection.csv"))
    try:
        commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    except OSError:
        commit = ""
    cfg = bench.get("config", {})
    batch, n_iter = cfg.get("per_gpu_batch"), 10
    table = {"source": "profiles/%s_pmc.md" % tag, "commit": commit or "see profiles/README.md", "kernels": []}
    lines = ["# %s: PMC passes on `python bench.py --steps 2 --warmup 1 --cpu-sample 0 --c4-steps 0`" % tag, "",
             "Separate `rocprofv3 --kernel-trace --pmc` passes (tools/prof_round.sh): FETCH_SIZE alone, WRITE_SIZE alone, one SQ "
             "pass.  HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE correction, MI355X_MICROARCH.md).", "",
             "| kernel | dispatches | FETCH_SIZE KiB | WRITE_SIZE KiB | HBM bytes / launch |", "|---|---|---|---|---|"]
    for k in sorted(set(fetch) | set(write)):
        f = fetch[k].get("FETCH_SIZE", [])
        wv = write[k].get("WRITE_SIZE", [])
        if not f or not wv:
            continue
        fm, wm = sum(f) / len(f), sum(wv) / len(wv)
        hbm = (2 * fm + wm) * 1024
        lines.append("| %s | %d | %.0f | %.0f | %.4g |" % (k, len(f), fm, wm, hbm))
        if k in KEYS:
            table["kernels"].append({"kernel": k, "batch": batch, "n_iter": n_iter, "hbm_bytes_per_launch": hbm,
                                 "fetch_kib": fm, "write_kib": wm})
    lines += ["", "SQ pass, mean per dispatch:", ""]
    names = sorted({c for k in sq for c in sq[k]})
    if names:
        lines.append("| kernel | " + " | ".join(names) + " |")
        lines.append("|---|" + "---|" * len(names))
        for k in sorted(sq):
            lines.append("| %s | " % k + " | ".join("%.4g" % (sum(sq[k][c]) / len(sq[k][c])) if sq[k][c] else "" for c in names) + " |")
    if bench:
        lines += ["", "bench line of the same build: value %.4g %s, ms_per_step %.4f, rank0_solve_ms %s, roofline.frac %s"
                  % (bench.get("value", 0), bench.get("unit", ""), bench.get("ms_per_step", 0),
                     bench.get("rank0_solve_ms"), bench.get("roofline", {}).get("frac"))]
    open(os.path.join(root, "pmc.md"), "w").write("\n".join(lines) + "\n")
    json.dump(table, open(os.path.join(root, "traffic.json"), "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
