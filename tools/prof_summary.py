#!/usr/bin/env python3
"""Turn rocprofv3 output (results .db from --kernel-trace --stats, or the --pmc
counter_collection.csv) into the small text summaries committed under profiles/.

    python tools/prof_summary.py kernels  <results.db>             > profiles/<name>.md
    python tools/prof_summary.py counters <counter_collection.csv> > profiles/<name>.md
"""
import collections
import csv
import sqlite3
import sys


def short(name):
    for key in ("fused_fc_solve_kernel", "dual_step_kernel", "fc_fg_kernel", "conv_fg_kernel", "state_init_kernel"):
        if key in name:
            return name[name.index(key):].split("(")[0]
    return name.split("(")[0][:70]


def kernels(db):
    c = sqlite3.connect(db)
    rows = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    print("| kernel | calls | total us | avg us | % |\n|---|---|---|---|---|")
    for name, calls, total, avg, pct in rows[:12]:
        print("| %s | %d | %.1f | %.2f | %.2f |" % (short(name), calls, total, avg, pct))
    cur = c.execute("select name, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, workgroup_x, grid_x "
                    "from kernels where name like '%icnn_be%' group by name")
    print("\n| kernel | VGPR | AGPR | SGPR | LDS bytes | workgroup | grid |\n|---|---|---|---|---|---|---|")
    for r in cur:
        print("| %s | %s | %s | %s | %s | %s | %s |" % ((short(r[0]),) + tuple(r[1:])))


def counters(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        if "icnn_be" in r["Kernel_Name"]:
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        n = len(next(iter(v.values())))
        print("### %s  (%d dispatches, mean per dispatch)\n" % (k, n))
        print("| counter | mean |\n|---|---|")
        for cname in sorted(v):
            print("| %s | %.4g |" % (cname, sum(v[cname]) / len(v[cname])))
        print()


if __name__ == "__main__":
    {"kernels": kernels, "counters": counters}[sys.argv[1]](sys.argv[2])
