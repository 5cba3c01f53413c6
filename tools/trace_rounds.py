#!/usr/bin/env python3
"""Print the per-launch timeline of the last solve in a rocprofv3 --kernel-trace CSV."""
import csv
import sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "icnn_be" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# split into solves at state_init_kernel
solves, cur = [], []
for r in rows:
    if "state_init" in r["Kernel_Name"]:
        if cur:
            solves.append(cur)
        cur = []
    cur.append(r)
solves.append(cur)
last = solves[-1]
t0 = int(last[0]["Start_Timestamp"])
print("launches in last solve:", len(last))
prev_end = t0
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = "fg  " if "fc_fg" in r["Kernel_Name"] else ("dual" if "dual_step" in r["Kernel_Name"] else "init")
    print("%s start %8.1f us  dur %7.1f us  gap %6.1f us" % (name, (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3))
    prev_end = e
print("total %.1f us" % ((prev_end - t0) / 1e3))
