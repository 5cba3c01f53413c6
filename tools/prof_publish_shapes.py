#!/usr/bin/env python3
"""Copy the condensed per-shape evidence of one tools/prof_shapes.sh run (gpurun_out/<dir>/<dir>_<shape>_*.{csv,md}) into
profiles/ as <tag>_<shape>_* and replace those shapes' rows of profiles/traffic.json:

    python tools/prof_publish_shapes.py r03_e_shapes r03_e
"""
import json
import os
import shutil
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src_dir, tag = sys.argv[1], sys.argv[2]
src = os.path.join(REPO, "gpurun_out", src_dir)
dst = os.path.join(REPO, "profiles")
shapes = []
for name in sorted(os.listdir(src)):
    if name.startswith(src_dir + "_") and name.endswith(("_kernel_stats.csv", "_pmc.md")):
        out = tag + name[len(src_dir):]
        text = open(os.path.join(src, name)).read().replace(src_dir, tag)
        open(os.path.join(dst, out), "w").write(text)
        shapes.append(out)
rows = json.load(open(os.path.join(src, "traffic_rows.json")))
table = json.load(open(os.path.join(dst, "traffic.json")))
new_shapes = {r["shape"] for r in rows}
table["kernels"] = [k for k in table["kernels"] if k.get("shape") not in new_shapes] + rows
head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=REPO).stdout.strip()
table["commit"] = head
table["source"] = (table.get("source", "") + "; " if table.get("source") else "") + "%s_{%s}_pmc.md @ %s" % (tag, ",".join(sorted(new_shapes)), head)
json.dump(table, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
print("published", shapes)
