#!/usr/bin/env python3
"""Copy the condensed evidence of one tools/prof_round.sh run from gpurun_out/<tag>/ into profiles/ (tracked) and
stamp profiles/traffic.json with the commit the box ran (the snapshot has no .git):

    python tools/prof_publish.py r02_a
"""
import json
import os
import shutil
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(REPO, "gpurun_out", tag)
dst = os.path.join(REPO, "profiles")
for name in ("bench.json", "kernel_stats.csv", "pmc.md"):
    if os.path.exists(os.path.join(src, name)):
        shutil.copy(os.path.join(src, name), os.path.join(dst, "%s_%s" % (tag, name)))
table = json.load(open(os.path.join(src, "traffic.json")))
table["kernels"] = [k for k in table["kernels"] if k["kernel"].endswith("_kernel") and "::" not in k["kernel"]]
head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=REPO).stdout.strip()
dirty = subprocess.run(["git", "status", "--porcelain", "--", "icnn_amd", "bench.py"], capture_output=True, text=True,
                       cwd=REPO).stdout.strip()
table["commit"] = head + ("+uncommitted" if dirty else "")
json.dump(table, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
print("published", tag, "at", table["commit"], [k["kernel"] for k in table["kernels"]])
