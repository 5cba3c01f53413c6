#!/usr/bin/env python3
"""Registers, spills and scratch of every kernel of the library, read from the compiler (no GPU needed):
    python tools/kernel_resources.py [be_fused.hip be_dual.hip ...] > profiles/rNN_kernel_resources.md
Compiles each translation unit with the flags of icnn_amd/build.py plus -Rpass-analysis=kernel-resource-usage and tabulates the
remarks (VERDICT r5 #3b: report vgpr_spill_count per instance)."""
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from icnn_amd import build  # noqa: E402

units = sys.argv[1:] or build.SOURCES
hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
print("| kernel | VGPRs | spilled VGPRs | spilled SGPRs | scratch B/lane | waves/SIMD |\n|---|---|---|---|---|---|")
for unit in units:
    with tempfile.TemporaryDirectory() as tmp:
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wno-pass-failed",
               "-I" + build.INCLUDE, "-I" + build.CSRC] + build.EXTRA_FLAGS.get(unit, []) + \
              ["-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(build.CSRC, unit), "-o", os.path.join(tmp, "x.o")]
        text = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
    cur = {}
    rows = []
    for line in text.splitlines():
        m = re.search(r"remark:\s+(Function Name|VGPRs|SGPRs Spill|VGPRs Spill|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]): (\S+)", line)
        if not m:
            continue
        key, val = m.group(1), m.group(2)
        if key == "Function Name":
            cur = {"name": val}
            rows.append(cur)
        else:
            cur[key] = val
    names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), stdout=subprocess.PIPE, text=True).stdout.splitlines()
    for r, nm in zip(rows, names):
        nm = nm.replace("icnn_be::(anonymous namespace)::", "").replace("void ", "")
        nm = re.sub(r"\(.*\)$", "", nm)
        print("| `%s` (%s) | %s | %s | %s | %s | %s |" % (nm, unit, r.get("VGPRs"), r.get("VGPRs Spill"), r.get("SGPRs Spill"),
                                                      r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]")))
