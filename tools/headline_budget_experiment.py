#!/usr/bin/env python3
"""Headline shape (4096 x nIter 10): lockstep tiles (default) against the budgeted tile kernel + finishing launch
(ICNN_BE_FLAG_PERSISTENT | ICNN_BE_FLAG_TIME_SLICE, ICNN_BE_TILE_BUDGET per process).  GPU box only."""
import os
import subprocess
import sys

CHILD = r"""
import sys, time, numpy as np, torch
sys.path.insert(0, %r)
from icnn_amd import _lib, bundle_entropy, picnn
spec = picnn.bibtex_spec()
params = picnn.init_params(spec, 0, "spread")
B = 4096
x = torch.from_numpy((np.random.RandomState(1000).rand(B, spec.n_features) < 0.04).astype(np.float32)).cuda()
model = picnn.FCModel(spec, params)
ctx = model.context(x)
out = []
for name, flags in (("lockstep tiles", 0), ("budgeted tiles", _lib.FLAG_PERSISTENT | _lib.FLAG_TIME_SLICE)):
    s = bundle_entropy.FusedSolver(model, B, 10, flags=flags)
    for _ in range(3):
        s.solve(ctx)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        res = s.solve(ctx)
    torch.cuda.synchronize()
    out.append("%%s %%.3f ms" %% (name, 1e3 * (time.perf_counter() - t0) / 20))
print("; ".join(out))
"""
repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
r = subprocess.run([sys.executable, "-c", CHILD % repo], capture_output=True, text=True)
print(r.stdout.strip().splitlines()[-1] if r.returncode == 0 else r.stderr[-400:])
