#!/usr/bin/env python3
"""BASELINE configs[3] on one GPU (4096 samples, nIter 30): solve time against the per-round Newton budget of the persistent
tile kernel (ICNN_BE_TILE_BUDGET, read once per process -> one child process per value).  GPU box only."""
import os
import subprocess
import sys

CHILD = r"""
import sys, time, numpy as np, torch
sys.path.insert(0, %r)
from icnn_amd import bundle_entropy, picnn
spec = picnn.bibtex_spec()
params = picnn.init_params(spec, 0, "spread")
out = []
for seed, B in ((1000, 4096), (7, 4096), (1000, 2048)):
    x = torch.from_numpy((np.random.RandomState(seed).rand(B, spec.n_features) < 0.04).astype(np.float32)).cuda()
    model = picnn.FCModel(spec, params)
    ctx = model.context(x)
    solver = bundle_entropy.FusedSolver(model, B, 30)
    for _ in range(2):
        solver.solve(ctx)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        res = solver.solve(ctx)
    torch.cuda.synchronize()
    out.append("%%d/%%d: %%.2f ms (newton max %%d)" %% (B, seed, 1e3 * (time.perf_counter() - t0) / 5, int(res.newton_iters[:B].max())))
print("  ".join(out))
"""
repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for budget in sys.argv[1:] or ["4", "6", "8", "12", "16", "0"]:
    env = dict(os.environ, ICNN_BE_TILE_BUDGET=budget)
    r = subprocess.run([sys.executable, "-c", CHILD % repo], env=env, capture_output=True, text=True)
    print("budget %2s: %s" % (budget, r.stdout.strip().splitlines()[-1] if r.returncode == 0 else r.stderr[-300:]))
