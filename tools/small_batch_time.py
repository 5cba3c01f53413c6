#!/usr/bin/env python3
"""Fused solve times of the small-batch (per-sample kernel) shapes: python tools/small_batch_time.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icnn_amd import bundle_entropy, picnn
spec = picnn.bibtex_spec(); params = picnn.init_params(spec, 0, "spread")
for B in (128, 256, 512, 1024):
    x = (np.random.RandomState(100).rand(B, spec.n_features) < 0.04).astype(np.float32)
    model = picnn.FCModel(spec, params); ctx = model.context(torch.from_numpy(x))
    solver = bundle_entropy.FusedSolver(model, B, 10, "dual")
    for _ in range(3): solver.solve(ctx, 0.5)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): solver.solve(ctx, 0.5)
    torch.cuda.synchronize()
    print("B=%4d nIter=10: %.3f ms" % (B, (time.perf_counter() - t0) / 20 * 1e3))
y = torch.rand(256, spec.n_labels, dtype=torch.float64, device="cuda")
x = (np.random.RandomState(100).rand(256, spec.n_features) < 0.04).astype(np.float32)
ctx = model.context(torch.from_numpy(x))
for _ in range(5): model.fg(ctx, y)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): model.fg(ctx, y)
torch.cuda.synchronize()
print("fc_fg rows kernel B=256: %.1f us" % ((time.perf_counter() - t0) / 50 * 1e6))
