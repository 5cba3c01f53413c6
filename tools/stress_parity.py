#!/usr/bin/env python3
"""Wider parity sweep on the GPU box: fused HIP solve vs the oracle solver fed by the MFMA-order
PICNN, several seeds / regimes / nIter, both scheduling modes.  Prints one line per case and a
final verdict; exits non-zero on any sample with a different discrete outcome, or |dy| > 1e-7 at nIter <= 15.  At
nIter = 30 the bundles hold 16-23 nearly parallel cuts and the reference algorithm amplifies float64 rounding by about
an order of magnitude per outer iteration: there the bound is 1e-4 and the count above 1e-7 is
printed (which samples those are moves with the compiler's fused-multiply-add contraction choices in the dual step)."""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from icnn_amd import _lib, bundle_entropy, picnn  # noqa: E402
from oracle import bundle_entropy_oracle as oracle  # noqa: E402
from oracle import picnn_oracle  # noqa: E402

bad = 0
cases = [(0, "spread", 1024, 10, 0), (1, "spread", 1024, 10, 0), (2, "spread", 512, 30, 0), (3, "spread", 512, 30, _lib.FLAG_LOCKSTEP),
         (4, "init", 1024, 10, 0), (5, "spread", 1024, 10, _lib.FLAG_TIME_SLICE), (6, "spread", 777, 7, 0), (7, "spread", 300, 15, 0)]
for seed, regime, B, n_iter, flags in cases:
    spec = picnn.bibtex_spec()
    params = picnn.init_params(spec, seed, regime)
    x = (np.random.RandomState(100 + seed).rand(B, spec.n_features) < 0.04).astype(np.float32)
    model = picnn.FCModel(spec, params)
    ctx = model.context(torch.from_numpy(x))
    y0 = np.full((B, spec.n_labels), 0.5)
    res = bundle_entropy.solveBatch(f=model, ctx=ctx, y0=y0, nIter=n_iter, native=True, flags=flags)
    fg = picnn_oracle.make_fg_chain(params, ctx.cpu().numpy(), list(spec.szs))
    t0 = time.time()
    with np.errstate(all="ignore"):
        ora = oracle.solve_batch(fg, np.full((B, spec.n_labels), 0.5), n_iter)
    act = res.active.cpu().numpy(); cnt = res.count[:B].cpu().numpy(); its = res.n_iters[:B].cpu().numpy()
    dy = np.max(np.abs(res.y.cpu().numpy() - ora.y), axis=1)
    disc = sum(1 for u in range(B) if list(act[u, :cnt[u]]) != list(ora.active[u]) or int(its[u]) != int(ora.n_iters[u]))
    print("seed %d %-6s B=%4d nIter=%2d flags=%d: max|dy| %.2e, discrete diffs %d, cuts mean %.1f max %d, "
          "finished early %.0f%%, newton max %d  (oracle %.0fs)"
          % (seed, regime, B, n_iter, flags, dy.max(), disc, cnt.mean(), cnt.max(), 100 * np.mean(its < n_iter),
             int(res.newton_iters[:B].max().item()), time.time() - t0), flush=True)
    bad += disc + int((dy > (1e-7 if n_iter <= 15 else 1e-4)).sum())
    if n_iter > 15 and (dy > 1e-7).any():
        print("        (%d of %d samples above 1e-7 at nIter=%d)" % (int((dy > 1e-7).sum()), B, n_iter))
print("STRESS", "OK" if bad == 0 else "FAILED (%d)" % bad)
sys.exit(1 if bad else 0)
