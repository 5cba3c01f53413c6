#!/usr/bin/env python3
"""Two small measurements (GPU box only): (1) the implicit-differentiation feed behind a solve of the benchmark batch
(icnn_be_implicit_feed: what a training step adds to the solve, multi-label-cls/icnn_ebundle.py:296-314); (2) the conv solve
(150 launches at nIter = 30) eagerly and replayed from a HIP graph."""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from icnn_amd import bundle_entropy, picnn  # noqa: E402


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / reps


spec = picnn.bibtex_spec()
params = picnn.init_params(spec, 0, "spread")
model = picnn.FCModel(spec, params)
for B, n_iter in ((4096, 10), (4096, 30), (512, 30)):
    x = torch.from_numpy((np.random.RandomState(1000).rand(B, spec.n_features) < 0.04).astype(np.float32)).cuda()
    ctx = model.context(x)
    solver = bundle_entropy.FusedSolver(model, B, n_iter)
    res = solver.solve(ctx)
    true_y = torch.from_numpy((np.random.RandomState(1).rand(B, spec.n_labels) < 0.05).astype(np.float64)).cuda()
    ms_feed = timed(lambda: bundle_entropy.implicit_feed(res, true_y, "xent"))
    ms_solve = timed(lambda: solver.solve(ctx))
    rows = int(res.count[:B].sum().item())
    print("B=%4d nIter=%2d: solve %.3f ms, implicit feed %.3f ms (%d rows of %d doubles x 2)" % (B, n_iter, ms_solve, ms_feed, rows, spec.n_labels))

cs = picnn.ConvSpec()
cparams = picnn.init_conv_params(cs, 0, "spread")
cmodel = picnn.ConvModel(cs, cparams)
B = 256
x = np.random.RandomState(5).rand(B, cs.H, cs.W, 1).astype(np.float32)[:, :, ::-1, :].copy()
cctx = cmodel.context(torch.from_numpy(x))
y0 = torch.from_numpy(np.repeat((0.2 + 0.6 * np.random.RandomState(9).rand(cs.n_labels))[None], B, axis=0)).cuda()
for n_iter in (5, 30):
    solver = bundle_entropy.FusedSolver(cmodel, B, n_iter)
    eager = timed(lambda: solver.solve(cctx, y0), 5)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        solver.solve(cctx, y0)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        solver.solve(cctx, y0)
    graph = timed(g.replay, 5)
    print("conv B=256 nIter=%2d: eager %.3f ms, HIP graph replay %.3f ms" % (n_iter, eager, graph))
