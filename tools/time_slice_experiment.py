"""Lockstep rounds against time slicing (ICNN_BE_FLAG_TIME_SLICE: Newton solves that exceed the per-round budget are
parked and resumed next round) on the Bibsonomy shape (GPU box only): the data behind the automatic policy
(time slicing for nIter > 15)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icnn_amd import bundle_entropy, picnn, _lib
spec = picnn.bibtex_spec(); params = picnn.init_params(spec, 0, "spread")
for B, n_iter in ((4096, 30), (4096, 10), (16384, 10)):
    x = torch.from_numpy((np.random.RandomState(7).rand(B, spec.n_features) < 0.04).astype(np.float32)).cuda()
    model = picnn.FCModel(spec, params); ctx = model.context(x)
    for flags in (_lib.FLAG_LOCKSTEP, _lib.FLAG_TIME_SLICE):
        solver = bundle_entropy.FusedSolver(model, B, n_iter, "dual", flags=flags)
        for _ in range(2): solver.solve(ctx, 0.5)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): res = solver.solve(ctx, 0.5)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        print("B=%d nIter=%d flags=%d: %.2f ms, %.2f M inner-solves/s, rounds %d" % (B, n_iter, flags, dt * 1e3, B * n_iter / dt / 1e6, res.state.rounds))
