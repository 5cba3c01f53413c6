#!/usr/bin/env python3
"""Completion model (n = 2048, B = 256): lockstep launches against time-sliced rounds with the fixed finishing rounds
(no host polling since round 3), nIter = 5 and 30.  GPU box only."""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from icnn_amd import _lib, bundle_entropy, picnn  # noqa: E402

spec = picnn.ConvSpec()
B = 256
params = picnn.init_conv_params(spec, 0, "spread")
model = picnn.ConvModel(spec, params)
y0 = torch.from_numpy(np.repeat((0.2 + 0.6 * np.random.RandomState(9).rand(spec.n_labels))[None], B, axis=0)).cuda()
for seed in (5, 6):
    x = np.random.RandomState(seed).rand(B, spec.H, spec.W, 1).astype(np.float32)[:, :, ::-1, :].copy()
    ctx = model.context(torch.from_numpy(x))
    for n_iter in (5, 30):
        ref = None
        for name, flags in (("lockstep", _lib.FLAG_LOCKSTEP), ("time-sliced", _lib.FLAG_TIME_SLICE)):
            s = bundle_entropy.FusedSolver(model, B, n_iter, "dual", flags=flags)
            for _ in range(2):
                res = s.solve(ctx, y0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                res = s.solve(ctx, y0)
            torch.cuda.synchronize()
            y = res.y.cpu().numpy().copy()
            same = "" if ref is None else " identical to lockstep: %s" % np.array_equal(ref, y)
            ref = y if ref is None else ref
            print("x seed %d nIter %2d %-12s %.3f ms, rounds %d, newton max %d%s"
                  % (seed, n_iter, name, 1e3 * (time.perf_counter() - t0) / 5, res.state.rounds,
                     int(res.newton_iters[:B].max()), same))
