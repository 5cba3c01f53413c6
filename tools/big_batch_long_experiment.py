#!/usr/bin/env python3
"""Launch-pair path of the FC model at nIter > 15 (batches beyond two tiles per CU, or ICNN_BE_FLAG_TWO_KERNELS): time-sliced
rounds + finishing launch (the default there) against lockstep rounds, now that 100-update Newton solves are rare.  GPU box only."""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from icnn_amd import _lib, bundle_entropy, picnn  # noqa: E402

spec = picnn.bibtex_spec()
params = picnn.init_params(spec, 0, "spread")
model = picnn.FCModel(spec, params)
for B, n_iter in ((16384, 30), (16384, 20), (4096, 30)):
    x = torch.from_numpy((np.random.RandomState(1000).rand(B, spec.n_features) < 0.04).astype(np.float32)).cuda()
    ctx = model.context(x)
    for name, flags in (("default", 0), ("pairs, time-sliced", _lib.FLAG_TWO_KERNELS), ("pairs, lockstep", _lib.FLAG_TWO_KERNELS | _lib.FLAG_LOCKSTEP),
                        ("persistent tiles", _lib.FLAG_PERSISTENT)):
        s = bundle_entropy.FusedSolver(model, B, n_iter, flags=flags)
        for _ in range(2):
            res = s.solve(ctx)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            res = s.solve(ctx)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / 3
        print("B=%5d nIter=%2d %-20s %.2f ms (%.1f M inner-solves/s) rounds %d" % (B, n_iter, name, ms, B * n_iter / ms / 1e3, res.state.rounds), flush=True)
