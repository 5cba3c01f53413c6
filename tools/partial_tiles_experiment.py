#!/usr/bin/env python3
"""Shards of the 4096-sample north-star batch (what one of N GPUs gets under strong scaling): default path of
icnn_be_solve_fc against the persistent per-tile kernel with partial tiles (ICNN_BE_FLAG_PERSISTENT)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icnn_amd import _lib, bundle_entropy, picnn  # noqa: E402

spec = picnn.bibtex_spec()
params = picnn.init_params(spec, 0, "spread")
model = picnn.FCModel(spec, params)
x = torch.from_numpy((np.random.RandomState(1000).rand(4096, spec.n_features) < 0.04).astype(np.float32)).cuda()
ctx_full = model.context(x)
for B in (512, 1024, 2048, 3072, 4096):
    ctx = ctx_full[:B].contiguous()
    out = {}
    for name, flags in (("default", 0), ("persistent", _lib.FLAG_PERSISTENT)):
        solver = bundle_entropy.FusedSolver(model, B, 10, "dual", flags=flags)
        for _ in range(3):
            solver.solve(ctx, 0.5)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            res = solver.solve(ctx, 0.5)
        e1.record()
        torch.cuda.synchronize()
        out[name] = (e0.elapsed_time(e1) / 20, res.y.cpu().numpy().copy())
    print("B=%4d  default %.3f ms   persistent(partial tiles) %.3f ms   bit-identical %s"
          % (B, out["default"][0], out["persistent"][0], np.array_equal(out["default"][1], out["persistent"][1])))
