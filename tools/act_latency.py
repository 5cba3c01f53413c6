#!/usr/bin/env python3
"""act() of the RL agent (one observation per environment step, RL/src/icnn.py:264-288): context + Adam as separate
launches (3 GEMM launches of be_context.hip + 1) against observation -> action in ONE launch (icnn_be_adam_fc_obs)."""
import dataclasses
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icnn_amd import picnn, rl_adam  # noqa: E402

spec = dataclasses.replace(picnn.halfcheetah_spec(), action_box=False)
params = picnn.init_params(spec, 11, "spread", yu_bias=1.0, gate_bias=1.0)
model = picnn.FCModel(spec, params)
for B in (1, 4, 256):
    obs = torch.from_numpy(np.random.RandomState(12).randn(B, spec.n_features).astype(np.float32)).cuda()
    solver = rl_adam.AdamSolver(model, B, 1000)

    def two():
        return solver.solve(model.context(obs))

    def one():
        return solver.solve_obs(obs)

    out = {}
    for name, fn in (("context + adam", two), ("one launch", one)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(50):
            t0 = time.perf_counter()
            res = fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        out[name] = (1e6 * float(np.median(ts)), int(res.iters.item()))
    print("B=%3d  context + adam: %.1f us (%d its)   one launch: %.1f us (%d its)"
          % (B, out["context + adam"][0], out["context + adam"][1], out["one launch"][0], out["one launch"][1]))
