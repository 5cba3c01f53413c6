"""Config C3 (conv PICNN, n = 2048, B = 256, nIter = 5): lockstep rounds (default for nIter <= 15) against time
slicing (a sample whose Newton solve exceeds the per-round budget is parked and resumed next round) -- the batch
holds solves of 100+ Newton updates at 19 k cycles each, which hold a lockstep launch for 0.8 ms."""
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from icnn_amd import _lib, bundle_entropy, picnn   # noqa: E402

spec = picnn.ConvSpec()
params = picnn.init_conv_params(spec, 0, "spread")
B, n_iter = 256, 5
x = np.random.RandomState(5).rand(B, spec.H, spec.W, 1).astype(np.float32)[:, :, ::-1, :].copy()
model = picnn.ConvModel(spec, params)
ctx = model.context(torch.from_numpy(x))
y0 = torch.from_numpy(np.repeat((0.2 + 0.6 * np.random.RandomState(9).rand(spec.n_labels))[None], B, axis=0)).cuda()
out = {}
for name, flags in (("lockstep", _lib.FLAG_LOCKSTEP), ("time_slice", _lib.FLAG_TIME_SLICE)):
    solver = bundle_entropy.FusedSolver(model, B, n_iter, "dual", flags=flags)
    for _ in range(2):
        solver.solve(ctx, y0)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        res = solver.solve(ctx, y0)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    out[name] = (1e3 * float(np.median(ts)), res.y.cpu().numpy().copy(), int(res.state.rounds),
                 res.newton_iters[:B].cpu().numpy().copy())
    print(name, "%.3f ms" % out[name][0], "rounds", out[name][2], "newton updates mean %.1f max %d" % (out[name][3].mean(), out[name][3].max()))
print("bit-identical y*:", bool(np.array_equal(out["lockstep"][1], out["time_slice"][1])))
