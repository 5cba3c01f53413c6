#!/usr/bin/env python3
"""Time one energy+gradient evaluation of the conv PICNN (BASELINE.json configs[2] shape, B = 256) and a whole fused
solve (nIter = 5) on the GPU box.   python tools/conv_time.py [B]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icnn_amd import bundle_entropy, picnn  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
spec = picnn.ConvSpec()
params = picnn.init_conv_params(spec, 1, "spread")
x = np.random.RandomState(51).rand(B, spec.H, spec.W, 1).astype(np.float32)
model = picnn.ConvModel(spec, params)
ctx = model.context(torch.from_numpy(x))
y = torch.from_numpy(0.2 + 0.6 * np.random.RandomState(9).rand(B, spec.n_labels)).cuda()
for _ in range(3):
    model.fg(ctx, y)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    model.fg(ctx, y)
e1.record()
torch.cuda.synchronize()
print("conv fg (4 launches) at B=%d: %.1f us per evaluation" % (B, 1e3 * e0.elapsed_time(e1) / 50))
solver = bundle_entropy.FusedSolver(model, B, 5, "dual")
for _ in range(2):
    solver.solve(ctx, y)
torch.cuda.synchronize()
e0.record()
for _ in range(10):
    solver.solve(ctx, y)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print("fused conv solve B=%d nIter=5: %.3f ms, %.0f inner-solves/s" % (B, ms, B * 5 / ms * 1e3))
# x-only context: the device producer (icnn_be_conv_context) against the same statement in torch ops on the GPU
xd = torch.from_numpy(x).cuda()
for name, fn in (("icnn_be_conv_context (7 GEMM + 4 BN launches)", lambda: model.context(xd)),
                 ("torch conv2d / addmm / BatchNorm ops", lambda: picnn.conv_context(spec, params, xd))):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print("x-only context at B=%d, %s: %.3f ms" % (B, name, e0.elapsed_time(e1) / 20))
