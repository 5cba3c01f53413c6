#!/usr/bin/env python3
"""Per-phase cycle breakdown of dual_step_kernel<float, 16, 8> (n = 2048, eight waves per sample) on the
completion workload (GPU box only)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from icnn_amd import _lib, bundle_entropy, picnn  # noqa: E402

PH = ["cut+h+stage", "zero/ones rows", "rank test", "row sums c", "column phase (a,z,w)", "mfma H (combine, barrier)",
      "line search+cycle test", "y update+prune", "grad/argmax/free set", "reduced Newton solve", "mfma: operand setup",
      "mfma: column sweep"]
B, n_iter = 256, (int(sys.argv[1]) if len(sys.argv) > 1 else 5)
FLAGS = int(sys.argv[2]) if len(sys.argv) > 2 else 0      # e.g. 64 = ICNN_BE_FLAG_GLOBAL_BUNDLE
spec = picnn.ConvSpec()
params = picnn.init_conv_params(spec, 0, "spread")
x = np.random.RandomState(5).rand(B, spec.H, spec.W, 1).astype(np.float32)
model = picnn.ConvModel(spec, params)
ctx = model.context(torch.from_numpy(x))
solver = bundle_entropy.FusedSolver(model, B, n_iter, "dual", flags=FLAGS)
y0 = torch.full((B, spec.n_labels), 0.5, dtype=torch.float64, device="cuda")
solver.solve(ctx, 0.5)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    solver.solve(ctx, 0.5)
e1.record()
torch.cuda.synchronize()
print("fused solve: %.3f ms" % (e0.elapsed_time(e1) / 5))
prof = torch.zeros(B, len(PH), dtype=torch.int64, device="cuda")
lib = _lib.load()
lib.icnn_be_debug_profile(C.c_void_p(prof.data_ptr()))
res = solver.solve(ctx, 0.5)
torch.cuda.synchronize()
lib.icnn_be_debug_profile(None)
p = prof.cpu().numpy().astype(np.float64)
tot = p.sum(1)
print("cycles per sample (thread 0) over %d outer iterations: mean %.0f max %.0f" % (n_iter, tot.mean(), tot.max()))
for i, name in enumerate(PH):
    print("  %-30s mean %9.0f (%5.1f%%)   max %9.0f" % (name, p[:, i].mean(), 100 * p[:, i].sum() / tot.sum(), p[:, i].max()))
print("newton updates per sample: mean %.1f max %d; cuts mean %.1f" % (res.newton_iters[:B].float().mean().item(), res.newton_iters[:B].max().item(), res.count[:B].float().mean().item()))
