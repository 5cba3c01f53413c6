#!/usr/bin/env python3
"""Per-phase cycle breakdown of the PICNN evaluation inside fused_rows_solve_kernel (1-4 samples per workgroup, VALU fma
chains; GPU box only): diagnostic hook icnn_be_debug_profile_fc, cycle-counter laps by lane 0 of every wave, summed over
the rounds of one solve and divided by nIter.  usage: rows_phase_profile.py [batch [nIter]]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from icnn_amd import _lib, bundle_entropy, picnn  # noqa: E402

_lib.use_profiling_build()            # the laps are compiled into the profiling variant of the library only

PH = {13: "y load, operands y * yu_i", 0: "L0 chains y->600 + epilogue", 1: "L0 barrier wait", 2: "L1 chains (y, z0)->159",
      3: "L1 barrier wait", 8: "bwd1: dE/dy, d0 = d1 Wzu1^T, E", 10: "bwd1 barrier wait", 11: "bwd0: dE/dy += d0 Wyu0^T",
      12: "bwd0 barrier wait", 14: "f, g store"}
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n_iter = int(sys.argv[2]) if len(sys.argv) > 2 else 10
spec = picnn.bibtex_spec()
params = picnn.init_params(spec, 0, "spread")
x = torch.from_numpy((np.random.RandomState(1000).rand(B, spec.n_features) < 0.04).astype(np.float32)).cuda()
model = picnn.FCModel(spec, params)
ctx = model.context(x)
y0 = torch.full((B, spec.n_labels), 0.5, dtype=torch.float64, device="cuda")
fs = bundle_entropy.FusedSolver(model, B, n_iter, "dual")
fs.solve(ctx, y0)
torch.cuda.synchronize()
prof = torch.zeros(4096, 8, 16, dtype=torch.int64, device="cuda")
lib = _lib.load()
lib.icnn_be_debug_profile_fc(C.c_void_p(prof.data_ptr()))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
fs.solve(ctx, y0)
e1.record()
torch.cuda.synchronize()
lib.icnn_be_debug_profile_fc(None)
p = prof.cpu().numpy().astype(np.float64) / n_iter
p = p[p.sum((1, 2)) > 0]
tot = p.sum(2)
print("B = %d, nIter = %d: %d workgroups, solve %.3f ms; cycles per wave and evaluation: mean %.0f  max %.0f  (%.1f us at 2.4 GHz)"
      % (B, n_iter, p.shape[0], e0.elapsed_time(e1), tot.mean(), tot.max(), tot.mean() / 2400))
for i, name in PH.items():
    print("  %-32s mean %8.0f (%5.1f%%)   wave-min %8.0f  wave-max %8.0f" %
          (name, p[:, :, i].mean(), 100 * p[:, :, i].sum() / tot.sum(), p[:, :, i].mean(0).min(), p[:, :, i].mean(0).max()))
