#!/usr/bin/env python3
"""Per-phase cycle breakdown of fc_fg_kernel on the benchmark workload (GPU box only): diagnostic hook
icnn_be_debug_profile_fc, s_memtime laps by lane 0 of every wave of every workgroup."""
import ctypes as C
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from icnn_amd import _lib, picnn  # noqa: E402

_lib.use_profiling_build()            # the laps are compiled into the profiling variant of the library only

PH = ["y load", "L0 prep (y*yu)", "L0 GEMM y->600 + epilogue", "L0 barrier wait", "L1 prep", "L1 GEMMs (z0->159, y->159)",
      "L1 barrier wait", "scalar layer, E, delta init", "bwd1 dE/dy += d1 Wyu1^T", "bwd1 d0 = d1 Wzu1^T", "bwd1 barrier wait",
      "bwd0 dE/dy += d0 Wyu0^T", "bwd0 barrier wait", "g store", "adam: entropy, best, stop rule", "adam: moments, step"]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
if len(sys.argv) > 2 and sys.argv[2] == "rl":          # the RL agent's 200-200 negQ network (same phase list)
    spec = picnn.halfcheetah_spec()
    params = picnn.init_params(spec, 0, "spread", yu_bias=1.0, gate_bias=1.0)
    x = torch.from_numpy(np.random.RandomState(1000).randn(max(B, 64), spec.n_features).astype(np.float32)).cuda()
else:
    spec = picnn.bibtex_spec()
    params = picnn.init_params(spec, 0, "spread")
    x = torch.from_numpy((np.random.RandomState(1000).rand(B, spec.n_features) < 0.04).astype(np.float32)).cuda()
model = picnn.FCModel(spec, params)
ctx = model.context(x)[:B].contiguous()
y = torch.full((B, spec.n_labels), 0.5, dtype=torch.float64, device="cuda")
ADAM = len(sys.argv) > 3 and sys.argv[3] == "adam"      # profile the persistent Adam loop instead (per iteration)
if ADAM:
    import dataclasses

    from icnn_amd import rl_adam
    model = picnn.FCModel(dataclasses.replace(spec, action_box=False), params)
    solver = rl_adam.AdamSolver(model, B, int(sys.argv[4]) if len(sys.argv) > 4 else 1000)
    solver.solve(ctx)
for _ in range(3):
    model.fg(ctx, y)
torch.cuda.synchronize()
nwg = (B + 15) // 16
prof = torch.zeros(nwg, 16, 16, dtype=torch.int64, device="cuda")
lib = _lib.load()
lib.icnn_be_debug_profile_fc(C.c_void_p(prof.data_ptr()))
if ADAM:
    evals = int(solver.solve(ctx).iters.item()) + 1
else:
    model.fg(ctx, y)
    evals = 1
torch.cuda.synchronize()
lib.icnn_be_debug_profile_fc(None)
p = prof.cpu().numpy().astype(np.float64) / evals
if ADAM:
    print("Adam loop: %d PICNN evaluations; cycles below are per evaluation" % evals)
tot = p.sum(2)
print("cycles per wave for one fc_fg launch: mean %.0f  max %.0f  (%.1f us at 2.4 GHz)" % (tot.mean(), tot.max(), tot.mean() / 2400))
for i, name in enumerate(PH[:16 if ADAM else 14]):
    print("  %-30s mean %8.0f (%5.1f%%)   wave-min %8.0f  wave-max %8.0f" %
          (name, p[:, :, i].mean(), 100 * p[:, :, i].sum() / tot.sum(), p[:, :, i].mean(0).min(), p[:, :, i].mean(0).max()))
