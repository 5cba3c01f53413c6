#!/usr/bin/env python3
"""Per-phase cycle breakdown of the n = 2048 dual step (8 waves per sample) on the completion workload of
tools/prof_target.py c3 (GPU box only).  Same hook as tools/dual_phase_profile.py: s_memtime laps of thread 0."""
import ctypes as C
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from icnn_amd import _lib, bundle_entropy, picnn  # noqa: E402

_lib.use_profiling_build()            # the laps are compiled into the profiling variant of the library only

PH = ["cut+h", "stage rows", "rank test", "row sums c", "column phase (a,z,w)", "mfma H + combine", "line search+cycle test",
      "y update+prune", "grad/argmax/free set", "reduced Newton solve", "mfma: operand setup", "mfma: column sweep"]
PH = PH + ["control words (global round trip)", "new cut + older rows: loads, staging", "(spare)", "(spare)"]
NPH = _lib.load().icnn_be_debug_profile_phases()     # the buffer row length is the library's (DUAL_PROF_PHASES)
n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 5
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
spec = picnn.ConvSpec()
params = picnn.init_conv_params(spec, 0, "spread")
x = np.random.RandomState(5).rand(B, spec.H, spec.W, 1).astype(np.float32)[:, :, ::-1, :].copy()
model = picnn.ConvModel(spec, params)
ctx = model.context(torch.from_numpy(x))
y0 = torch.from_numpy(np.repeat((0.2 + 0.6 * np.random.RandomState(9).rand(spec.n_labels))[None], B, axis=0)).cuda()
fs = bundle_entropy.FusedSolver(model, B, n_iter, "dual")
fs.solve(ctx, y0)
torch.cuda.synchronize()
prof = torch.zeros(max(B, 4096) + 8, NPH, dtype=torch.int64, device="cuda")
lib = _lib.load()
lib.icnn_be_debug_profile(C.c_void_p(prof.data_ptr()))
res = fs.solve(ctx, y0)
torch.cuda.synchronize()
lib.icnn_be_debug_profile(None)
p = prof.cpu().numpy().astype(np.float64)[:B]
upd = res.newton_iters[:B].cpu().numpy().astype(np.float64)
tot = p.sum(1)
print("ticks per sample over %d outer iterations: mean %.0f  median %.0f  max %.0f" % (n_iter, tot.mean(), np.median(tot), tot.max()))
for i, name in enumerate(PH):
    print("  %-26s mean %9.0f (%5.1f%%)   max %9.0f   per update %7.0f" % (name, p[:, i].mean(), 100 * p[:, i].sum() / tot.sum(),
                                                                           p[:, i].max(), p[:, i].sum() / max(upd.sum(), 1)))
print("newton updates per sample: mean %.1f max %d; sorted top 8: %s" % (upd.mean(), upd.max(), np.sort(upd)[-8:]))
