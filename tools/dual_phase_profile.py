#!/usr/bin/env python3
"""Per-phase cycle breakdown of the dual step on the benchmark workload (GPU box only):
    python tools/dual_phase_profile.py [nIter [B [two] [pdipm]]]
Uses the library's diagnostic hook icnn_be_debug_profile (s_memtime laps, lane 0 of every wave)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from icnn_amd import _lib, bundle_entropy, picnn  # noqa: E402

_lib.use_profiling_build()            # the laps are compiled into the profiling variant of the library only

PH = ["cut+h", "stage rows", "rank test", "row sums c", "column phase (a,z,w)", "mfma H", "line search+cycle test",
      "y update+prune", "grad/argmax/free set", "reduced Newton solve", "mfma: operand setup", "mfma: column sweep"]
# variant pdipm (round 4): the laps inside ipm_solve (be_ipm_dev.h) reuse the same twelve counters
PH_IPM = ["cut+h", "stage rows", "rank test", "(unused)", "ipm: residual column pass", "ipm: mfma sweep", "ipm: affine dy pass + steps",
          "y update+prune", "ipm: G y, norms, stop test", "ipm: solve 1 (affine)", "ipm: solve 2 (corrector)", "ipm: corrector pass + update"]
PH = PH + ["control words (global round trip)", "new cut + older rows: loads, staging", "queue for the LDS staging region (grouped tile phase)", "end-of-phase barrier (grouped tile phase)"]
NPH = _lib.load().icnn_be_debug_profile_phases()     # DUAL_PROF_PHASES in be_kernels.h: the buffer's row length comes from the library
assert NPH == len(PH), (NPH, len(PH))
n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
variant = "pdipm" if "pdipm" in sys.argv[3:] else "dual"
if variant == "pdipm":
    PH = PH_IPM + PH[12:]
spec = picnn.bibtex_spec()
params = picnn.init_params(spec, 0, "spread")
x = torch.from_numpy((np.random.RandomState(1000).rand(B, spec.n_features) < 0.04).astype(np.float32)).cuda()
model = picnn.FCModel(spec, params)
ctx = model.context(x)
flags = _lib.FLAG_TWO_KERNELS if (len(sys.argv) > 3 and sys.argv[3] == 'two') else 0
solver = bundle_entropy.FusedSolver(model, B, n_iter, variant, flags=flags)
print('variant %s, B = %d, nIter = %d; path: %s' % (variant, B, n_iter, 'two kernels per round' if flags else 'persistent kernel (where eligible)'))
solver.solve(ctx)
torch.cuda.synchronize()
prof = torch.zeros(max(B, 4096) + 8, NPH, dtype=torch.int64, device="cuda")
lib = _lib.load()
lib.icnn_be_debug_profile(C.c_void_p(prof.data_ptr()))
res = solver.solve(ctx)
torch.cuda.synchronize()
lib.icnn_be_debug_profile(None)
p = prof.cpu().numpy().astype(np.float64)[:B]
tot = p.sum(1)
print("cycles per sample over %d outer iterations (s_memtime ticks): mean %.0f  median %.0f  max %.0f"
      % (n_iter, tot.mean(), np.median(tot), tot.max()))
for i, name in enumerate(PH):
    print("  %-24s mean %9.0f (%5.1f%%)   max %9.0f" % (name, p[:, i].mean(), 100 * p[:, i].sum() / tot.sum(), p[:, i].max()))
print("newton updates per sample: mean %.1f max %d" % (res.newton_iters[:B].float().mean().item(), res.newton_iters[:B].max().item()))
