#!/usr/bin/env python3
"""Single-wave latency of one dual step as a function of the bundle size k (GPU box only; profiling variant of the library):
    python tools/update_latency.py [B]
A piecewise-linear energy of sixty pieces keeps every cut active (k = t + 1 cuts in round t), generic mode (icnn_be_dual_step:
the stand-alone dual_step_kernel, one wave per sample; B = 256 = one sample per CU: no contention).  Two solves, nIter = K and
K - 1: the difference of the phase counters is round K - 1 alone, i.e. the dual step at exactly k = K cuts."""
import ctypes as C
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import problems  # noqa: E402
from icnn_amd import _lib, bundle_entropy  # noqa: E402

_lib.use_profiling_build()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lib = _lib.load()
NPH = lib.icnn_be_debug_profile_phases()
PH = ["cut+h", "stage", "rank test", "row sums", "column phase", "mfma H", "line search", "y update", "grad/free set", "Newton solve",
      "-", "-", "control", "new cut+rows"]
prob = problems.max_affine(77, B, 159, 60, 1.0)


def run(n_iter):
    prof = torch.zeros(max(B, 4096) + 8, NPH, dtype=torch.int64, device="cuda")
    lib.icnn_be_debug_profile(C.c_void_p(prof.data_ptr()))
    res = bundle_entropy.solveBatch(prob.fg, prob.y0(), nIter=n_iter, native=True, check=False)
    torch.cuda.synchronize()
    lib.icnn_be_debug_profile(None)
    return prof.cpu().numpy().astype(np.float64)[:B, :14], res.newton_iters[:B].cpu().numpy().astype(np.float64), res.count[:B].cpu().numpy()


run(4)
print("k    updates  per-update: " + " ".join("%12s" % p for p in ("column", "mfma H", "solve", "line search", "grad")) +
      "   | per round: " + " ".join("%9s" % p for p in ("cut+h", "rank", "rowsums", "y upd", "rows")) + "   total")
for K in (4, 8, 9, 12, 15, 16, 17, 20, 21, 24, 28, 31):
    p1, u1, c1 = run(K)
    p0, u0, c0 = run(K - 1)
    ok = (c1 == K) & (c0 == K - 1) & (u1 > u0)
    d, du = (p1 - p0)[ok], (u1 - u0)[ok]
    if not ok.any():
        print("%2d   (no sample kept all cuts)" % K)
        continue
    per_upd = d[:, [4, 5, 9, 6, 8]].sum(0) / du.sum()
    per_round = d[:, [0, 2, 3, 7, 13]].mean(0)
    print("%2d   %6.1f              " % (K, du.mean()) + " ".join("%12.0f" % v for v in per_upd) + "                " +
          " ".join("%9.0f" % v for v in per_round) + "   %7.0f" % d.sum(1).mean())
