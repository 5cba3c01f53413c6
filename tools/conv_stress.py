#!/usr/bin/env python3
"""Stress of the wide-row dual step (completion model, n = 2048) over seeds and iteration counts, GPU box only:
default dispatch (LDS kernel, then dual_step_wide_kernel with split staging) against forced device-memory staging
(ICNN_BE_FLAG_GLOBAL_BUNDLE: bit-identical), the MFMA sweep (ICNN_BE_FLAG_MFMA_CONTRACTION: same discrete outcomes, y* to
rounding) and the time-sliced rounds (bit-identical)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icnn_amd import _lib, bundle_entropy, picnn  # noqa: E402

spec = picnn.ConvSpec()
bad = 0
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    for n_iter in (5, 13, 20, 31):
        B = 12 + 3 * seed
        params = picnn.init_conv_params(spec, seed, "spread")
        x = np.random.RandomState(seed + 50).rand(B, spec.H, spec.W, 1).astype(np.float32)
        model = picnn.ConvModel(spec, params)
        ctx = model.context(torch.from_numpy(x))
        y0 = torch.from_numpy(np.repeat((0.2 + 0.6 * np.random.RandomState(9 + seed).rand(spec.n_labels))[None], B, axis=0)).cuda()
        outs = {}
        for name, flags in (("default", 0), ("glb", _lib.FLAG_GLOBAL_BUNDLE), ("mfma", _lib.FLAG_MFMA_CONTRACTION),
                            ("sliced", _lib.FLAG_TIME_SLICE)):
            res = bundle_entropy.FusedSolver(model, B, n_iter, "dual", flags=flags).solve(ctx, y0)
            torch.cuda.synchronize()
            outs[name] = [t.cpu().numpy().copy() for t in (res.y, res.count[:B], res.n_iters[:B], res.status[:B], res.newton_iters[:B])]
        d = outs["default"]
        same_glb = all(np.array_equal(a, b) for a, b in zip(d, outs["glb"]))
        same_sl = all(np.array_equal(a, b) for a, b in zip(d, outs["sliced"]))
        disc = np.array_equal(d[1], outs["mfma"][1]) and np.array_equal(d[2], outs["mfma"][2])
        dy = np.abs(d[0] - outs["mfma"][0]).max()
        ok = same_glb and same_sl and disc and dy <= 1e-7 and not d[3].any()
        bad += not ok
        print("seed %d nIter %2d B %2d: cuts max %2d, newton max %3d | glb identical %s, sliced identical %s, mfma discrete %s max|dy| %.1e %s"
              % (seed, n_iter, B, d[1].max(), d[4].max(), same_glb, same_sl, disc, dy, "" if ok else "<-- CHECK"), flush=True)
print("CONV STRESS OK" if not bad else "CONV STRESS: %d to check" % bad)
