#!/usr/bin/env python3
"""A/B of the fused VALU column/Hessian pass (be_dual_valu_dev.h) against the float64-MFMA sweep it replaces
(ICNN_BE_FLAG_MFMA_CONTRACTION) on the wide rows of the completion model, GPU box only: solve time and max|dy*|.
Round 4: the one-wave kernels (narrow rows) take the pass too, for bundles of up to 8 cuts, in EVERY kernel at once (so the
dispatch paths stay bit-identical); the Bibsonomy shapes are timed below as well.  (Round 3 had tried it at commit 028f7c8 with
four columns per lane and non-inlined instances -- 2-5 % -- and withdrawn it.)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icnn_amd import _lib, bundle_entropy, picnn  # noqa: E402


def timed(fs, ctx, y0, reps):
    for _ in range(2):
        res = fs.solve(ctx, y0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        res = fs.solve(ctx, y0)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / reps, res.y.cpu().numpy().copy()


def ab(name, model, ctx, y0, B, n_iter, reps=10):
    out = []
    for flags in (0, _lib.FLAG_MFMA_CONTRACTION):
        fs = bundle_entropy.FusedSolver(model, B, n_iter, "dual", flags=flags)
        out.append(timed(fs, ctx, y0, reps))
    print("%-28s valu %.3f ms   mfma %.3f ms   max|dy| %.2e" % (name, out[0][0], out[1][0], np.abs(out[0][1] - out[1][1]).max()),
          flush=True)


cspec = picnn.ConvSpec()
cparams = picnn.init_conv_params(cspec, 0, "spread")
cx = np.random.RandomState(5).rand(256, cspec.H, cspec.W, 1).astype(np.float32)[:, :, ::-1, :].copy()
cmodel = picnn.ConvModel(cspec, cparams)
cctx = cmodel.context(torch.from_numpy(cx))
cy0 = torch.from_numpy(np.repeat((0.2 + 0.6 * np.random.RandomState(9).rand(cspec.n_labels))[None], 256, axis=0)).cuda()
spec = picnn.bibtex_spec()
params = picnn.init_params(spec, 0, "spread")
x = torch.from_numpy((np.random.RandomState(1000).rand(4096, spec.n_features) < 0.04).astype(np.float32)).cuda()
model = picnn.FCModel(spec, params)
ctx = model.context(x)
for B, n_iter in ((4096, 10), (512, 10), (128, 10), (4096, 30)):
    ab("bibtex %d x %d" % (B, n_iter), model, ctx[:B].contiguous(), 0.5, B, n_iter, 10 if n_iter == 10 else 3)

ab("conv 256 x 5", cmodel, cctx, cy0, 256, 5, 5)
ab("conv 256 x 30", cmodel, cctx, cy0, 256, 30, 3)
