#!/usr/bin/env python3
"""BASELINE configs[3] at full size (GPU box only): which samples of the 4096 x nIter-30 solve differ from the order-matched
oracle, and is it the device formulation (cycle shortcut / extrapolation) or plain float64 rounding that moved them?
Runs the default dispatch and ICNN_BE_FLAG_NO_CYCLE_SHORTCUT (the reference's full Newton cap), the oracle on the slice
tests/test_gpu_parity.py uses, and saves the context rows of the samples that differ to gpurun_out/c4_outliers.npz."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from icnn_amd import _lib, bundle_entropy, picnn  # noqa: E402
from oracle import bundle_entropy_oracle as oracle  # noqa: E402
from oracle import picnn_oracle  # noqa: E402
from gpu_util import compare_with_oracle, result_to_host  # noqa: E402
import test_gpu_parity as T  # noqa: E402

spec = picnn.bibtex_spec()
B, n_iter, S = 4096, 30, int(sys.argv[1]) if len(sys.argv) > 1 else 256
params, x = T._picnn_problem(spec, B, 0, "spread")
model = picnn.FCModel(spec, params)
ctx = model.context(torch.from_numpy(x))
hosts = {}
for name, flags in (("default", 0), ("full_cap", _lib.FLAG_NO_CYCLE_SHORTCUT), ("two_kernels", _lib.FLAG_TWO_KERNELS)):
    res = bundle_entropy.FusedSolver(model, B, n_iter, "dual", flags=flags).solve(ctx, 0.5)
    hosts[name] = result_to_host(res)
idx = T._oracle_slice(hosts["default"], B, S)
ctx_rows = ctx[torch.from_numpy(idx).cuda()].cpu().numpy()
fg = picnn_oracle.make_fg_chain(params, ctx_rows, list(spec.szs))
with np.errstate(all="ignore"):
    ora = oracle.solve_batch(fg, np.full((len(idx), spec.n_labels), 0.5), n_iter)
bad = set()
for name, host in hosts.items():
    dy, discrete = compare_with_oracle(T._slice_host(host, idx), ora)
    worst = np.argsort(-dy)[:6]
    print("%-11s vs oracle: max|dy| %.3e, %d discrete, samples > 1e-7: %s" % (name, dy.max(), len(discrete),
          [(int(idx[i]), "%.1e" % dy[i], int(host["newton"][idx[i]])) for i in worst if dy[i] > 1e-7]))
    bad |= set(int(i) for i in np.nonzero(dy > 1e-7)[0]) | set(discrete)
d = np.max(np.abs(hosts["default"]["y"] - hosts["full_cap"]["y"]), axis=1)
print("default vs full_cap over the whole batch: max %.3e, samples > 1e-7: %d, > 1e-9: %d" % (d.max(), int((d > 1e-7).sum()), int((d > 1e-9).sum())))
d2 = np.max(np.abs(hosts["default"]["y"] - hosts["two_kernels"]["y"]), axis=1)
print("default vs two_kernels: max %.3e" % d2.max())
bad = sorted(bad)
np.savez(os.path.join(REPO, "gpurun_out", "c4_outliers.npz"), idx=idx[bad], ctx=ctx_rows[bad],
         y_default=hosts["default"]["y"][idx[bad]], y_full=hosts["full_cap"]["y"][idx[bad]], y_oracle=ora.y[bad],
         newton=hosts["default"]["newton"][idx[bad]], n_iters_gpu=np.array(hosts["default"]["n_iters"])[idx[bad]],
         n_iters_ora=np.array(ora.n_iters)[bad])
print("saved %d outliers" % len(bad))
