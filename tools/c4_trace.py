#!/usr/bin/env python3
"""Per-round timeline of the persistent tile kernel's grouped dual phase (BASELINE configs[3]: 4096 x 30; GPU box only):
    python tools/c4_trace.py [B [nIter [tiles-to-print]]]
Profiling variant of the library + icnn_be_debug_trace: for every sample and round the shader-clock stamps of the tile's dual
phase starting, the sample's dual step starting (behind its wait for a staging region) and ending, and its Newton updates."""
import ctypes as C
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from icnn_amd import _lib, bundle_entropy, picnn  # noqa: E402

_lib.use_profiling_build()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n_iter = int(sys.argv[2]) if len(sys.argv) > 2 else 30
n_print = int(sys.argv[3]) if len(sys.argv) > 3 else 2
spec = picnn.bibtex_spec()
params = picnn.init_params(spec, 0, "spread")
x = torch.from_numpy((np.random.RandomState(1000).rand(B, spec.n_features) < 0.04).astype(np.float32)).cuda()
model = picnn.FCModel(spec, params)
ctx = model.context(x)
lib = _lib.load()
lib.icnn_be_debug_trace.argtypes = [C.c_void_p]
solver = bundle_entropy.FusedSolver(model, B, n_iter, "dual")
solver.solve(ctx)
torch.cuda.synchronize()
prof = torch.zeros(max(B, 4096) + 8, lib.icnn_be_debug_profile_phases(), dtype=torch.int64, device="cuda")
trace = torch.zeros(B, _lib.MAX_ITERS, 4, dtype=torch.int64, device="cuda")
lib.icnn_be_debug_profile(C.c_void_p(prof.data_ptr()))
lib.icnn_be_debug_trace(C.c_void_p(trace.data_ptr()))
res = solver.solve(ctx)
torch.cuda.synchronize()
lib.icnn_be_debug_profile(None)
lib.icnn_be_debug_trace(None)
t = trace.cpu().numpy().astype(np.float64)[:, :n_iter]                   # [B, rounds, 4]
upd_total = res.newton_iters[:B].cpu().numpy()
live = t[:, :, 2] > 0
tiles = B // 16
T = t.reshape(tiles, 16, n_iter, 4)
L = live.reshape(tiles, 16, n_iter)
start = np.where(L, T[..., 0], np.inf).min(1)                            # [tiles, rounds]
end = np.where(L, T[..., 2], -np.inf).max(1)
dur = np.where(np.isfinite(start) & np.isfinite(end), end - start, 0.0)
run = np.where(L, T[..., 2] - T[..., 1], 0.0)
wait = np.where(L, T[..., 1] - T[..., 0], 0.0)
upd_next = np.concatenate([T[..., 3][:, :, 1:], upd_total.reshape(tiles, 16, 1)], axis=2)
upd = np.where(L, upd_next - T[..., 3], 0.0)                             # updates of the round (valid while the next round ran)
tot = dur.sum(1)
print("dual phase per tile, summed over %d rounds (k cycles): mean %.0f p50 %.0f p90 %.0f max %.0f" % (
    n_iter, tot.mean() / 1e3, np.median(tot) / 1e3, np.percentile(tot, 90) / 1e3, tot.max() / 1e3))
print("per round, mean over tiles (k cycles): " + " ".join("%.0f" % (v / 1e3) for v in dur.mean(0)))
print("per round, max over tiles  (k cycles): " + " ".join("%.0f" % (v / 1e3) for v in dur.max(0)))
# how much of a tile's dual phase is the critical sample's own run, how much its wait
crit = np.where(L, T[..., 2], -np.inf).argmax(1)                         # [tiles, rounds] the sample that ends last
ci = np.arange(tiles)[:, None], crit, np.arange(n_iter)[None, :]
crit_run, crit_wait, crit_upd = run[ci], wait[ci], upd[ci]
has = dur > 0
print("critical sample of a round: run %.0f k + wait %.0f k of the phase's %.0f k (means over tiles and rounds); it has the round's most "
      "updates in %.0f %% of the rounds; its updates mean %.1f, the round's mean %.1f"
      % (crit_run[has].mean() / 1e3, crit_wait[has].mean() / 1e3, dur[has].mean() / 1e3,
         100 * (crit_upd >= upd.max(1))[has].mean(), crit_upd[has].mean(), (upd.sum(1) / np.maximum(L.sum(1), 1))[has].mean()))
# was the critical sample predictable from its previous round?
prev_upd = np.concatenate([np.zeros((tiles, 16, 1)), upd[:, :, :-1]], axis=2)
rank_prev = (-prev_upd).argsort(1).argsort(1)                            # rank of every sample by last round's updates
print("rank of the critical sample by its PREVIOUS round's updates (0 = most): mean %.1f; in the top 4 in %.0f %% of the rounds"
      % (rank_prev[ci][has].mean(), 100 * (rank_prev[ci] < 4)[has].mean()))
# cost per update as a function of how many samples run at the same time is not separable here; print the worst tiles instead
for tix in np.argsort(-tot)[:n_print]:
    print("tile %d: dual phase total %.0f k" % (tix, tot[tix] / 1e3))
    print("  round live  phase  crit  c.wait  c.run c.upd  max.upd  mean.run  sum.wait")
    for r in range(n_iter):
        if not has[tix, r]:
            continue
        c = crit[tix, r]
        lv = L[tix, :, r]
        print("  %5d %4d %6.0f %5d %7.0f %6.0f %5.0f %8.0f %9.0f %9.0f" % (
            r, lv.sum(), dur[tix, r] / 1e3, c, wait[tix, c, r] / 1e3, run[tix, c, r] / 1e3, upd[tix, c, r], upd[tix, :, r].max(),
            run[tix, lv, r].mean() / 1e3, wait[tix, lv, r].sum() / 1e3))
