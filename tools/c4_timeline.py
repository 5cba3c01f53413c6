#!/usr/bin/env python3
"""Where a round of the persistent tile kernel goes at nIter > 15 (BASELINE configs[3]: 4096 x 30; GPU box only):
    python tools/c4_timeline.py [B [nIter ...]]
Profiling variant of the library: per sample the cycles its wave spends (a) running the dual step (phases 0-13 of
icnn_be_debug_profile), (b) queueing for its LDS staging region (phase 14), (c) at the barrier that ends the tile's dual phase
(phase 15).  A run per nIter: the differences between them are the rounds in between (a solve of nIter' < nIter iterations is the
first nIter' rounds of the longer one)."""
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
iters = [int(v) for v in sys.argv[2:]] or [16, 20, 25, 30]
spec = picnn.bibtex_spec()
params = picnn.init_params(spec, 0, "spread")
x = torch.from_numpy((np.random.RandomState(1000).rand(B, spec.n_features) < 0.04).astype(np.float32)).cuda()
model = picnn.FCModel(spec, params)
ctx = model.context(x)
lib = _lib.load()
NPH = lib.icnn_be_debug_profile_phases()
prev = None
for n_iter in iters:
    solver = bundle_entropy.FusedSolver(model, B, n_iter, "dual")
    solver.solve(ctx)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3):
        solver.solve(ctx)
    b.record()
    torch.cuda.synchronize()
    ms_plain = a.elapsed_time(b) / 3
    prof = torch.zeros(max(B, 4096) + 8, NPH, dtype=torch.int64, device="cuda")
    lib.icnn_be_debug_profile(C.c_void_p(prof.data_ptr()))
    a.record()
    res = solver.solve(ctx)
    b.record()
    torch.cuda.synchronize()
    lib.icnn_be_debug_profile(None)
    ms = a.elapsed_time(b)
    p = prof.cpu().numpy().astype(np.float64)[:B]
    run, queue, bar = p[:, :14].sum(1), p[:, 14], p[:, 15]
    cnt = res.count[:B].cpu().numpy()
    upd = res.newton_iters[:B].cpu().numpy()
    fin = res.finished[:B].cpu().numpy()
    tiles = B // 16
    # per tile: the wave with the largest run + queue is the one the others wait for
    per_tile = (run + queue + bar).reshape(tiles, 16).max(1)
    print("nIter %2d: %.3f ms (%.3f with laps), %.0f k cycles per round at 2.4 GHz" % (n_iter, ms_plain, ms, ms_plain * 2.4e3 / n_iter))
    print("   per sample over the solve (k cycles): run mean %.0f max %.0f | queue for LDS mean %.0f max %.0f | end barrier mean %.0f"
          % (run.mean() / 1e3, run.max() / 1e3, queue.mean() / 1e3, queue.max() / 1e3, bar.mean() / 1e3))
    print("   dual phase of a tile (longest wave, run + queue + barrier): mean %.0f k, max %.0f k; phase A + rest = %.0f k per round"
          % (per_tile.mean() / 1e3, per_tile.max() / 1e3, (ms_plain * 2.4e6 - per_tile.max()) / n_iter / 1e3))
    print("   active cuts mean %.1f max %d, finished early %.1f %%, newton updates mean %.1f max %d"
          % (cnt.mean(), cnt.max(), 100 * fin.mean(), upd.mean(), upd.max()))
    if prev is not None:
        d_it = n_iter - prev[0]
        print("   rounds %d..%d: %.0f k cycles per round; per sample and round: run %.1f k, queue %.1f k, barrier %.1f k, updates %.2f"
              % (prev[0], n_iter - 1, (ms_plain - prev[1]) * 2.4e3 / d_it, (run.mean() - prev[2]) / d_it / 1e3,
                 (queue.mean() - prev[3]) / d_it / 1e3, (bar.mean() - prev[4]) / d_it / 1e3, (upd.mean() - prev[5]) / d_it))
    prev = (n_iter, ms_plain, run.mean(), queue.mean(), bar.mean(), upd.mean())
    if n_iter == iters[-1]:
        tot_tile = (run + queue + bar).reshape(tiles, 16).max(1)
        print("   tile dual-phase totals (k cycles): p10 %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f"
              % tuple(np.percentile(tot_tile, q) / 1e3 for q in (10, 50, 90, 99, 100)))
        for tix in np.argsort(-tot_tile)[:4]:
            sl = slice(tix * 16, tix * 16 + 16)
            print("   tile %d (total %.0f k):" % (tix, tot_tile[tix] / 1e3))
            print("      run     " + " ".join("%5.0f" % (v / 1e3) for v in run[sl]))
            print("      queue   " + " ".join("%5.0f" % (v / 1e3) for v in queue[sl]))
            print("      barrier " + " ".join("%5.0f" % (v / 1e3) for v in bar[sl]))
            print("      updates " + " ".join("%5d" % v for v in upd[sl]))
            print("      cuts    " + " ".join("%5d" % v for v in cnt[sl]))
            print("      nIters  " + " ".join("%5d" % v for v in res.n_iters[:B].cpu().numpy()[sl]))
        # cost per update as a function of the final bundle size (samples that ran all rounds)
        full = fin == 0
        for lo, hi in ((1, 8), (9, 12), (13, 16), (17, 20), (21, 31)):
            m = full & (cnt >= lo) & (cnt <= hi)
            if m.any():
                print("   final cuts %2d..%2d: %4d samples, run %.0f k, updates %.1f, run / update %.1f k"
                      % (lo, hi, m.sum(), run[m].mean() / 1e3, upd[m].mean(), run[m].sum() / upd[m].sum() / 1e3))
