#!/usr/bin/env python3
"""Would sharding one batch over S HIP streams (independent sub-batches, each running its own
{fc_fg ; dual_step} x nIter chain) shorten the solve?  Emulated with S FusedSolvers on S torch streams."""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from icnn_amd import bundle_entropy, picnn  # noqa: E402

B, n_iter = 4096, 10
spec = picnn.bibtex_spec()
params = picnn.init_params(spec, 0, "spread")
x = torch.from_numpy((np.random.RandomState(1000).rand(B, spec.n_features) < 0.04).astype(np.float32)).cuda()
model = picnn.FCModel(spec, params)
ctx = model.context(x)
for S in (1, 2, 4, 8, 16):
    bs = B // S
    solvers = [bundle_entropy.FusedSolver(model, bs, n_iter) for _ in range(S)]
    ctxs = [ctx[i * bs:(i + 1) * bs].contiguous() for i in range(S)]
    streams = [torch.cuda.Stream() for _ in range(S)]
    main = torch.cuda.current_stream()

    def run():
        ev0 = torch.cuda.Event()
        ev0.record(main)
        for s, sol, c in zip(streams, solvers, ctxs):
            s.wait_event(ev0)
            with torch.cuda.stream(s):
                sol.solve(c)
        for s in streams:
            ev = torch.cuda.Event()
            ev.record(s)
            main.wait_event(ev)

    for _ in range(3):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    K = 10
    for _ in range(K):
        run()
    e1.record()
    torch.cuda.synchronize()
    print("S=%2d shards of %4d: %.3f ms per solve (GPU events), %.3f ms wall" %
          (S, bs, e0.elapsed_time(e1) / K, (time.perf_counter() - t0) * 1e3 / K), flush=True)
