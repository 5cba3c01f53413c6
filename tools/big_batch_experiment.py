#!/usr/bin/env python3
"""Batches beyond two tiles per CU (8192 < B): the default (launch pairs), the persistent tile kernel forced, and the batch
split over S streams with a persistent tile kernel each (tiles of different streams can overlap their phases).  GPU box only."""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from icnn_amd import _lib, bundle_entropy, picnn  # noqa: E402

n_iter = 10
spec = picnn.bibtex_spec()
params = picnn.init_params(spec, 0, "spread")
model = picnn.FCModel(spec, params)
for B in (8192, 16384):
    x = torch.from_numpy((np.random.RandomState(1000).rand(B, spec.n_features) < 0.04).astype(np.float32)).cuda()
    ctx = model.context(x)
    for name, flags, S in (("default", 0, 1), ("persistent tiles", _lib.FLAG_PERSISTENT, 1), ("two kernels", _lib.FLAG_TWO_KERNELS, 1),
                           ("persistent x 2 streams", _lib.FLAG_PERSISTENT, 2), ("persistent x 4 streams", _lib.FLAG_PERSISTENT, 4)):
        bs = B // S
        solvers = [bundle_entropy.FusedSolver(model, bs, n_iter, flags=flags) for _ in range(S)]
        ctxs = [ctx[i * bs:(i + 1) * bs].contiguous() for i in range(S)]
        streams = [torch.cuda.Stream() for _ in range(S)] if S > 1 else [torch.cuda.current_stream()]
        main = torch.cuda.current_stream()

        def run():
            if S == 1:
                solvers[0].solve(ctxs[0])
                return
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
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        K = 8
        for _ in range(K):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / K
        print("B=%5d %-24s %.3f ms  %.1f M inner-solves/s  frac %.3f" % (B, name, ms, B * n_iter / ms / 1e3,
              B * n_iter * 4 * spec.y_path_params / (ms * 1e-3) / 157.3e12), flush=True)
