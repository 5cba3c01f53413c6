#!/usr/bin/env python3
"""Stepwise solve on the benchmark workload (fc_fg + dual_step launched round by round through the
C ABI): per round, the dual step's duration and the distribution of Newton updates per sample."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from icnn_amd import bundle_entropy, picnn  # noqa: E402

n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
spec = picnn.bibtex_spec()
params = picnn.init_params(spec, 0, "spread")
x = torch.from_numpy((np.random.RandomState(1000).rand(B, spec.n_features) < 0.04).astype(np.float32)).cuda()
model = picnn.FCModel(spec, params)
ctx = model.context(x)
for rep in range(2):
    y = torch.full((B, spec.n_labels), 0.5, dtype=torch.float64, device="cuda")
    st = bundle_entropy.BundleState(y, n_iter, "dual")
    st.init()
    prev = torch.zeros(B, dtype=torch.int32, device="cuda")
    for t in range(n_iter):
        f, g = model.fg(ctx, y, st.finished)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        st.step(t, f, g)
        e1.record()
        torch.cuda.synchronize()
        now = st.newton_iters.clone()
        d = (now - prev).cpu().numpy()
        prev = now
        if rep == 1:
            cnt = st.count.cpu().numpy()
            top = np.sort(d)[-6:][::-1]
            print("round %2d: dual %.3f ms  newton updates mean %.1f  p50 %d p99 %d  top %s  (>=40: %d)  cuts mean %.1f"
                  % (t, e0.elapsed_time(e1), d.mean(), np.percentile(d, 50), np.percentile(d, 99), list(top),
                     int((d >= 40).sum()), cnt.mean()))
