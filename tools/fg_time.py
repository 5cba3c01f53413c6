#!/usr/bin/env python3
"""Time fc_fg_kernel alone on the benchmark shape (GPU box only)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icnn_amd import picnn
spec = picnn.bibtex_spec(); params = picnn.init_params(spec, 0, "spread")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
x = torch.from_numpy((np.random.RandomState(1).rand(B, spec.n_features) < 0.04).astype(np.float32)).cuda()
model = picnn.FCModel(spec, params); ctx = model.context(x)
y = torch.rand(B, spec.n_labels, dtype=torch.float64, device="cuda")
for _ in range(5): model.fg(ctx, y)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): model.fg(ctx, y)
e1.record(); torch.cuda.synchronize()
print("fc_fg B=%d: %.1f us per launch" % (B, e0.elapsed_time(e1) * 1e3 / 50))
