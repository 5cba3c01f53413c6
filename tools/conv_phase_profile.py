#!/usr/bin/env python3
"""Per-phase cycle breakdown of conv_fg_kernel on the completion shape (GPU box only)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from icnn_amd import _lib, picnn  # noqa: E402

PH = ["P0 y load, y*yu0", "P1 y_red_1, z0 (conv k8/s4, 1->32)", "P2 y_red_2", "P3 z1 (conv k4/s2, 32->64)",
      "P4 z2 (conv k3/s1, 64->64)", "P5 z3 (fc 2048->512)", "P6 energy", "P7 delta3", "P8 delta2 (fc^T)",
      "P9 delta1 (convT)", "P10 delta0 (convT)", "P11 dE/dy (convT to image)"]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
spec = picnn.ConvSpec()
params = picnn.init_conv_params(spec, 0, "spread")
x = np.random.RandomState(5).rand(B, spec.H, spec.W, 1).astype(np.float32)
model = picnn.ConvModel(spec, params)
ctx = model.context(torch.from_numpy(x))
y = torch.full((B, spec.n_labels), 0.5, dtype=torch.float64, device="cuda")
for _ in range(3):
    model.fg(ctx, y)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    model.fg(ctx, y)
e1.record()
torch.cuda.synchronize()
print("conv_fg at B=%d: %.1f us per launch" % (B, 100 * e0.elapsed_time(e1)))
prof = torch.zeros(B, 16, dtype=torch.int64, device="cuda")
lib = _lib.load()
lib.icnn_be_debug_profile_conv(C.c_void_p(prof.data_ptr()))
model.fg(ctx, y)
torch.cuda.synchronize()
lib.icnn_be_debug_profile_conv(None)
p = prof.cpu().numpy().astype(np.float64)
tot = p.sum(1)
print("cycles per workgroup: mean %.0f max %.0f (%.1f us at 2.4 GHz)" % (tot.mean(), tot.max(), tot.mean() / 2400))
for i, name in enumerate(PH):
    print("  %-38s mean %9.0f (%5.1f%%)" % (name, p[:, i].mean(), 100 * p[:, i].sum() / tot.sum()))
