#!/usr/bin/env python3
"""One BASELINE.json shape, a few solves, nothing else: the process rocprofv3 wraps in tools/prof_shapes.sh (GPU box only).

    python tools/prof_target.py c2|c2pdipm|pdipm|shard512|c4|c4shard|c3|c3n30|c3pdipm|c3n30pdipm|c5|adam [solves]
Prints a JSON line {shape, batch, n_iter, variant, ms_per_solve}."""
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from icnn_amd import bundle_entropy, picnn  # noqa: E402

shape = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
out = {"shape": shape}
if shape in ("c4", "c4shard", "c2", "shard512", "c2pdipm", "pdipm"):
    spec = picnn.bibtex_spec()
    # shard512: the 8-GPU shard of the north-star batch; c2pdipm / pdipm: the interior-point variant (the module the reference's
    # scripts import) on configs[1] and on the benchmark batch
    B, n_iter, variant = {"c4": (4096, 30, "dual"), "c4shard": (512, 30, "dual"), "c2": (128, 10, "dual"),
                          "shard512": (512, 10, "dual"), "c2pdipm": (128, 10, "pdipm"), "pdipm": (4096, 10, "pdipm")}[shape]
    params = picnn.init_params(spec, 0, "spread")
    x = (np.random.RandomState(1000).rand(4096, spec.n_features) < 0.04).astype(np.float32)
    model = picnn.FCModel(spec, params)
    ctx = model.context(torch.from_numpy(x))[:B].contiguous()          # the shard of the bench batch
    y0 = 0.5
elif shape == "c5":
    spec = picnn.halfcheetah_spec()
    B, n_iter, variant = 8192, 5, "rl"
    params = picnn.init_params(spec, 0, "spread", yu_bias=1.0, gate_bias=1.0)
    x = np.random.RandomState(7).randn(B, spec.n_features).astype(np.float32)
    model = picnn.FCModel(spec, params)
    ctx = model.context(torch.from_numpy(x))
    y0 = 0.5
elif shape in ("c3", "c3n30", "c3pdipm", "c3n30pdipm"):
    spec = picnn.ConvSpec()
    B, n_iter, variant = 256, 30 if "n30" in shape else 5, "pdipm" if "pdipm" in shape else "dual"
    params = picnn.init_conv_params(spec, 0, "spread")
    x = np.random.RandomState(5).rand(B, spec.H, spec.W, 1).astype(np.float32)[:, :, ::-1, :].copy()
    model = picnn.ConvModel(spec, params)
    ctx = model.context(torch.from_numpy(x))
    y0 = torch.from_numpy(np.repeat((0.2 + 0.6 * np.random.RandomState(9).rand(spec.n_labels))[None], B, axis=0)).cuda()
elif shape == "adam":
    import dataclasses
    from icnn_amd import rl_adam
    spec = dataclasses.replace(picnn.halfcheetah_spec(), action_box=False)
    B, n_iter, variant = 256, 0, "adam"
    params = picnn.init_params(spec, 0, "spread", yu_bias=1.0, gate_bias=1.0)
    model = picnn.FCModel(spec, params)
    ctx = model.context(torch.from_numpy(np.random.RandomState(5).randn(B, spec.n_features).astype(np.float32)))
    solver = rl_adam.AdamSolver(model, B)
else:
    raise SystemExit("unknown shape %r" % shape)

if shape == "adam":
    run = lambda: solver.solve(ctx)                                   # noqa: E731
else:
    fs = bundle_entropy.FusedSolver(model, B, n_iter, variant)
    run = lambda: fs.solve(ctx, y0)                                   # noqa: E731
run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    res = run()
torch.cuda.synchronize()
out.update(batch=B, n_iter=n_iter, variant=variant, ms_per_solve=1e3 * (time.perf_counter() - t0) / reps, solves=reps + 1)
print(json.dumps(out))
