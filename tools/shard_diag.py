#!/usr/bin/env python3
"""Why the 512-sample shard of BASELINE configs[3] takes 2.8 ms on one instance and 4.2 ms on tools/bench_configs.py's (GPU box only):
Newton updates of the slowest samples, and the solve time with them replaced.  r03: 4.18 ms; without the sample with the most
updates (293) 4.13 ms; without the top four (293, 212, 207, 137 updates, 9-17 cuts) 2.67 ms -- the launch ends with whichever
workgroup holds a sample whose 200+ updates run at 15+ cuts (25 k cycles each)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from icnn_amd import bundle_entropy, picnn
spec = picnn.bibtex_spec(); params = picnn.init_params(spec, 0, "spread")
rng = np.random.RandomState(7)
B = 512
x = (rng.rand(B, spec.n_features) < 0.04).astype(np.float32)
model = picnn.FCModel(spec, params); ctx = model.context(torch.from_numpy(x))
def run(ctx, B):
    fs = bundle_entropy.FusedSolver(model, B, 30, "dual")
    for _ in range(2): res = fs.solve(ctx, 0.5)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): res = fs.solve(ctx, 0.5)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 5 * 1e3, res
ms, res = run(ctx, B)
nw = res.newton_iters[:B].cpu().numpy(); ni = res.n_iters[:B].cpu().numpy(); cnt = res.count[:B].cpu().numpy()
top = np.argsort(-nw)[:8]
print("ms %.3f  newton mean %.1f  top %s  n_iters of top %s  cuts of top %s" % (ms, nw.mean(), list(zip(top.tolist(), nw[top].tolist())), ni[top].tolist(), cnt[top].tolist()))
ctx2 = ctx.clone()
for t in top[:1]: ctx2[t] = ctx[(t + 1) % B]
ms2, res2 = run(ctx2, B)
print("without the top sample: %.3f ms, newton max %d" % (ms2, res2.newton_iters[:B].max().item()))
ctx3 = ctx.clone()
for t in top[:4]: ctx3[t] = ctx[(t + 7) % B]
ms3, res3 = run(ctx3, B)
print("without the top four: %.3f ms, newton max %d" % (ms3, res3.newton_iters[:B].max().item()))
