#!/usr/bin/env python3
"""Upper bound of what re-tiling could buy the persistent tile kernel (VERDICT r5 #3c; GPU box only):
    python tools/sorted_batch_bound.py
A tile of 16 samples waits for its slowest sample in every round.  With the per-sample cost known from a first solve, the
batch is permuted so that tiles hold samples of similar cost (sorted by Newton updates, then by active cuts) and solved
again: samples are independent given their context rows, so every sample's result is bit-identical -- only the tiling
changes.  The difference is what a scheme that knew the costs in advance could gain at most; a random permutation is the
control.  Tool only: nothing in the library sorts."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from icnn_amd import bundle_entropy, picnn  # noqa: E402

spec = picnn.bibtex_spec()
params = picnn.init_params(spec, 0, "spread")
B = 4096
x = torch.from_numpy((np.random.RandomState(1000).rand(B, spec.n_features) < 0.04).astype(np.float32)).cuda()
model = picnn.FCModel(spec, params)
ctx = model.context(x)


def timed(fs, c, reps):
    for _ in range(3):
        res = fs.solve(c, 0.5)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        res = fs.solve(c, 0.5)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps, res


for n_iter in (10, 30):
    fs = bundle_entropy.FusedSolver(model, B, n_iter, "dual")
    reps = 20 if n_iter == 10 else 6
    t0, res = timed(fs, ctx, reps)
    upd = res.newton_iters[:B].cpu().numpy().astype(np.int64)
    cnt = res.count[:B].cpu().numpy().astype(np.int64)
    y0 = res.y.cpu().numpy().copy()
    order = np.lexsort((cnt, upd))                                      # by updates, then by bundle size
    rnd = np.random.RandomState(3).permutation(B)
    out = {}
    for name, perm in (("sorted by total Newton updates", order), ("random permutation (control)", rnd)):
        p = torch.from_numpy(perm).cuda()
        t, r = timed(fs, ctx[p].contiguous(), reps)
        assert np.array_equal(r.y.cpu().numpy(), y0[perm]), "a sample's result must not depend on its tile"
        out[name] = t
    tile_max = upd.reshape(-1, 16).max(1)
    srt_max = upd[order].reshape(-1, 16).max(1)
    print("4096 x %d: as is %.4f ms | %s" % (n_iter, t0, " | ".join("%s %.4f ms (%+.1f %%)" % (k, v, 100 * (v / t0 - 1)) for k, v in out.items())))
    print("   Newton updates per sample: mean %.1f, max %d; mean over tiles of the tile's maximum: as is %.1f, sorted %.1f"
          % (upd.mean(), upd.max(), tile_max.mean(), srt_max.mean()))
