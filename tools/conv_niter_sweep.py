import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from icnn_amd import bundle_entropy, picnn
B = 256
spec = picnn.ConvSpec(); params = picnn.init_conv_params(spec, 1, "spread")
x = np.random.RandomState(51).rand(B, spec.H, spec.W, 1).astype(np.float32)
model = picnn.ConvModel(spec, params); ctx = model.context(torch.from_numpy(x))
y = torch.from_numpy(0.2 + 0.6 * np.random.RandomState(9).rand(B, spec.n_labels)).cuda()
for n_iter in (5, 15, 30):
    solver = bundle_entropy.FusedSolver(model, B, n_iter, "dual")
    for _ in range(2): res = solver.solve(ctx, y)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): res = solver.solve(ctx, y)
    torch.cuda.synchronize()
    print("conv solve B=256 nIter=%d: %.2f ms; newton mean %.1f max %d; cuts mean %.2f max %d" % (n_iter, (time.perf_counter() - t0) / 3 * 1e3,
          res.newton_iters[:B].float().mean().item(), res.newton_iters[:B].max().item(), res.count[:B].float().mean().item(), res.count[:B].max().item()))
