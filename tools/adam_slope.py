"""Time per PICNN evaluation inside the persistent Adam kernels (GPU box only): the stopping rule cannot fire before
iteration 6, so calls with max_iter 2 and 6 differ by exactly four evaluations -- the slope is free of launch and
synchronisation overhead.  B <= 4 runs adam_rows_kernel, larger batches adam_fc_kernel."""
import dataclasses, sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from icnn_amd import picnn, rl_adam
spec = dataclasses.replace(picnn.halfcheetah_spec(), action_box=False)
params = picnn.init_params(spec, 0, "spread", yu_bias=1.0, gate_bias=1.0)
model = picnn.FCModel(spec, params)
for B in (1, 4):
    obs = np.random.RandomState(5).randn(64, spec.n_features).astype(np.float32)
    ctx = model.context(torch.from_numpy(obs))[:B].contiguous()
    out = {}
    for mi in (1, 2, 6):
        solver = rl_adam.AdamSolver(model, B, mi)
        for _ in range(5): solver.solve(ctx)
        torch.cuda.synchronize()
        ts = []
        for _ in range(200):
            t0 = time.perf_counter(); solver.solve(ctx); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        out[mi] = 1e6 * float(np.median(ts))
    print("B=%d: call us at max_iter 1/2/6: %.1f %.1f %.1f -> %.2f us per evaluation" % (B, out[1], out[2], out[6], (out[6] - out[2]) / 4))
