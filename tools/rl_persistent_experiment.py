"""RL variant (RL/src/bundle_entropy.py) through the persistent per-tile kernel vs one launch per phase:
bit equality of every output and time per solve over the batch sizes of the RL agent (act(): 1 state,
train(): a minibatch) and of config C5.  usage: python tools/rl_persistent_experiment.py"""
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from icnn_amd import _lib, bundle_entropy, picnn   # noqa: E402


def timed(solver, ctx, reps):
    for _ in range(3):
        solver.solve(ctx, 0.5)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        res = solver.solve(ctx, 0.5)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)), res


def main():
    spec = picnn.halfcheetah_spec()
    params = picnn.init_params(spec, 0, "spread", yu_bias=1.0, gate_bias=1.0)
    model = picnn.FCModel(spec, params)
    out = []
    for B in (1, 16, 64, 128, 1024, 8192):
        xs = np.random.RandomState(4).randn(max(B, 64), spec.n_features).astype(np.float32)
        ctx = model.context(torch.from_numpy(xs))[:B].contiguous()
        row = {"B": B}
        outs = {}
        for name, flags in (("persistent", _lib.FLAG_PERSISTENT), ("two_kernels", _lib.FLAG_TWO_KERNELS), ("default", 0)):
            solver = bundle_entropy.FusedSolver(model, B, 5, "rl", flags=flags)
            sec, res = timed(solver, ctx, 50 if B <= 1024 else 20)
            row[name + "_us"] = 1e6 * sec
            outs[name] = [t.cpu().numpy().copy() for t in (res.y, res.lam, res.active, res.count[:B], res.n_iters[:B],
                                                             res.finished[:B], res.status[:B])]
        row["bit_identical"] = all(np.array_equal(a, b) for a, b in zip(outs["persistent"], outs["two_kernels"]))
        out.append(row)
        print(json.dumps(row), flush=True)
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    with open(os.path.join(REPO, "gpurun_out", "rl_persistent.json"), "w") as fh:
        json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
