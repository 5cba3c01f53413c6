#!/usr/bin/env python3
"""Time and parity-check every BASELINE.json config shape on one MI355X (GPU box only).

Not the driver's bench (that is bench.py); this records the per-config evidence quoted in
DESIGN.md: fused solve time, inner-solves/s, and max|y* - y*_ref| against the CPU oracle on a
bounded sample of the same inputs.  Writes gpurun_out/configs.json.
"""
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

from icnn_amd import bundle_entropy, picnn  # noqa: E402
from oracle import bundle_entropy_oracle as oracle  # noqa: E402
from oracle import picnn_conv_oracle, picnn_oracle  # noqa: E402
import problems  # noqa: E402


def timed(solver, ctx, y0, reps=10):
    for _ in range(2):
        solver.solve(ctx, y0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        res = solver.solve(ctx, y0)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, res


def parity(res, fg, y0, n_iter, variant, S):
    with np.errstate(all="ignore"):
        t0 = time.perf_counter()
        ora = oracle.solve_batch(fg, y0[:S].copy(), n_iter, variant=variant)
        wall = time.perf_counter() - t0
    dy = np.max(np.abs(res.y[:S].cpu().numpy() - ora.y), axis=1)
    return {"samples": S, "max_abs_dy": float(dy.max()), "median_abs_dy": float(np.median(dy)),
            "frac_above_1e-5": float((dy > 1e-5).mean()), "cpu_oracle_inner_solves_per_s": S * n_iter / wall}


def fc_config(name, spec, B, n_iter, variant, regime, kw, S, chain=True):
    params = picnn.init_params(spec, 0, regime, **kw)
    rng = np.random.RandomState(7)
    x = (rng.rand(B, spec.n_features) < 0.04).astype(np.float32) if spec.n_features > 100 \
        else rng.randn(B, spec.n_features).astype(np.float32)
    model = picnn.FCModel(spec, params)
    ctx = model.context(torch.from_numpy(x))
    solver = bundle_entropy.FusedSolver(model, B, n_iter, variant)
    sec, res = timed(solver, ctx, 0.5)
    y0 = np.full((B, spec.n_labels), 0.5)
    fg = picnn_oracle.make_fg_chain(params, ctx[:S].cpu().numpy(), list(spec.szs), spec.alpha, spec.action_box)
    out = {"config": name, "batch": B, "n": spec.n_labels, "nIter": n_iter, "variant": variant, "regime": regime,
           "ms_per_solve": 1e3 * sec, "inner_solves_per_s": B * n_iter / sec,
           "parity_vs_mfma_order_oracle": parity(res, fg, y0, n_iter, variant, S),
           "mean_active_cuts": float(res.count[:B].float().mean().item()),
           "frac_finished_early": float(res.finished[:B].float().mean().item())}
    fg2 = picnn_oracle.make_fg_from_context(params, ctx[:S].cpu().numpy(), list(spec.szs), spec.alpha,
                                            "action" if spec.action_box else None)
    out["parity_vs_sgemm_order_oracle"] = parity(res, fg2, y0, n_iter, variant, S)
    return out


def latency_config():
    """the RL actor's act() shape -- ONE state, n = 6 actions, nIter = 5 -- where the
    solve is launch-latency bound.  Reports the synchronous latency of a fused solve (11 launches) and of
    the same launches replayed from a HIP graph (torch.cuda.CUDAGraph capturing the C-ABI calls)."""
    spec = picnn.halfcheetah_spec()
    params = picnn.init_params(spec, 0, "spread", **{})
    x = np.random.RandomState(3).randn(1, spec.n_features).astype(np.float32)
    model = picnn.FCModel(spec, params)
    xs = np.random.RandomState(4).randn(64, spec.n_features).astype(np.float32)    # BatchNorm needs a batch
    xs[0] = x[0]
    ctx = model.context(torch.from_numpy(xs))[:1].contiguous()
    solver = bundle_entropy.FusedSolver(model, 1, 5, "rl")
    for _ in range(5):
        solver.solve(ctx, 0.5)
    torch.cuda.synchronize()
    lat = []
    for _ in range(200):
        t0 = time.perf_counter()
        solver.solve(ctx, 0.5)
        torch.cuda.synchronize()
        lat.append(time.perf_counter() - t0)
    out = {"config": "f4 latency: RL act() shape B=1 n=6 nIter=5 (variant rl)",
           "sync_latency_us_median": 1e6 * float(np.median(lat)), "sync_latency_us_p90": 1e6 * float(np.percentile(lat, 90))}
    y_eager = solver.y.clone()
    try:
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            solver.solve(ctx, 0.5)
        torch.cuda.current_stream().wait_stream(s)
        with torch.cuda.graph(g):
            solver.solve(ctx, 0.5)
        torch.cuda.synchronize()
        lat = []
        for _ in range(200):
            t0 = time.perf_counter()
            g.replay()
            torch.cuda.synchronize()
            lat.append(time.perf_counter() - t0)
        out["graph_replay_latency_us_median"] = 1e6 * float(np.median(lat))
        out["graph_equals_eager"] = bool(torch.equal(solver.y, y_eager))
    except Exception as e:                                   # noqa: BLE001 -- diagnostic tool
        out["graph_replay_error"] = repr(e)[:200]
    return out


def conv_config(B, n_iter, S, name="C3 completion conv-PICNN", chain=False):
    spec = picnn.ConvSpec()
    params = picnn.init_conv_params(spec, 0, "spread")
    x = np.random.RandomState(5).rand(B, spec.H, spec.W, 1).astype(np.float32)[:, :, ::-1, :].copy()   # h-flip (:215)
    model = picnn.ConvModel(spec, params)
    ctx = model.context(torch.from_numpy(x))
    mean_img = 0.2 + 0.6 * np.random.RandomState(9).rand(spec.n_labels)
    y0 = np.repeat(mean_img[None], B, axis=0)
    y0_dev = torch.from_numpy(y0).cuda()
    solver = bundle_entropy.FusedSolver(model, B, n_iter, "dual")
    sec, res = timed(solver, ctx, y0_dev, reps=5)
    # oracle PICNN: torch-CPU autograd (another float32 summation order) or the kernel-order C chain (identical cuts)
    fg = (picnn_conv_oracle.make_fg_chain if chain else picnn_conv_oracle.make_fg_from_context)(
        params, ctx[:S].cpu().numpy(), spec.H, spec.W)
    xd = torch.from_numpy(x).cuda()                          # image -> context (icnn_be_conv_context) -> solve
    for _ in range(2):
        solver.solve(model.context(xd), y0_dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        solver.solve(model.context(xd), y0_dev)
    torch.cuda.synchronize()
    sec_img = (time.perf_counter() - t0) / 5
    return {"config": name, "batch": B, "n": spec.n_labels, "nIter": n_iter, "variant": "dual",
            "regime": "spread", "ms_per_solve": 1e3 * sec, "inner_solves_per_s": B * n_iter / sec,
            "ms_from_image": 1e3 * sec_img,
            ("parity_vs_kernel_order_oracle" if chain else "parity_vs_torch_conv_oracle"): parity(res, fg, y0, n_iter, "dual", S),
            "mean_active_cuts": float(res.count[:B].float().mean().item()),
            "max_active_cuts": int(res.count[:B].max().item()),
            "overflowed": int((res.status[:B] & 4).ne(0).sum().item())}


def c1_config():
    factory, n_iter = problems.GOLDEN_CASES["c1_quadratic"]
    prob = factory()
    y0 = prob.y0()
    res = bundle_entropy.solveBatch(prob.fg, y0, nIter=n_iter, native=True)
    with np.errstate(all="ignore"):
        ora = oracle.solve_batch(prob.fg, prob.y0(), n_iter)
    return {"config": "C1 quadratic n=4 B=32 nIter=10 (generic fg, float64 cuts)",
            "max_abs_dy": float(np.max(np.abs(res.y.cpu().numpy() - ora.y))), "sum_y": float(res.y.sum().item())}


def adam_config():
    """`Agent.adam` (the RL default inner optimiser) as one persistent launch.  B = 1 is the
    act() shape, B = 256 the default training minibatch (RL/src/agent.py:7) of next observations, 4096 the largest batch one cooperative launch holds; CPU = the NumPy oracle
    with the float32 sgemm-order PICNN on this box's cores."""
    import dataclasses

    from icnn_amd import rl_adam
    from oracle import adam_oracle
    spec = dataclasses.replace(picnn.halfcheetah_spec(), action_box=False)
    params = picnn.init_params(spec, 0, "spread", yu_bias=1.0, gate_bias=1.0)
    model = picnn.FCModel(spec, params)
    rows = []
    for B in (1, 32, 256, 1024, 4096):
        obs = np.random.RandomState(5).randn(max(B, 64), spec.n_features).astype(np.float32)
        ctx = model.context(torch.from_numpy(obs))[:B].contiguous()
        solver = rl_adam.AdamSolver(model, B)
        for _ in range(3):
            res = solver.solve(ctx)
        torch.cuda.synchronize()
        ts = []
        for _ in range(20):
            t0 = time.perf_counter()
            res = solver.solve(ctx)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        sec, iters = float(np.median(ts)), int(res.iters.item())
        row = {"B": B, "iterations": iters, "ms_per_call": 1e3 * sec, "us_per_iteration": 1e6 * sec / max(iters, 1),
               "evaluations_per_s": B * (iters + 1) / sec}
        if B <= 1024:
            fg2 = picnn_oracle.make_fg_from_context(params, ctx.cpu().numpy(), list(spec.szs), spec.alpha, None)
            func = adam_oracle.entropy_fg(lambda o, a: fg2(a))
            t0 = time.perf_counter()
            best, it_cpu, _ = adam_oracle.adam(func, ctx.cpu().numpy(), spec.n_labels)
            row["cpu_oracle_ms"] = 1e3 * (time.perf_counter() - t0)
            row["cpu_oracle_iterations"] = it_cpu
            row["max_abs_d_act_best_vs_sgemm_order_oracle"] = float(np.max(np.abs(best - res.act_best.cpu().numpy())))
        rows.append(row)
    return {"config": "f4 adam: RL inner optimiser, 17 obs -> 6 actions, 200-200 PICNN, max_iter 1000", "rows": rows}


def main():
    only = sys.argv[1] if len(sys.argv) > 1 else ""
    out = []
    if not only or only == "C1":
        out.append(c1_config())
    if not only or only == "C2":
        out.append(fc_config("C2 Bibsonomy B=128 nIter=10", picnn.bibtex_spec(), 128, 10, "dual", "spread", {}, 128))
    if not only or only == "C3":
        out.append(conv_config(256, 5, 48))
        # the reference's default number of bundle iterations for this model (completion/icnn_ebundle.py:41): bundles outgrow
        # the 12 cuts a workgroup stages in LDS and continue in device memory
        out.append(conv_config(256, 30, 6, "C3 at the reference default nBundleIter=30", chain=True))
    if not only or only == "C4":
        out.append(fc_config("C4 shard (512 of 4096) nIter=30", picnn.bibtex_spec(), 512, 30, "dual", "spread", {},
                             128))
        out.append(fc_config("C4 whole batch on one GPU B=4096 nIter=30", picnn.bibtex_spec(), 4096, 30, "dual",
                             "spread", {}, 128))
    if not only or only == "F4":
        out.append(latency_config())
        out.append(adam_config())
    if not only or only == "C5":
        out.append(fc_config("C5 RL HalfCheetah B=8192 nIter=5", picnn.halfcheetah_spec(), 8192, 5, "rl", "spread",
                             dict(yu_bias=1.0, gate_bias=1.0), 1024))
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    with open(os.path.join(REPO, "gpurun_out", "configs%s.json" % ("_" + only if only else "")), "w") as fh:
        json.dump(out, fh, indent=1)
    for o in out:
        print(json.dumps(o))


if __name__ == "__main__":
    main()
