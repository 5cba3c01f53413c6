#!/usr/bin/env python3
"""Benchmark of the hot path: fused bundle-entropy inference of the Bibsonomy-shaped PICNN.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one complete solveBatch on a resident minibatch: state reset, then nIter x
{ PICNN energy+gradient kernel, dual-step kernel } (+ one RCCL gather of y* to rank 0 when N > 1).
Workload = BASELINE.json's metric shape: n = 159, K = nIter = 10, batch 4096 per GPU
(weak scaling: every rank solves its own 4096-sample shard, no data-path collective).
Inputs (context, weights, y0) are resident in HBM before the timed region.

Rank 0 prints ONE JSON line; see DESIGN.md "Measurement" for the definitions of
`roofline` and `cpu_baseline`.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from icnn_amd import bundle_entropy, dist as be_dist, picnn  # noqa: E402

PEAK_FP32_TFLOPS = 157.3      # MI355X_MICROARCH.md: dense f32 MFMA = f32 vector peak
PEAK_HBM_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E spec
# HBM bytes per fc_fg launch at batch 4096 from the PMC passes committed in profiles/r01_d_pmc.md:
# (2 * FETCH_SIZE + WRITE_SIZE) * 1024 with the guide's gfx950 FETCH_SIZE correction.
MEASURED_FC_FG_TRAFFIC_BYTES = {4096: (2 * 29200.0 + 2560.0) * 1024}
MEASURED_DUAL_TRAFFIC_BYTES = {4096: (2 * 10700.0 + 20680.0) * 1024}   # incl. 24 spilled VGPRs at occupancy 4
# fused_fc_solve_kernel, same recipe (profiles/r01_g_pmc.md); ~1.2 GB of it is scratch traffic: the two phase
# functions save and restore 48 callee-saved VGPRs per call
MEASURED_FUSED_TRAFFIC_BYTES = {(4096, 10): (2 * 1007100.0 + 1192400.0) * 1024}


def per_kernel_times(model, ctx, B, n_iter, reps):
    """Average duration of each kernel over `reps` complete solves, measured with HIP events on
    the stream the kernels are launched on (torch's current stream is the one handed to the C ABI)."""
    dev = ctx.device
    n = model.spec.n_labels
    y = torch.empty(B, n, dtype=torch.float64, device=dev)
    state = bundle_entropy.BundleState(y, n_iter, "dual", torch.float32)
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(2 * n_iter + 1)] for _ in range(reps)]
    for r in range(reps):
        y.fill_(0.5)
        state.init()
        ev[r][0].record()
        for t in range(n_iter):
            f, g = model.fg(ctx, y, state.finished)
            ev[r][2 * t + 1].record()
            state.step(t, f, g)
            ev[r][2 * t + 2].record()
    torch.cuda.synchronize()
    fg_ms, dual_ms = [], []
    for r in range(reps):
        for t in range(n_iter):
            fg_ms.append(ev[r][2 * t].elapsed_time(ev[r][2 * t + 1]))
            dual_ms.append(ev[r][2 * t + 1].elapsed_time(ev[r][2 * t + 2]))
    return float(np.mean(fg_ms)), float(np.mean(dual_ms)), fg_ms[:n_iter], dual_ms[:n_iter]


def cpu_baseline(params, spec, ctx_rows, n_iter, y_gpu):
    """The oracle (NumPy restatement of the reference solver + PICNN) timed on the host cores on a
    bounded sample of the same workload; also yields max|y* - y*_ref| for those samples."""
    from oracle import bundle_entropy_oracle as oracle
    from oracle import picnn_oracle
    fg = picnn_oracle.make_fg_from_context(params, ctx_rows, list(spec.szs), spec.alpha)
    S = ctx_rows.shape[0]
    y0 = np.full((S, spec.n_labels), 0.5)
    t0 = time.perf_counter()
    with np.errstate(all="ignore"):
        ref = oracle.solve_batch(fg, y0, n_iter)
    wall = time.perf_counter() - t0
    dy = np.max(np.abs(ref.y - y_gpu[:S]), axis=1)
    # bit-tight check: the same oracle solver fed by the PICNN evaluated in the MFMA's float32
    # accumulation order (oracle/picnn_chain.c), so both sides see identical cuts
    fg_chain = picnn_oracle.make_fg_chain(params, ctx_rows, list(spec.szs), spec.alpha)
    with np.errstate(all="ignore"):
        ref_chain = oracle.solve_batch(fg_chain, np.full((S, spec.n_labels), 0.5), n_iter)
    dyc = np.max(np.abs(ref_chain.y - y_gpu[:S]), axis=1)
    executed = int(sum(min(n_iter, it + 2) if it < n_iter else n_iter for it in ref.n_iters))
    return {
        "value": S * n_iter / wall, "unit": "inner-solves/s", "cores": 1, "kind": "port",
        "sample": "first %d samples of the benchmark batch, nIter=%d, 1 run, %.1f s; solver is "
                  "single-threaded NumPy like the reference, PICNN fg uses %d BLAS threads"
                  % (S, n_iter, wall, torch.get_num_threads()),
        "host_cpus": os.cpu_count(),
        "executed_inner_solves": executed,
    }, {
        "samples": int(S),
        "vs_oracle_mfma_order_fp32": {"max_abs_dy": float(dyc.max()), "frac_above_1e-5": float((dyc > 1e-5).mean()),
                                      "note": "oracle PICNN accumulates float32 in the kernel's order "
                                              "(oracle/picnn_chain.c): identical cuts on both sides"},
        "vs_oracle_sgemm_order_fp32": {"max_abs_dy": float(dy.max()), "median_abs_dy": float(np.median(dy)),
                                       "frac_above_1e-5": float((dy > 1e-5).mean()),
                                       "note": "different float32 summation order in the PICNN; the tail is the "
                                               "reference algorithm's own sensitivity (DESIGN.md section 2)"},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4096, help="samples per GPU")
    ap.add_argument("--n-iter", type=int, default=10)
    ap.add_argument("--regime", default="spread")
    ap.add_argument("--cpu-sample", type=int, default=4096, help="samples for the CPU baseline (0 = skip)")
    args = ap.parse_args()

    rank, world, local = be_dist.init_from_env()
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    spec = picnn.bibtex_spec()
    B, n_iter = args.batch, args.n_iter
    params = picnn.init_params(spec, 0, args.regime)
    rng = np.random.RandomState(1000 + rank)
    x = torch.from_numpy((rng.rand(B, spec.n_features) < 0.04).astype(np.float32)).to(dev)
    model = picnn.FCModel(spec, params, dev)
    ctx = model.context(x)                      # x-only, once per minibatch: not part of the hot path
    solver = bundle_entropy.FusedSolver(model, B, n_iter, "dual", dev)

    gather_dst = [0]      # rank 0 collects y*; [None] = all-gather (fallback if the backend refuses gather)

    def step():
        res = solver.solve(ctx, 0.5)
        if world > 1:
            return res, be_dist.gather_rows(res.y, B * world, world, rank, dst=gather_dst[0])
        return res, res.y

    def fence():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    if world > 1:
        try:
            step()
            torch.cuda.synchronize()
        except RuntimeError as e:          # same exception on every rank: all of them switch
            if rank == 0:
                print("gather to rank 0 not available (%s); using all_gather" % str(e)[:120], file=sys.stderr)
            gather_dst[0] = None
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res, y_all = step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())

    ms_per_step = 1e3 * elapsed / args.steps
    inner = world * B * n_iter
    value = inner * args.steps / elapsed
    res.raise_on_error()

    out = {
        "metric": "inner-solves/sec (batch x iters), n=159 K=10", "value": value, "unit": "inner-solves/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32 PICNN / f64 dual solve",
        "data": "synthetic (random-init weights, 'spread' regime; x ~ Bernoulli(0.04))",
        "config": {"workload": "Bibsonomy FC-PICNN 1836->[600,159], fused solveBatch, n=159, nIter=K=%d, "
                               "batch %d per GPU" % (n_iter, B),
                   "variant": "dual (lib/bundle_entropy_dual.py)", "global_batch": B * world,
                   "parallelism": "independent batch shards x%d, one RCCL %s of y*" % (world, "gather to rank 0" if gather_dst[0] == 0 else "all-gather")},
    }

    if rank == 0:
        nact = res.count[:B].float()
        its = res.n_iters[:B].float()
        out["solve_stats"] = {
            "mean_active_cuts": float(nact.mean().item()), "max_active_cuts": int(nact.max().item()),
            "frac_finished_early": float((its < n_iter).float().mean().item()),
            "mean_newton_updates_per_sample": float(res.newton_iters[:B].float().mean().item()),
        }
        fg_ms, dual_ms, fg_list, dual_list = per_kernel_times(model, ctx, B, n_iter, reps=5)
        n = spec.n_labels
        kbar = float(nact.mean().item())
        # fc_fg_kernel: algorithmic flops = fwd + bwd of the y-path, 2 flop per MAC; bytes = context row, y (f64),
        # dE/dy, E per sample + the weights once per launch
        flops = B * 4.0 * spec.y_path_params
        bytes_fg = B * (4.0 * spec.ctx_width + 8.0 * n + 4.0 * n + 4.0) + 4.0 * spec.y_path_params
        # dual_step_kernel: read g (4n) and y (8n), write the cut row (4n), its point (8n), y (8n), h, lam;
        # re-read the k-1 older active rows (4n each)
        bytes_dual = B * (4.0 * n + 8.0 * n + 4.0 * n + 8.0 * n + 8.0 * n + 16.0 + max(kbar - 1.0, 0.0) * 4.0 * n)
        fg_tflops = flops / (fg_ms * 1e-3) / 1e12
        fg_roof = {
            "kernel": "fc_fg_kernel", "bound": "mfma", "achieved": fg_tflops, "peak": PEAK_FP32_TFLOPS,
            "unit": "TFLOP/s", "frac": fg_tflops / PEAK_FP32_TFLOPS,
            "traffic": MEASURED_FC_FG_TRAFFIC_BYTES.get(B), "avg_launch_ms": fg_ms,
            "algorithmic_flops_per_launch": flops, "algorithmic_bytes_per_launch": bytes_fg,
            "hbm_achieved_GBps": bytes_fg / (fg_ms * 1e-3) / 1e9,
        }
        dual_gbs = bytes_dual / (dual_ms * 1e-3) / 1e9
        dual_roof = {
            "kernel": "dual_step_kernel", "bound": "hbm", "achieved": dual_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
            "frac": dual_gbs / PEAK_HBM_GBS, "traffic": MEASURED_DUAL_TRAFFIC_BYTES.get(B),
            "avg_launch_ms": dual_ms, "algorithmic_bytes_per_launch": bytes_dual,
            "note": "per-sample dependency chains (f64 exp, Newton, elimination), not bandwidth, bound this kernel: "
                    "DESIGN.md section 4",
        }
        # Which kernels did the timed solves launch?  With the persistent per-tile kernel (default for this batch)
        # the whole solve is ONE launch of fused_fc_solve_kernel, whose two phases are the device functions of the
        # two kernels above: `roofline` then describes that launch, the per-phase kernels (timed one launch per
        # round through the two-kernel entry points) are reported next to it.
        tiles = (B + 15) // 16
        cus = torch.cuda.get_device_properties(dev).multi_processor_count
        persistent = n_iter <= 15 and 4 * tiles >= cus and tiles <= 2 * cus
        if persistent:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            launch_ms = []
            for _ in range(5):
                solver.y.fill_(0.5)
                solver.state.init()
                e0.record()
                rounds = getattr(solver.state.lib, model.solve_entry)(
                    C.byref(model.c_model), ctx.data_ptr(), C.byref(solver.state.c_state),
                    solver.f_work.data_ptr(), solver.g_work.data_ptr(), solver.state.stream())
                e1.record()
                torch.cuda.synchronize()
                assert rounds == n_iter
                launch_ms.append(e0.elapsed_time(e1))
            fused_ms = float(np.mean(launch_ms))
            fl, by = n_iter * flops, n_iter * (bytes_fg + bytes_dual)
            tf = fl / (fused_ms * 1e-3) / 1e12
            out["roofline"] = {
                "kernel": "fused_fc_solve_kernel (persistent: %d rounds of {fc_fg tile phase ; dual-step phase})" % n_iter,
                "bound": "mfma", "achieved": tf, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": tf / PEAK_FP32_TFLOPS,
                "traffic": MEASURED_FUSED_TRAFFIC_BYTES.get((B, n_iter)), "avg_launch_ms": fused_ms,
                "algorithmic_flops_per_launch": fl, "algorithmic_bytes_per_launch": by,
                "hbm_achieved_GBps": by / (fused_ms * 1e-3) / 1e9, "hbm_frac": by / (fused_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                "note": "a launch is as long as its slowest tile's chain of 2 x %d phases; neither the MFMA pipes nor HBM "
                        "are the limit (DESIGN.md section 4)" % n_iter,
            }
            out["roofline_phase_kernels"] = {"fc_fg_kernel": fg_roof, "dual_step_kernel": dual_roof}
        else:
            out["roofline"] = dual_roof if dual_ms >= fg_ms else fg_roof
            out["roofline_other_kernel"] = fg_roof if dual_ms >= fg_ms else dual_roof
        out["per_iteration_ms"] = {"fc_fg": [round(v, 4) for v in fg_list], "dual_step": [round(v, 4) for v in dual_list]}
        if world == 1 and args.cpu_sample > 0:
            S = min(args.cpu_sample, B)
            base, parity = cpu_baseline(params, spec, ctx[:S].cpu().numpy(), n_iter, res.y.cpu().numpy())
            out["cpu_baseline"] = base
            out["parity"] = parity
        print(json.dumps(out))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
