#!/usr/bin/env python3
"""Benchmark of the hot path: fused bundle-entropy inference of the Bibsonomy-shaped PICNN.

    python bench.py --gpus N --steps K --warmup W [--scaling strong|weak]
    N > 1 from a plain shell: bench.py starts its own N ranks (self_launch: one child process per GPU, rank 0 prints);
    under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...` (WORLD_SIZE set)
    it is one of the ranks.

One "step" = one complete solveBatch on a resident minibatch: state reset, then nIter rounds of
{ PICNN energy+gradient ; dual step } (+ ONE RCCL gather of y* to rank 0 when N > 1).

Workload = BASELINE.json's north_star / metric: global batch 4096, n = 159, K = nIter = 10.
  --scaling strong (default): the SAME 4096-sample batch at every N, split into contiguous shards of 4096/N;
      the x-only context is computed on the full batch (BatchNorm statistics) before it is sliced; every rank
      solves its shard with no data-path collective; rank 0 gathers y*.  `value` = 4096 * K / step time.
  --scaling weak: 4096 samples per rank (per-GPU work fixed).  A strong-scaling run at N > 1 also times that point
      (`extra.weak`, --weak-steps), so that one run shows both curves.
`extra.c4` times BASELINE.json configs[3] the same way (Bibsonomy batch 4096 sharded N ways, nIter = 30); at N = 1 `extra.c3`
times configs[2] (completion conv PICNN, batch 256) at nIter = 5 and 30 and `extra.c5` configs[4] (RL critic, batch 8192).
Inputs (context, weights, y0) are resident in HBM before the timed region.

Rank 0 prints ONE JSON line; see DESIGN.md "Measurement" for the definitions of `roofline` and `cpu_baseline`.
The control flow (sharding, timing, max over ranks, JSON) is `run()`; the HIP workload is injected into it so
that tests/test_bench_flow.py can drive the same code on CPU with gloo and a stub solver.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from icnn_amd import dist as be_dist  # noqa: E402

PEAK_FP32_TFLOPS = 157.3      # MI355X_MICROARCH.md: dense f32 MFMA = f32 vector peak
PEAK_HBM_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E spec
TRAFFIC_FILE = os.path.join(REPO, "profiles", "traffic.json")   # written by tools/prof_round.sh from the PMC passes


def measured_traffic(kernel, batch, n_iter):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/traffic.json: (2*FETCH_SIZE +
    WRITE_SIZE)*1024 with the guide's gfx950 FETCH_SIZE correction), or None if that shape was not profiled."""
    try:
        table = json.load(open(TRAFFIC_FILE))
    except (OSError, ValueError):
        return None, None
    for row in table.get("kernels", []):
        if row["kernel"] == kernel and row["batch"] == batch and row["n_iter"] == n_iter:
            return row["hbm_bytes_per_launch"], "%s @ %s" % (table.get("source", "profiles/"), table.get("commit", "?"))
    return None, None


# --------------------------------------------------------------------------------------------------------
# the HIP workload (product path)
# --------------------------------------------------------------------------------------------------------
class HipWorkload:
    """Bibsonomy FC-PICNN on this rank's GPU.  `step()` enqueues one fused solve of the local shard."""

    name = "Bibsonomy FC-PICNN 1836->[600,159], fused solveBatch"

    def __init__(self, args, rank, world, local):
        from icnn_amd import bundle_entropy, picnn
        self.be, self.picnn = bundle_entropy, picnn
        self.dev = torch.device("cuda", local)
        torch.cuda.set_device(self.dev)
        self.spec = picnn.bibtex_spec()
        self.n = self.spec.n_labels
        self.params = picnn.init_params(self.spec, 0, args.regime)
        self.model = picnn.FCModel(self.spec, self.params, self.dev)
        self.rank, self.world, self.strong = rank, world, args.scaling == "strong"
        self.context_mode = "single rank"
        if self.strong:      # the same global batch on every rank, every rank owns a contiguous shard of it
            self.global_batch = args.batch
            self.x_full = self._features(1000, args.batch)
            lo, hi = be_dist.shard_bounds(args.batch, world, rank)
            self.ctx = None
            if world > 1:
                # each rank computes the context rows of ITS shard; the u-path BatchNorm statistics of the global batch
                # come from one RCCL all-reduce of 2 x 600 float64 sums (FCModel.context_sharded)
                try:
                    self.ctx = self.model.context_sharded(self.x_full[lo:hi].contiguous(), batch_total=float(args.batch))
                    self.context_mode = "sharded: shard rows only, BatchNorm sums all-reduced (2 x 600 float64)"
                except RuntimeError as e:      # the same failure on every rank (a collective): all of them fall back
                    print("rank %d: sharded context failed (%s); using the full-batch producer" % (rank, str(e)[:160]),
                          file=sys.stderr)
            if self.ctx is None:
                self.ctx = self.model.context(self.x_full)[lo:hi].contiguous()
                if world > 1:
                    self.context_mode = "replicated: full-batch context on every rank, then sliced"
        else:                # weak: every rank owns its own batch
            self.global_batch = args.batch * world
            self.x_full = self._features(1000 + rank, args.batch)
            self.ctx = self.model.context(self.x_full)
        self.local_batch = self.ctx.shape[0]
        self.solvers = {}

    def _features(self, seed, B):
        rng = np.random.RandomState(seed)
        return torch.from_numpy((rng.rand(B, self.spec.n_features) < 0.04).astype(np.float32)).to(self.dev)

    def solver(self, n_iter):
        if n_iter not in self.solvers:
            self.solvers[n_iter] = self.be.FusedSolver(self.model, self.local_batch, n_iter, "dual", self.dev)
        return self.solvers[n_iter]

    def step(self, n_iter, events=None):
        """-> (result, y_local).  events: optional (start, end) HIP events recorded around the solve's launches on
        the stream they are issued on."""
        s = self.solver(n_iter)
        if events is not None:
            events[0].record()
        res = s.solve(self.ctx, 0.5)
        if events is not None:
            events[1].record()
        return res, res.y

    def step_from_features(self, n_iter):
        """One solve INCLUDING the x-only context producer (be_context.hip) on the full batch: what a training step
        pays per minibatch (the reference's fg recomputes that part on every bundle iteration)."""
        lo, hi = (be_dist.shard_bounds(self.global_batch, self.world, self.rank) if self.strong
                  else (0, self.local_batch))
        ctx = self.model.context(self.x_full)
        ctx = ctx[lo:hi].contiguous() if (lo, hi) != (0, ctx.shape[0]) else ctx
        return self.solver(n_iter).solve(ctx, 0.5)

    def new_events(self):
        return torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def sync(self):
        torch.cuda.synchronize()

    def kernel_for(self, n_iter):
        """Name of the kernel icnn_be_solve_fc dispatches for this rank's shard (be_api.hip)."""
        cus = torch.cuda.get_device_properties(self.dev).multi_processor_count
        B = self.local_batch
        tiles = (B + 15) // 16
        if (B + cus - 1) // cus <= 4:
            return "fused_rows_solve_kernel"
        if 4 * tiles >= cus and (tiles <= 2 * cus or n_iter > 15):
            return "fused_fc_solve_kernel"
        return "fc_fg_kernel + dual_step_kernel"

    # ---- rank-0 extras at N = 1 -------------------------------------------------------------------
    def roofline(self, n_iter, launch_ms, res):
        spec, B, n = self.spec, self.local_batch, self.n
        kbar = float(res.count[:B].float().mean().item())
        flops = n_iter * B * 4.0 * spec.y_path_params          # fwd + bwd of the y-path, 2 flop per MAC
        # per round: context row, y (f64), dE/dy, E per sample + the weights once per launch (fc_fg phase);
        # g, y, cut row, its point, y, h, lam and the older active rows (dual phase) -- DESIGN.md section 4
        bytes_fg = B * (4.0 * spec.ctx_width + 8.0 * n + 4.0 * n + 4.0) + 4.0 * spec.y_path_params
        bytes_dual = B * (4.0 * n + 8.0 * n + 4.0 * n + 8.0 * n + 8.0 * n + 16.0 + max(kbar - 1.0, 0.0) * 4.0 * n)
        by = n_iter * (bytes_fg + bytes_dual)
        kernel = self.kernel_for(n_iter)
        tf = flops / (launch_ms * 1e-3) / 1e12
        traffic, src = measured_traffic(kernel.split(" ")[0], B, n_iter)
        return {
            "kernel": "%s (the whole solve: %d rounds of {PICNN energy+gradient ; dual step})" % (kernel, n_iter),
            "bound": "mfma", "achieved": tf, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": tf / PEAK_FP32_TFLOPS,
            "traffic": traffic, "traffic_source": src, "avg_launch_ms": launch_ms,
            "timing": "HIP events around every solve of the timed loop, on the launch stream (includes the 2 us state reset)",
            "algorithmic_flops_per_launch": flops, "algorithmic_bytes_per_launch": by,
            "hbm_achieved_GBps": by / (launch_ms * 1e-3) / 1e9, "hbm_frac": by / (launch_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
        }

    def solve_stats(self, res, n_iter):
        """+ EXECUTED inner-solves (SURVEY.md 8(d) asks for both): a sample that leaves the loop at outer iteration t (rank test,
        dual :155-161; the state then holds nIters = t - 1) was evaluated and stepped t + 1 times, every other sample nIter
        times.  `value` and `roofline.frac` of the bench line are NOMINAL (batch x nIter, BASELINE.json's definition)."""
        B = self.local_batch
        nact, its = res.count[:B].float(), res.n_iters[:B].float()
        early = its < n_iter
        executed = torch.where(early, torch.clamp(its + 2.0, 0.0, float(n_iter)), torch.full_like(its, float(n_iter)))
        return {"mean_active_cuts": float(nact.mean().item()), "max_active_cuts": int(nact.max().item()),
                "frac_finished_early": float(early.float().mean().item()),
                "mean_newton_updates_per_sample": float(res.newton_iters[:B].float().mean().item()),
                "executed_inner_solves": float(executed.sum().item()), "nominal_inner_solves": float(B * n_iter),
                "executed_over_nominal": float(executed.sum().item()) / float(B * n_iter)}

    def cpu_baseline(self, n_iter, sample, repeats=3):
        """The oracle (NumPy restatement of the reference solver + PICNN; pinned to the reference's own outputs at 1e-12,
        tests/test_oracle_golden.py) timed on the host cores on a bounded slice of the same workload: median of
        `repeats` runs with the BLAS pool limited to ONE thread (the reference's solver loop is single-threaded Python;
        only the PICNN fg inside it has BLAS calls) and with the default pool -- BASELINE.md section 3."""
        from threadpoolctl import threadpool_info, threadpool_limits
        from oracle import bundle_entropy_oracle as oracle
        from oracle import picnn_oracle
        spec, params = self.spec, self.params
        S = min(sample, self.local_batch)
        ctx_rows = self.ctx[:S].cpu().numpy()
        fg = picnn_oracle.make_fg_from_context(params, ctx_rows, list(spec.szs), spec.alpha)

        def timed():
            walls = []
            for _ in range(repeats):
                y0 = np.full((S, spec.n_labels), 0.5)
                t0 = time.perf_counter()
                with np.errstate(all="ignore"):
                    oracle.solve_batch(fg, y0, n_iter)
                walls.append(time.perf_counter() - t0)
            return float(np.median(walls)), walls

        blas = max([i.get("num_threads", 1) for i in threadpool_info() if i.get("user_api") == "blas"] or [1])
        with threadpool_limits(limits=1):
            one, walls_one = timed()
        many, walls_many = timed()
        return {
            "value": S * n_iter / many, "unit": "inner-solves/s", "cores": int(blas), "kind": "port",
            "kind_detail": "NumPy restatement of lib/bundle_entropy_dual.py (oracle/bundle_entropy_oracle.py) pinned to the reference's "
                           "own outputs at 1e-12 (tests/test_oracle_golden.py; fixtures regenerate bit for bit from /root/reference), "
                           "driving the NumPy float32 PICNN restatement (oracle/picnn_oracle.py); the reference file itself does not "
                           "exist on the GPU box and its TensorFlow r0.10 fg cannot run anywhere in this image",
            "sample": "first %d samples of the benchmark batch, nIter=%d, median of %d runs (%.2f s each); the solver loop is "
                      "single-threaded NumPy like the reference, the PICNN fg inside it runs on %d BLAS threads"
                      % (S, n_iter, repeats, many, blas),
            "single_thread": {"value": S * n_iter / one, "cores": 1, "median_wall_s": one,
                              "what": "the same with the BLAS pool limited to one thread (OMP_NUM_THREADS=1)"},
            "median_wall_s": many, "walls_s": {"one_thread": walls_one, "default": walls_many},
            "host_cpus": os.cpu_count(),
        }

    def parity(self, n_iter, y_gpu, sample):
        """max|y* - y*_ref| of the first `sample` samples against the oracle fed by the order-matched PICNN (identical cuts on
        both sides) and by the sgemm-order PICNN (another float32 summation order: the reference algorithm's own band)."""
        from oracle import bundle_entropy_oracle as oracle
        from oracle import picnn_oracle
        spec, params = self.spec, self.params
        S = min(sample, self.local_batch)
        ctx_rows = self.ctx[:S].cpu().numpy()
        out = {"samples": int(S)}
        for key, make, note in (
                ("vs_oracle_mfma_order_fp32", picnn_oracle.make_fg_chain,
                 "oracle PICNN accumulates float32 in the kernel's order (oracle/picnn_chain.c): identical cuts on both sides"),
                ("vs_oracle_sgemm_order_fp32", picnn_oracle.make_fg_from_context,
                 "different float32 summation order in the PICNN; the tail is the reference algorithm's own sensitivity "
                 "(tests/test_sensitivity.py)")):
            fg = make(params, ctx_rows, list(spec.szs), spec.alpha)
            with np.errstate(all="ignore"):
                ref = oracle.solve_batch(fg, np.full((S, spec.n_labels), 0.5), n_iter)
            dy = np.max(np.abs(ref.y - y_gpu[:S]), axis=1)
            out[key] = {"max_abs_dy": float(dy.max()), "median_abs_dy": float(np.median(dy)),
                        "frac_above_1e-5": float((dy > 1e-5).mean()), "note": note}
        return out

    def completion_extra(self, steps):
        """BASELINE.json configs[2] on this GPU: the completion conv PICNN (n = 2048, batch 256) at nIter = 5 and at the
        reference default of 30 (completion/icnn_ebundle.py:41), context resident, timed like the headline (HIP events
        around every solve of the timed loop)."""
        from icnn_amd import bundle_entropy, picnn
        spec = picnn.ConvSpec()
        B = 256
        params = picnn.init_conv_params(spec, 0, "spread")
        x = np.random.RandomState(5).rand(B, spec.H, spec.W, 1).astype(np.float32)[:, :, ::-1, :].copy()
        model = picnn.ConvModel(spec, params)
        ctx = model.context(torch.from_numpy(x))
        y0 = torch.from_numpy(np.repeat((0.2 + 0.6 * np.random.RandomState(9).rand(spec.n_labels))[None], B, axis=0)).to(ctx.device)
        out = {"workload": "BASELINE.json configs[2]: completion conv PICNN 32x64 -> n = 2048, batch 256, one GPU; synthetic image "
                           "halves, random-init weights ('spread')", "steps": steps}
        for n_iter in (5, 30):
            fs = bundle_entropy.FusedSolver(model, B, n_iter, "dual")
            for _ in range(2):
                res = fs.solve(ctx, y0)
            torch.cuda.synchronize()
            evs = []
            for _ in range(steps):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                res = fs.solve(ctx, y0)
                b.record()
                evs.append((a, b))
            torch.cuda.synchronize()
            ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
            out["n_iter_%d" % n_iter] = {"ms_per_solve": ms, "inner_solves_per_s": B * n_iter / (1e-3 * ms),
                                         "max_active_cuts": int(res.count[:B].max().item()),
                                         "mean_newton_updates_per_sample": float(res.newton_iters[:B].float().mean().item())}
        return out

    def rl_extra(self, steps):
        """BASELINE.json configs[4] on this GPU: the RL critic's bundle solve (HalfCheetah, n = 6 actions, batch 8192, nIter = 5,
        variant rl: RL/src/bundle_entropy.py), context resident."""
        from icnn_amd import bundle_entropy, picnn
        spec = picnn.halfcheetah_spec()
        B, n_iter = 8192, 5
        params = picnn.init_params(spec, 0, "spread", yu_bias=1.0, gate_bias=1.0)
        x = np.random.RandomState(7).randn(B, spec.n_features).astype(np.float32)
        model = picnn.FCModel(spec, params)
        ctx = model.context(torch.from_numpy(x))
        fs = bundle_entropy.FusedSolver(model, B, n_iter, "rl")
        for _ in range(2):
            res = fs.solve(ctx, 0.5)
        torch.cuda.synchronize()
        evs = []
        for _ in range(steps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            res = fs.solve(ctx, 0.5)
            b.record()
            evs.append((a, b))
        torch.cuda.synchronize()
        ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
        return {"workload": "BASELINE.json configs[4]: RL critic 17 obs -> 6 actions, 200-200 PICNN, batch 8192, nIter=5, variant rl, "
                            "one GPU; synthetic observations, random-init weights", "steps": steps, "ms_per_solve": ms,
                "inner_solves_per_s": B * n_iter / (1e-3 * ms), "mean_active_cuts": float(res.count[:B].float().mean().item())}

    @staticmethod
    def _time_solver(fs, ctx, y0, steps, warm=2):
        """mean HIP-event milliseconds of `steps` solves (events on the launch stream), last result"""
        for _ in range(warm):
            res = fs.solve(ctx, y0)
        torch.cuda.synchronize()
        evs = []
        for _ in range(steps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            res = fs.solve(ctx, y0)
            b.record()
            evs.append((a, b))
        torch.cuda.synchronize()
        return float(np.mean([a.elapsed_time(b) for a, b in evs])), res

    def shards_extra(self, steps):
        """PROJECTION of the strong-scaling curve from ONE GPU: the solve time of the contiguous shard a rank owns when the
        north-star batch (4096, nIter 10) and configs[3] (nIter 30) are split over 2 / 4 / 8 / 32 GPUs -- context rows of the
        full batch (global BatchNorm statistics), first shard.  What N GPUs can reach at best is 4096 x nIter / shard time,
        before the gather; the measured curve is the driver's SCALE file."""
        out = {"what": "one-GPU solve time of the first contiguous shard of the benchmark batch: a projection of --gpus N "
                       "(value_if_all_ranks_alike = 4096 x nIter / shard ms), not a multi-GPU measurement", "rows": []}
        for n_iter in (10, 30):
            for shard in (2048, 1024, 512, 128):
                if shard >= self.local_batch:
                    continue
                ctx = self.ctx[:shard].contiguous()
                fs = self.be.FusedSolver(self.model, shard, n_iter, "dual", self.dev)
                ms, _ = self._time_solver(fs, ctx, 0.5, steps if n_iter == 10 else max(2, steps // 2))
                out["rows"].append({"shard": shard, "gpus": self.local_batch // shard, "n_iter": n_iter, "ms_per_solve": ms,
                                    "value_if_all_ranks_alike": self.local_batch * n_iter / (1e-3 * ms)})
        return out

    def pdipm_extra(self, steps):
        """variant 'pdipm' (lib/bundle_entropy.py solver='pc', the module the reference's icnn_ebundle.py scripts import and
        dropin/bundle_entropy.py selects) next to variant 'dual' on the same inputs: BASELINE configs[1] (Bibsonomy, batch
        128, nIter 10), the benchmark batch (4096 x 10) and configs[2] (completion conv PICNN, batch 256, nIter 5)."""
        from icnn_amd import bundle_entropy, picnn
        out = {"what": "fused solve, variant pdipm (Mehrotra predictor-corrector per sample, be_ipm_dev.h) vs variant dual on the "
                       "same inputs", "rows": []}
        for name, B, n_iter in (("configs[1] Bibsonomy 128 x 10", 128, 10), ("headline Bibsonomy 4096 x 10", self.local_batch, 10)):
            ctx = self.ctx[:B].contiguous()
            row = {"shape": name, "batch": B, "n_iter": n_iter}
            for variant in ("dual", "pdipm"):
                fs = bundle_entropy.FusedSolver(self.model, B, n_iter, variant, self.dev)
                ms, res = self._time_solver(fs, ctx, 0.5, steps)
                row[variant + "_ms"] = ms
                row[variant + "_y"] = res.y.clone()
            row["max_abs_dy_pdipm_vs_dual"] = float((row.pop("dual_y") - row.pop("pdipm_y")).abs().max().item())
            row["pdipm_over_dual"] = row["pdipm_ms"] / row["dual_ms"]
            out["rows"].append(row)
        spec = picnn.ConvSpec()
        B, n_iter = 256, 5
        params = picnn.init_conv_params(spec, 0, "spread")
        x = np.random.RandomState(5).rand(B, spec.H, spec.W, 1).astype(np.float32)[:, :, ::-1, :].copy()
        model = picnn.ConvModel(spec, params)
        ctx = model.context(torch.from_numpy(x))
        y0 = torch.from_numpy(np.repeat((0.2 + 0.6 * np.random.RandomState(9).rand(spec.n_labels))[None], B, axis=0)).to(ctx.device)
        row = {"shape": "configs[2] completion conv 256 x 5", "batch": B, "n_iter": n_iter}
        for variant in ("dual", "pdipm"):
            fs = bundle_entropy.FusedSolver(model, B, n_iter, variant)
            ms, res = self._time_solver(fs, ctx, y0, steps)
            row[variant + "_ms"] = ms
            row[variant + "_y"] = res.y.clone()
        row["max_abs_dy_pdipm_vs_dual"] = float((row.pop("dual_y") - row.pop("pdipm_y")).abs().max().item())
        row["pdipm_over_dual"] = row["pdipm_ms"] / row["dual_ms"]
        out["rows"].append(row)
        return out

    def generic_fg_extra(self, steps):
        """The ZERO-CHANGE drop-in: solveBatch(fg, initXs, nIter) with an opaque `fg` (what an unmodified icnn_ebundle.py calls,
        multi-label-cls/icnn_ebundle.py:218-226) -- here a device callable, so that the time is the library's: per outer
        iteration one icnn_be_dual_step launch (cut, rank test, projected Newton, y update for the whole batch) between two
        calls of fg.  dual_step_ms_per_round is the mean over the rounds of HIP events around that launch alone."""
        from icnn_amd import bundle_entropy
        out = {"what": "generic mode: icnn_be_dual_step per round between calls of an opaque device fg (here FCModel.fg); "
                       "events around the dual-step launch alone, and around the whole 10-iteration loop", "rows": []}
        for B in (128, self.local_batch):
            ctx = self.ctx[:B].contiguous()
            n_iter = 10
            y = torch.empty(B, self.n, dtype=torch.float64, device=self.dev)
            state = bundle_entropy.BundleState(y, n_iter, "dual", torch.float32, 0)
            dual_ms, loop_ms = [], []
            for rep in range(steps + 2):
                y.fill_(0.5)
                state.init()
                evs = []
                a0, b0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a0.record()
                for t in range(n_iter):
                    f_t, g_t = self.model.fg(ctx, y)
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    state.step(t, f_t, g_t)
                    b.record()
                    evs.append((a, b))
                b0.record()
                torch.cuda.synchronize()
                if rep >= 2:
                    dual_ms.append(float(np.mean([a.elapsed_time(b) for a, b in evs])))
                    loop_ms.append(a0.elapsed_time(b0))
            out["rows"].append({"batch": B, "n_iter": n_iter, "dual_step_ms_per_round": float(np.mean(dual_ms)),
                                "loop_ms_with_device_fg": float(np.mean(loop_ms)),
                                "inner_solves_per_s": B * n_iter / (1e-3 * float(np.mean(loop_ms)))})
        return out

    def compat_cost(self, n_iter, res):
        """What an UNMODIFIED icnn_ebundle.py pays on top of the solve: BundleResult.as_reference_tuple builds the
        reference's 6-tuple (NumPy y, ragged Python lists of the active cuts, their offsets, points and multipliers --
        lib/bundle_entropy_dual.py:179) from the device state: one packed device-to-host copy of the active rows of G / ys /
        h / lam into pinned memory; the O(B K) Python objects of each of the four ragged lists are built in one bulk pass
        when that list is first used (SURVEY.md hard part 6).  `ms` = the call itself; `ms_all_lists_built` = the call plus
        the bulk pass of all four lists (what train_step_fd, multi-label-cls/icnn_ebundle.py:296-307, ends up paying)."""
        res.as_reference_tuple()                  # untimed: the first call pins the host block and loads torch's kernels
        self.sync()
        walls, walls_built, cuts = [], [], 0
        for _ in range(5):
            t0 = time.perf_counter()
            tup = res.as_reference_tuple()
            t1 = time.perf_counter()
            for c in tup[1:5]:
                c.materialize()
            walls.append(t1 - t0)
            walls_built.append(time.perf_counter() - t0)
            cuts = sum(len(a) for a in tup[1])
            del tup, c                            # the pinned block goes back to torch's host allocator for the next call
        return {"what": "BundleResult.as_reference_tuple() after a solve (native mode skips it): icnn_be_export_active packs the "
                        "%d active rows on the device, ONE device -> pinned-host copy, ragged lists built in bulk on first use; "
                        "median of 5" % cuts,
                "ms": 1e3 * float(np.median(walls)), "ms_all_lists_built": 1e3 * float(np.median(walls_built)),
                "batch": self.local_batch, "n_iter": n_iter}


# --------------------------------------------------------------------------------------------------------
# control flow shared by the product run and the CPU dry run
# --------------------------------------------------------------------------------------------------------
def timed_steps(wl, n_iter, steps, warmup, rank, world, gather_dst, with_events):
    """W untimed + K timed steps bracketed by barrier + device sync; returns (max-over-ranks seconds, per-rank
    seconds, mean per-solve milliseconds from the events of THIS rank, last result, last gathered y)."""
    global_batch = wl.global_batch

    def mark(ev):
        if hasattr(ev, "record"):
            ev.record()
        else:
            ev.t = time.perf_counter()

    def one(events=None, gather_events=None):
        res, y_local = wl.step(n_iter, events)
        y_all = y_local
        if world > 1:
            if gather_events is not None:
                mark(gather_events[0])
            y_all = be_dist.gather_rows(y_local, global_batch, world, rank, dst=gather_dst[0])
            if gather_events is not None:
                mark(gather_events[1])
        return res, y_all

    def fence():
        if world > 1:
            torch.distributed.barrier()
        wl.sync()

    for _ in range(warmup):
        one()
    events = [wl.new_events() for _ in range(steps)] if with_events else [None] * steps
    gevents = [wl.new_events() for _ in range(steps)] if with_events and world > 1 else [None] * steps
    fence()
    t0 = time.perf_counter()
    for k in range(steps):
        res, y_all = one(events[k], gevents[k])
    fence()
    mine = time.perf_counter() - t0
    launch_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in events])) if with_events else None
    gather_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in gevents])) if gevents[0] is not None else None
    per_rank, per_rank_solve, per_rank_gather = [mine], [launch_ms], [gather_ms]
    if world > 1:      # every rank's wall clock, solve time and gather time (from its own events), in one all-gather
        t = torch.tensor([mine, launch_ms or 0.0, gather_ms or 0.0], dtype=torch.float64,
                         device=y_all.device if y_all is not None else res.y.device)
        out = [torch.zeros_like(t) for _ in range(world)]
        torch.distributed.all_gather(out, t)
        per_rank = [float(v[0].item()) for v in out]
        per_rank_solve = [float(v[1].item()) for v in out]
        per_rank_gather = [float(v[2].item()) for v in out]
    TIMING_DETAIL["solve_ms"], TIMING_DETAIL["gather_ms"] = per_rank_solve, per_rank_gather
    return max(per_rank), per_rank, launch_ms, res, y_all


TIMING_DETAIL = {}      # per-rank solve / gather milliseconds of the last timed_steps call (events on the launch stream)


def run(args, workload_factory=None, backend=None):
    if workload_factory is None:
        workload_factory = load_workload(getattr(args, "workload", "hip"))
    if backend is None:
        backend = getattr(args, "backend", None)
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:       # before the rendezvous: a wrong launch must not hang in it
        world = int(os.environ.get("WORLD_SIZE", "1"))
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (run `python bench.py --gpus %d` from a plain shell, or "
                         "torch.distributed.run with --nproc-per-node %d)" % (args.gpus, world, args.gpus, args.gpus))
    rank, world, local = be_dist.init_from_env(backend)
    if getattr(args, "one_device", False):
        local = 0
    wl = workload_factory(args, rank, world, local)
    n_iter = args.n_iter
    gather_dst = [0]      # rank 0 collects y*; [None] = all-gather (fallback if the backend refuses gather)
    if world > 1:
        try:
            _, y_local = wl.step(n_iter)
            be_dist.gather_rows(y_local, wl.global_batch, world, rank, dst=0)
            wl.sync()
        except RuntimeError as e:          # same exception on every rank: all of them switch
            if rank == 0:
                print("gather to rank 0 not available (%s); using all_gather" % str(e)[:120], file=sys.stderr)
            gather_dst[0] = None

    elapsed, per_rank, launch_ms, res, y_all = timed_steps(wl, n_iter, args.steps, args.warmup, rank, world,
                                                            gather_dst, with_events=True)
    ms_per_step = 1e3 * elapsed / args.steps
    value = wl.global_batch * n_iter * args.steps / elapsed
    if hasattr(res, "raise_on_error"):
        res.raise_on_error()
    if rank == 0 and world > 1:
        assert y_all is not None and y_all.shape[0] == wl.global_batch

    out = {
        "metric": "inner-solves/sec (batch x iters), n=159 K=10", "value": value, "unit": "inner-solves/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32 PICNN / f64 dual solve",
        "data": "synthetic (random-init weights, '%s' regime; x ~ Bernoulli(0.04))" % args.regime,
        "config": {"workload": "%s, n=%d, nIter=K=%d, global batch %d = %d per GPU x %d"
                               % (wl.name, wl.n, n_iter, wl.global_batch, wl.local_batch, world),
                   "variant": "dual (lib/bundle_entropy_dual.py)", "global_batch": wl.global_batch,
                   "per_gpu_batch": wl.local_batch,
                   "parallelism": "contiguous batch shards x%d, no data-path collective, one RCCL %s of y*"
                                  % (world, "gather to rank 0" if gather_dst[0] == 0 else "all-gather")},
        "world_size": torch.distributed.get_world_size() if world > 1 else 1,
        "per_rank_ms_per_step": [1e3 * t / args.steps for t in per_rank],
        "rank0_solve_ms": launch_ms,
        # where a step goes on every rank: the fused solve of its shard, and the one collective (rank 0's gather ends when the
        # slowest shard has arrived; the other ranks only post their send)
        "per_rank_solve_ms": TIMING_DETAIL.get("solve_ms"), "per_rank_gather_ms": TIMING_DETAIL.get("gather_ms"),
    }
    if hasattr(wl, "context_mode"):
        out["config"]["context"] = wl.context_mode

    # BASELINE.json configs[3]: the same batch, nIter = 30 (not the headline; timed the same way, fewer steps)
    if args.c4_steps > 0:
        c4_elapsed, c4_rank, c4_ms, c4_res, _ = timed_steps(wl, 30, args.c4_steps, 1, rank, world, gather_dst,
                                                            with_events=True)
        out["extra"] = {"c4": {
            "workload": "BASELINE.json configs[3]: Bibsonomy PICNN, global batch %d = %d per GPU x %d, nIter=30"
                        % (wl.global_batch, wl.local_batch, world),
            "steps": args.c4_steps, "ms_per_step": 1e3 * c4_elapsed / args.c4_steps,
            "value": wl.global_batch * 30 * args.c4_steps / c4_elapsed, "unit": "inner-solves/s",
            "per_rank_ms_per_step": [1e3 * t / args.c4_steps for t in c4_rank], "rank0_solve_ms": c4_ms,
            "per_rank_solve_ms": TIMING_DETAIL.get("solve_ms"), "per_rank_gather_ms": TIMING_DETAIL.get("gather_ms"),
            "kernel": wl.kernel_for(30) if hasattr(wl, "kernel_for") else None}}
        if rank == 0 and hasattr(wl, "roofline"):
            out["extra"]["c4"]["roofline"] = wl.roofline(30, c4_ms, c4_res)
            out["extra"]["c4"]["solve_stats"] = wl.solve_stats(c4_res, 30)
            ex = out["extra"]["c4"]["solve_stats"]["executed_over_nominal"]
            out["extra"]["c4"]["roofline"]["frac_executed"] = out["extra"]["c4"]["roofline"]["frac"] * ex
            out["extra"]["c4"]["value_executed"] = out["extra"]["c4"]["value"] * ex

    # N > 1, strong scaling (the default: BASELINE's batch split N ways): ALSO the weak-scaling point -- the headline batch per
    # rank, no data-path collective, the same gather -- so that one run of `--gpus N` shows both curves.  A second workload
    # object (its own context rows and solvers); timed like the headline.
    if world > 1 and args.scaling == "strong" and args.weak_steps > 0:
        import copy
        wargs = copy.copy(args)
        wargs.scaling = "weak"
        wl_w = workload_factory(wargs, rank, world, local)
        w_elapsed, w_rank, w_ms, _, _ = timed_steps(wl_w, n_iter, args.weak_steps, 1, rank, world, gather_dst, with_events=True)
        out.setdefault("extra", {})["weak"] = {
            "what": "weak-scaling point of the same run: %d samples PER RANK (global batch %d), nIter=%d, the same single gather of y*"
                    % (wl_w.local_batch, wl_w.global_batch, n_iter),
            "scaling": "weak", "steps": args.weak_steps, "ms_per_step": 1e3 * w_elapsed / args.weak_steps,
            "value": wl_w.global_batch * n_iter * args.weak_steps / w_elapsed, "unit": "inner-solves/s",
            "per_rank_ms_per_step": [1e3 * t / args.weak_steps for t in w_rank],
            "per_rank_solve_ms": TIMING_DETAIL.get("solve_ms"), "per_rank_gather_ms": TIMING_DETAIL.get("gather_ms")}
        del wl_w

    if rank == 0 and world == 1 and hasattr(wl, "step_from_features"):
        for _ in range(2):
            wl.step_from_features(n_iter)
        wl.sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            wl.step_from_features(n_iter)
        wl.sync()
        e2e = (time.perf_counter() - t0) / args.steps
        out.setdefault("extra", {})["from_features"] = {
            "what": "x [B, 1836] -> context (be_context.hip: 3 MFMA GEMMs + BatchNorm) -> fused solve, per minibatch",
            "ms_per_step": 1e3 * e2e, "value": wl.global_batch * n_iter / e2e, "unit": "inner-solves/s"}

    if rank == 0 and world == 1 and args.c3_steps > 0 and hasattr(wl, "completion_extra"):
        out.setdefault("extra", {})["c3"] = wl.completion_extra(args.c3_steps)
        out["extra"]["c5"] = wl.rl_extra(args.c3_steps)
        out["extra"]["shards"] = wl.shards_extra(args.c3_steps)
        out["extra"]["pdipm"] = wl.pdipm_extra(args.c3_steps)
        out["extra"]["generic_fg"] = wl.generic_fg_extra(args.c3_steps)

    if rank == 0:
        if hasattr(wl, "solve_stats"):
            out["solve_stats"] = wl.solve_stats(res, n_iter)
        if hasattr(wl, "roofline"):
            out["roofline"] = wl.roofline(n_iter, launch_ms, res)
            if "solve_stats" in out:
                out["roofline"]["frac_executed"] = out["roofline"]["frac"] * out["solve_stats"]["executed_over_nominal"]
        if world == 1 and args.cpu_sample > 0 and hasattr(wl, "cpu_baseline"):
            y_gpu = res.y.cpu().numpy()
            out.setdefault("extra", {})["compat"] = wl.compat_cost(n_iter, res)
            out["cpu_baseline"] = wl.cpu_baseline(n_iter, args.cpu_sample)
            out["parity"] = wl.parity(n_iter, y_gpu, args.parity_sample)
        print(json.dumps(out))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    return out


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong")
    ap.add_argument("--batch", type=int, default=4096, help="global batch (strong) / samples per GPU (weak)")
    ap.add_argument("--n-iter", type=int, default=10)
    ap.add_argument("--regime", default="spread")
    ap.add_argument("--c4-steps", type=int, default=3, help="timed steps of the nIter=30 configuration (0 = skip)")
    ap.add_argument("--weak-steps", type=int, default=10,
                    help="N > 1 with strong scaling: timed steps of the weak-scaling point (--batch samples per rank) reported as "
                         "extra.weak (0 = skip)")
    ap.add_argument("--c3-steps", type=int, default=5, help="timed solves of the completion and the RL configuration (N = 1 only; 0 = skip)")
    ap.add_argument("--cpu-sample", type=int, default=512, help="samples of the CPU baseline's slice (0 = skip it and the parity leg)")
    ap.add_argument("--parity-sample", type=int, default=1024, help="samples compared with the CPU oracle")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default=None,
                    help="process-group backend for N > 1 (default: nccl = RCCL when a GPU is visible, else gloo)")
    ap.add_argument("--workload", default="hip",
                    help="'hip' (the product path) or FILE.py:CLASS of a stand-in with HipWorkload's interface "
                         "(tests/bench_stub.py:StubWorkload drives the control flow on CPU)")
    ap.add_argument("--one-device", action="store_true",
                    help="every rank uses device 0 (two ranks on a one-GPU box, with --backend gloo: RCCL refuses that)")
    ap.add_argument("--launch-timeout", type=float, default=1500.0, help="seconds before self-launched ranks are killed")
    return ap.parse_args(argv)


def load_workload(spec):
    """'hip' -> HipWorkload; 'FILE.py:CLASS' -> that class, loaded by path (relative to the repo root)."""
    if spec in (None, "hip"):
        return HipWorkload
    import importlib.util
    path, _, cls = spec.partition(":")
    if not os.path.isabs(path):
        path = os.path.join(REPO, path)
    mod_spec = importlib.util.spec_from_file_location("bench_workload_%s" % os.path.basename(path)[:-3], path)
    mod = importlib.util.module_from_spec(mod_spec)
    mod_spec.loader.exec_module(mod)
    return getattr(mod, cls or "Workload")


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args, argv):
    """`python bench.py --gpus N` from a plain shell (no torchrun): start the N ranks here -- one child process per GPU,
    each THIS script with the same arguments and RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set the way
    torch.distributed.run sets them -- and wait.  Rank 0's stdout (the ONE JSON line) is this process's stdout.  If a rank
    fails the others are stopped (their own process groups, by PID) and the exit status is that rank's; a hung job is
    killed after --launch-timeout seconds.  Returns the exit status."""
    import signal
    import subprocess
    n = args.gpus
    if load_workload(args.workload) is HipWorkload and not args.one_device:
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < n:
            print("bench.py: --gpus %d but %d GPU(s) visible" % (n, have), file=sys.stderr)
            return 2
    env = dict(os.environ)
    env.update(WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=e,
                                      stdout=None if r == 0 else subprocess.DEVNULL, start_new_session=True))

    def stop_all():
        for p in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, signal.SIGTERM)          # the rank's own session: nothing else is in it
                except ProcessLookupError:
                    pass
        t_end = time.time() + 10.0
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_end - time.time()))
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(p.pid, signal.SIGKILL)
                except ProcessLookupError:
                    pass

    deadline = time.time() + args.launch_timeout
    status = 0
    try:
        live = set(range(n))
        while live:
            for r in sorted(live):
                rc = procs[r].poll()
                if rc is None:
                    continue
                live.discard(r)
                if rc != 0 and status == 0:
                    status = rc if rc > 0 else 128 - rc
                    print("bench.py: rank %d exited with status %d; stopping the other ranks" % (r, rc), file=sys.stderr)
                    stop_all()
                    live.clear()
                    break
            if live and time.time() > deadline:
                print("bench.py: ranks still running after %.0f s; stopping them" % args.launch_timeout, file=sys.stderr)
                status = 124
                stop_all()
                break
            if live:
                time.sleep(0.05)
    except KeyboardInterrupt:
        stop_all()
        status = 130
    return status


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args, argv)
    run(args)
    return 0


if __name__ == "__main__":
    sys.exit(main())
