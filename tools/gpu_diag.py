#!/usr/bin/env python3
"""Step-by-step diagnosis of the HIP dual step against the CPU oracle (GPU box only).

For each problem the oracle is run to every prefix length t+1 and the GPU state after
outer iteration t is compared with it, so the first diverging iteration / sample /
quantity is reported instead of an end-to-end mismatch.  Output goes to stdout and
gpurun_out/diag.log.  Not part of the product; imports oracle/ as a checker only.
"""
import os
import sys
import time
import traceback

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

import problems  # noqa: E402
from gpu_util import result_to_host  # noqa: E402
from oracle import bundle_entropy_oracle as oracle  # noqa: E402
from oracle import picnn_oracle  # noqa: E402

OUT = os.path.join(REPO, "gpurun_out")
os.makedirs(OUT, exist_ok=True)
LOG = open(os.path.join(OUT, "diag.log"), "w")


def say(*a):
    s = " ".join(str(x) for x in a)
    print(s, flush=True)
    LOG.write(s + "\n")
    LOG.flush()


def stepwise(name, prob, n_iter, variant):
    from icnn_amd import bundle_entropy as be
    y_dev = torch.from_numpy(prob.y0()).cuda()
    state = None
    y_host = prob.y0()
    for t in range(n_iter):
        f_t, g_t = prob.fg(y_host)
        dt = torch.float64 if np.asarray(g_t).dtype == np.float64 else torch.float32
        if state is None:
            state = be.BundleState(y_dev, n_iter, variant, dt)
            state.init()
        state.step(t, torch.as_tensor(f_t).to("cuda", dt).contiguous(),
                   torch.as_tensor(g_t).to("cuda", dt).contiguous())
        torch.cuda.synchronize()
        host = result_to_host(be.BundleResult(state))
        with np.errstate(all="ignore"):
            ora = oracle.solve_batch(prob.fg, prob.y0(), t + 1, variant=variant)
        dy = np.max(np.abs(host["y"] - ora.y), axis=1)
        # the oracle prefix run reports nIters = t+1 for samples still in the loop
        ora_it = [n_iter if v == t + 1 else v for v in ora.n_iters]
        disc = [u for u in range(prob.B)
                if list(host["active"][u]) != list(ora.active[u]) or int(host["n_iters"][u]) != int(ora_it[u])]
        dh = np.abs(host["h"][:, t] - ora.h[:, t]).max()
        say("  %-22s %-4s t=%2d  max|dy|=%.3e  max|dh_t|=%.3e  discrete-diff=%d  status!=0: %d  newton max %d"
            % (name, variant, t, dy.max(), dh, len(disc), int((host["status"] != 0).sum()), int(host["newton"].max())))
        if dy.max() > 1e-5 or disc:
            u = disc[0] if disc else int(np.argmax(dy))
            say("     first bad sample %d: gpu active %s lam %s nIters %s | oracle active %s lam %s nIters %s"
                % (u, host["active"][u], host["lam"][u], host["n_iters"][u], ora.active[u], ora.lam[u], ora.n_iters[u]))
            say("     y gpu %s\n     y ora %s" % (host["y"][u][:6], ora.y[u][:6]))
            return False
        y_host = host["y"].copy()
    return True


def main():
    say("device:", torch.cuda.get_device_name(0))
    from icnn_amd import _lib, picnn
    lib = _lib.load()
    say("abi", lib.icnn_be_abi_version())
    ok = True
    for case in ["single_sample", "zero_gradient", "maxaffine_n159", "lse_n159", "lse_n33", "n_equals_1",
                 "maxaffine_n159_long", "action_box", "c1_quadratic", "maxaffine_f64"]:
        factory, n_iter = problems.GOLDEN_CASES[case]
        for variant in ("dual", "rl"):
            try:
                ok &= stepwise(case, factory(), n_iter, variant)
            except Exception:
                ok = False
                say("EXC in", case, variant, traceback.format_exc())

    # fg kernel
    for which in ("bibtex", "halfcheetah"):
        try:
            spec = picnn.bibtex_spec() if which == "bibtex" else picnn.halfcheetah_spec()
            kw = {} if which == "bibtex" else dict(yu_bias=1.0, gate_bias=1.0)
            params = picnn.init_params(spec, 0, "spread", **kw)
            rng = np.random.RandomState(1)
            B = 100
            x = (rng.rand(B, spec.n_features) < 0.04).astype(np.float32) if which == "bibtex" \
                else rng.randn(B, spec.n_features).astype(np.float32)
            model = picnn.FCModel(spec, params)
            fg = picnn_oracle.make_fg(params, x, list(spec.szs), spec.alpha, spec.batchnorm,
                                      "action" if spec.action_box else None)
            ctx_ref = picnn_oracle.flat_context(fg.ctx)
            ctx = model.context(torch.from_numpy(x)).cpu().numpy()
            say("ctx %s max err %.3e (scale %.3e)" % (which, np.abs(ctx - ctx_ref).max(), np.abs(ctx_ref).max()))
            y = rng.rand(B, spec.n_labels)
            f, g = model.fg(torch.from_numpy(ctx_ref).cuda(), torch.from_numpy(y).cuda())
            torch.cuda.synchronize()
            f_ref, g_ref = fg(y)
            say("fg %s: max|df|=%.3e (|f| %.3e)  max|dg|=%.3e (|g| %.3e)"
                % (which, np.abs(f.cpu().numpy() - f_ref).max(), np.abs(f_ref).max(),
                   np.abs(g.cpu().numpy() - g_ref).max(), np.abs(g_ref).max()))
        except Exception:
            ok = False
            say("EXC in fg", which, traceback.format_exc())
    say("DIAG", "OK" if ok else "FAILED")


if __name__ == "__main__":
    t0 = time.time()
    main()
    say("diag took %.1fs" % (time.time() - t0))
