#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE solver itself.

Runs only in the build container (needs /root/reference, which does not exist on
the GPU box).  The reference modules are loaded by path with bytecode writing
disabled -- nothing is copied out of, or written into, the reference tree.

For every case in tests/problems.py::GOLDEN_CASES and every solver variant

    dual   lib/bundle_entropy_dual.py::solveBatch      (oracle of record, C1-C4)
    rl     RL/src/bundle_entropy.py::solveBatch        (oracle of record, C5)
    pdipm  lib/bundle_entropy.py::solveBatch('pc')     (loose cross-check only)

the ragged 6-tuple is flattened into padded arrays and saved to
tests/golden/<case>__<variant>.npz.  Cut gradients / cut points are stored as
per-row float64 checksums (sum and index-weighted sum) instead of full rows to
keep the fixtures small; `lam`, `b`, `y`, counts and nIters are stored in full.

Usage:  python oracle/gen_golden.py [--ref /root/reference] [--only case]
        python oracle/gen_golden.py --coretype Haswell      (-> tests/golden/<case>__rl@haswell.npz)

`--coretype X` re-runs the RL variant of the reference in a child process with OPENBLAS_CORETYPE=X, i.e. with
NumPy's own OpenBLAS dispatched to another x86 kernel family (the fixtures of record were made with the family
this container's CPU selects, SkylakeX).  Nothing else changes -- same reference code, same NumPy, same inputs --
so the distance between those outputs is the reference's own sensitivity to the rounding of its BLAS/LAPACK
kernels (tests/test_rl_sensitivity.py, tests/test_gpu_parity.py use it as the tolerance band on degenerate bundles).
"""
import argparse
import contextlib
import importlib.util
import io
import os
import subprocess
import sys

if "--coretype" in sys.argv and "ICNN_GOLDEN_CHILD" not in os.environ:      # before NumPy loads its OpenBLAS
    env = dict(os.environ, OPENBLAS_CORETYPE=sys.argv[sys.argv.index("--coretype") + 1], ICNN_GOLDEN_CHILD="1")
    sys.exit(subprocess.call([sys.executable] + sys.argv, env=env))

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(REPO, "tests"))

import problems  # noqa: E402


def load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def row_checksums(rows, n):
    w = np.arange(1, n + 1, dtype=np.float64)
    out = np.zeros((len(rows), 2))
    for i, r in enumerate(rows):
        r = np.asarray(r, dtype=np.float64)
        out[i, 0] = r.sum()
        out[i, 1] = (r * w).sum()
    return out


def flatten(result, B, n, T):
    x, A, b, lam, xs, n_iters = result
    cnt = np.array([len(a) for a in A], dtype=np.int64)
    lam_none = np.array([l is None for l in lam], dtype=bool)
    lam_pad = np.zeros((B, T))
    b_pad = np.zeros((B, T))
    a_chk = np.zeros((B, T, 2))
    ys_chk = np.zeros((B, T, 2))
    for u in range(B):
        k = cnt[u]
        if lam[u] is not None:
            assert len(lam[u]) == k or k == 0, (len(lam[u]), k)
            lam_pad[u, :len(lam[u])] = lam[u]
        b_pad[u, :k] = np.asarray(b[u], dtype=np.float64)
        if k:
            a_chk[u, :k] = row_checksums(A[u], n)
            ys_chk[u, :k] = row_checksums(xs[u], n)
    return dict(y=np.asarray(x, dtype=np.float64), cnt=cnt, lam_none=lam_none,
                lam=lam_pad, b=b_pad, a_chk=a_chk, ys_chk=ys_chk,
                n_iters=np.asarray(n_iters, dtype=np.int64))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--only", default=None)
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    ap.add_argument("--coretype", default=None, help="OpenBLAS kernel family for an RL-variant sensitivity run")
    args = ap.parse_args()
    suffix = ""
    if args.coretype:
        from threadpoolctl import threadpool_info
        arch = [i.get("architecture") for i in threadpool_info() if i.get("internal_api") == "openblas"]
        assert arch and arch[0].lower() == args.coretype.lower(), "OpenBLAS runs %s, not %s" % (arch, args.coretype)
        suffix = "@" + args.coretype.lower()

    ref = {
        "dual": load_by_path("ref_be_dual", os.path.join(args.ref, "lib", "bundle_entropy_dual.py")),
        "rl": load_by_path("ref_be_rl", os.path.join(args.ref, "RL", "src", "bundle_entropy.py")),
        "pdipm": load_by_path("ref_be_pdipm", os.path.join(args.ref, "lib", "bundle_entropy.py")),
    }
    os.makedirs(args.out, exist_ok=True)

    for case, (factory, n_iter) in problems.GOLDEN_CASES.items():
        if args.only and case != args.only:
            continue
        for variant, mod in ref.items():
            if args.coretype and variant != "rl":
                continue
            prob = factory()
            y0 = prob.y0()
            sink = io.StringIO()
            err = ""
            try:
                with contextlib.redirect_stdout(sink), np.errstate(all="ignore"):
                    if variant == "pdipm":
                        res = mod.solveBatch(prob.fg, y0, nIter=n_iter, solver="pc")
                    else:
                        res = mod.solveBatch(prob.fg, y0, nIter=n_iter)
            except Exception as exc:  # the reference raises on singular systems
                err = "%s: %s" % (type(exc).__name__, exc)
                res = None
            path = os.path.join(args.out, "%s__%s%s.npz" % (case, variant, suffix))
            if res is None:
                np.savez_compressed(path, error=np.array(err))
                print("%-24s %-6s raised %s" % (case, variant, err))
                continue
            assert res[0] is y0, "reference must update initXs in place"
            flat = flatten(res, prob.B, prob.n, n_iter)
            np.savez_compressed(path, error=np.array(""), n_iter=np.array(n_iter), **flat)
            print("%-24s %-6s sum(y)=%.15g  cnt[min,max]=%d,%d  nIters[min,max]=%d,%d"
                  % (case, variant, flat["y"].sum(), flat["cnt"].min(), flat["cnt"].max(),
                     flat["n_iters"].min(), flat["n_iters"].max()))


def single_sample_fixture(ref_dual, out_dir):
    """lib/bundle_entropy_dual.py::solve (:87-127), the single-sample form: run on a few samples of the golden problems, one
    sample at a time (fg of a 1-D point -> scalar energy, 1-D gradient).  -> tests/golden/solve__dual.npz.  (The `solve` of
    lib/bundle_entropy.py, :168-190, calls `pdipm(G, h)` -- three undefined names -- and raises NameError in the reference.)"""
    picks = [("c1_quadratic", [0, 7, 31]), ("lse_n159", [0, 5]), ("maxaffine_n159", [3]), ("lse_n33", [2, 11])]
    out = {}
    for case, samples in picks:
        factory, n_iter = problems.GOLDEN_CASES[case]
        prob = factory()
        y0 = prob.y0()
        ys = []
        for u in samples:
            def fg1(x, u=u):
                Y = np.array(y0, copy=True)
                Y[u] = x
                f, g = prob.fg(Y)
                return f[u], g[u]
            with contextlib.redirect_stdout(io.StringIO()), np.errstate(all="ignore"):
                x = ref_dual.solve(fg1, np.array(y0[u], copy=True), nIter=n_iter)
            ys.append(np.asarray(x, dtype=np.float64))
        out[case + "__samples"] = np.array(samples)
        out[case + "__y"] = np.stack(ys)
        print("solve %-18s samples %s  sum(y)=%.15g" % (case, samples, float(np.sum(out[case + "__y"]))))
    np.savez_compressed(os.path.join(out_dir, "solve__dual.npz"), **out)


if __name__ == "__main__":
    main()
    if "--coretype" not in sys.argv and "--only" not in sys.argv:
        _args = [a for a in sys.argv[1:]]
        _ref = _args[_args.index("--ref") + 1] if "--ref" in _args else "/root/reference"
        _out = _args[_args.index("--out") + 1] if "--out" in _args else os.path.join(REPO, "tests", "golden")
        single_sample_fixture(load_by_path("ref_be_dual1", os.path.join(_ref, "lib", "bundle_entropy_dual.py")), _out)
