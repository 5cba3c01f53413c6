#!/usr/bin/env python3
"""Golden vectors for the implicit-differentiation feed from the REFERENCE's own functions.

multi-label-cls/icnn_ebundle.py and completion/icnn_ebundle.py import TensorFlow at module level, so
they cannot be imported here; `crossEntrGrad` (:390-417) and `mseGrad` (:493-522) are pure NumPy,
so their `def` nodes are lifted out of the source with `ast` and executed unmodified in a NumPy
namespace (nothing is copied into this repository).  The reference solver (lib/bundle_entropy_dual.py,
loaded by path) provides (yN, G, ys, lam) on the seeded problems of tests/problems.py; labels are
seeded Bernoulli / uniform draws.  Output: tests/golden/feed__<case>__<loss>.npz.
"""
import ast
import contextlib
import importlib.util
import io
import os
import sys

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(REPO, "tests"))
import problems  # noqa: E402

REF = "/root/reference"
CASES = {"maxaffine_n159": 10, "lse_n33": 12, "zero_gradient": 6}


def lift(path, name):
    tree = ast.parse(open(path).read())
    node = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == name)
    ns = {"np": np, "sys": sys}
    exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), ns)
    return ns[name]


def main():
    xent = lift(os.path.join(REF, "multi-label-cls", "icnn_ebundle.py"), "crossEntrGrad")
    mse = lift(os.path.join(REF, "completion", "icnn_ebundle.py"), "mseGrad")
    spec = importlib.util.spec_from_file_location("ref_dual", os.path.join(REF, "lib", "bundle_entropy_dual.py"))
    dual = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(dual)
    for case, n_iter in CASES.items():
        prob = problems.GOLDEN_CASES[case][0]()
        with contextlib.redirect_stdout(io.StringIO()), np.errstate(all="ignore"):
            yN, G, h, lam, ys, _ = dual.solveBatch(prob.fg, prob.y0(), nIter=n_iter)
        rng = np.random.RandomState(123)
        labels = {"xent": (rng.rand(prob.B, prob.n) < 0.3).astype(np.float64), "mse": rng.rand(prob.B, prob.n)}
        for loss, fn in (("xent", xent), ("mse", mse)):
            idx, c_rows, v_rows, cy_all = [], [], [], []
            for j in range(prob.B):
                if len(G[j]) == 0:
                    continue
                with np.errstate(all="ignore"):
                    cy, clam, ct = fn(yN[j], labels[loss][j], np.array(G[j]))
                cy_all.append(np.asarray(cy, dtype=np.float64))
                for i in range(len(G[j])):
                    idx.append(j)
                    v_rows.append(lam[j][i] * cy + clam[i] * (yN[j] - ys[j][i]))
                    c_rows.append(clam[i])
            out = os.path.join(REPO, "tests", "golden", "feed__%s__%s.npz" % (case, loss))
            np.savez_compressed(out, idx=np.array(idx), c=np.array(c_rows, dtype=np.float64),
                                v=np.array(v_rows, dtype=np.float64).reshape(len(idx), prob.n),
                                labels=labels[loss], n_iter=np.array(n_iter))
            print("%-16s %-5s rows %d  sum|v| %.12g" % (case, loss, len(idx), np.abs(np.array(v_rows)).sum()))


if __name__ == "__main__":
    main()
