#!/usr/bin/env python3
"""Golden vectors for the RL agent's Adam inner optimiser from the REFERENCE's own method.

RL/src/icnn.py imports TensorFlow/tflearn at module level and cannot be imported here, but `Agent.adam`
(:160-215) is pure NumPy: its `def` node is lifted out of the class with `ast` and executed unmodified with a
dummy `self` (nothing is copied into this repository).  It is driven with `_fg_entr`-shaped closures: the seeded
quadratics of tests/problems.py wrapped by oracle/adam_oracle.entropy_fg.  The iteration count is parsed from
the line the method prints.  Output: tests/golden/adam__<case>.npz.
"""
import ast
import contextlib
import io
import os
import re
import sys

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, REPO)
import problems  # noqa: E402
from oracle import adam_oracle  # noqa: E402

REF = "/root/reference/RL/src/icnn.py"


def lift_method(path, cls, name):
    tree = ast.parse(open(path).read())
    klass = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls)
    node = next(n for n in klass.body if isinstance(n, ast.FunctionDef) and n.name == name)
    ns = {"np": np, "npr": np.random}
    exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), ns)
    return ns[name]


def main():
    ref_adam = lift_method(REF, "Agent", "adam")

    class Self:                       # the method only touches self.dimA (and self.adam_plot when plotting)
        pass

    out_dir = os.path.join(REPO, "tests", "golden")
    for case, make in problems.ADAM_CASES.items():
        obs, n, neg_q = make()
        me = Self()
        me.dimA = n
        calls = [0]
        inner = adam_oracle.entropy_fg(neg_q)

        def func(o, a):
            calls[0] += 1
            return inner(o, a)

        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            best = ref_adam(me, func, obs)
        m = re.search(r"Adam took (\d+) iterations", buf.getvalue())
        iters = int(m.group(1)) if m else 1000
        f_at_best, _ = inner(obs, best)
        np.savez(os.path.join(out_dir, "adam__%s.npz" % case), act_best=best, iters=np.int64(iters),
                 func_calls=np.int64(calls[0]), f_at_best=f_at_best)
        print(case, "iters", iters, "calls", calls[0], "act_best[0]", best[0][:4])


if __name__ == "__main__":
    main()
