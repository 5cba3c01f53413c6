"""Drop-in for lib/bundle_entropy_dual.py (`solveBatch(fg, initXs, nIter=10, callback=None)`, :129; the single-sample
`solve(fg, initX, nIter=10, callback=None)`, :87): the dual projected-Newton variant, BASELINE.json's oracle of record for
the Bibsonomy / completion configurations."""
import functools
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from icnn_amd.bundle_entropy import solve as _solve_one  # noqa: E402
from icnn_amd.bundle_entropy import solveBatch as _solve  # noqa: E402

solveBatch = functools.partial(_solve, variant="dual")
solve = functools.partial(_solve_one, variant="dual")
