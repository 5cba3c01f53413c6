"""Drop-in for lib/bundle_entropy_dual.py (`solveBatch(fg, initXs, nIter=10, callback=None)`, :129): the dual
projected-Newton variant, BASELINE.json's oracle of record for the Bibsonomy / completion configurations."""
import functools
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from icnn_amd.bundle_entropy import solveBatch as _solve  # noqa: E402

solveBatch = functools.partial(_solve, variant="dual")
