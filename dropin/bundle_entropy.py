"""Drop-in for the reference's `bundle_entropy` module as multi-label-cls/icnn_ebundle.py:27-30 and
completion/icnn_ebundle.py:28-31 import it: `sys.path.append('../lib'); import bundle_entropy` resolves to
lib/bundle_entropy.py, the primal-dual interior-point variant, `solveBatch(fg, initXs, nIter=10, callback=None,
solver='pc')` (:192).  Same name, same signature, GPU inside; `solver='pc'` is the default here as there.
For the dual projected-Newton module lib/bundle_entropy_dual.py use dropin/bundle_entropy_dual.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from icnn_amd.bundle_entropy import solveBatch as _solve  # noqa: E402


def solveBatch(fg=None, initXs=None, nIter=10, callback=None, solver="pc", **kw):
    kw.pop("variant", None)
    return _solve(fg, initXs, nIter, callback, solver, **kw)
