"""Drop-in for RL/src/bundle_entropy.py (imported at RL/src/icnn.py:8): install as
`bundle_entropy.py` next to the agent.  Same signature, RL variant semantics (Armijo line
search, clip to [0.03, 0.97], per-sample early stop, nIter default 5, callback(t, f))."""
import functools
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from icnn_amd.bundle_entropy import solveBatch as _solve  # noqa: E402

solveBatch = functools.partial(_solve, variant="rl")
