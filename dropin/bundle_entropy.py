"""Drop-in for the reference's `bundle_entropy` module (lib/bundle_entropy_dual.py /
lib/bundle_entropy.py as imported by multi-label-cls/icnn_ebundle.py:27-30 and
completion/icnn_ebundle.py:28-31): same name, same `solveBatch` signature, GPU inside."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from icnn_amd.bundle_entropy import solveBatch  # noqa: E402,F401
