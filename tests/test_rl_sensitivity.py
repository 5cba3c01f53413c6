"""RL variant (RL/src/bundle_entropy.py) on degenerate bundles: how far the REFERENCE moves against itself.

The RL variant has no rank test, so repeated cuts stay in the bundle and the reduced Newton system
(rl :52-62, np.linalg.solve = LAPACK dgesv) is singular in exact arithmetic.  Whether dgesv then reports an exact
zero pivot (-> `break`, lam kept) or divides by a pivot of the size of the rounding noise (-> a unit-size step along
the null direction) depends on the last bits produced by the BLAS kernels that built the Hessian and ran the
elimination.  NumPy's OpenBLAS selects those kernels by CPU family at run time; `OPENBLAS_CORETYPE` overrides the
choice.  `oracle/gen_golden.py --coretype X` ran the reference itself that way for Haswell (what AVX2 hosts such
as AMD EPYC select), Sandybridge and Nehalem next to the fixtures of record (SkylakeX, this container's CPU):

    case                  spread of y* between the four reference runs
    maxaffine_f64         1.9e-1   (different active sets)
    maxaffine_n159_long   3.4e-2   (different active sets)
    n_equals_1            3.5e-3
    lse_n33               4.0e-5   (different active sets)
    the other six         <= 4.3e-6

So on those four problems "y* within 1e-5 of the reference" is not defined by the reference itself; the HIP path
is held to twice the reference's own spread there and to 1e-5 elsewhere (tests/test_gpu_parity.py), and the CPU
model of the device formulation is held to the same bar here.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

import problems
from golden_util import RL_FAMILIES, load_golden, rl_reference_band, rl_sample_check, rl_tolerance
from oracle import device_model

CASES = sorted(problems.GOLDEN_CASES)
DEGENERATE = {"maxaffine_f64", "maxaffine_n159_long", "n_equals_1", "lse_n33"}
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("case", CASES)
def test_reference_spread_over_blas_kernel_families(case):
    band, same_counts, same_iters = rl_reference_band(case)
    assert same_iters
    if case in DEGENERATE:
        assert band > 1e-5, "%s: the reference reproduces itself (%.2e), tighten the GPU tolerance" % (case, band)
    else:
        assert band <= 1e-5 and same_counts, "%s: spread %.2e" % (case, band)


_CHILD = r"""
import sys, numpy as np
sys.path.insert(0, %(repo)r); sys.path.insert(0, %(tests)r)
import problems
from golden_util import load_golden
from oracle import bundle_entropy_oracle as oracle
from threadpoolctl import threadpool_info
arch = [i.get("architecture") for i in threadpool_info() if i.get("internal_api") == "openblas"]
if not arch or arch[0].lower() != %(core)r.lower():
    print("SKIP", arch); sys.exit(0)
worst = 0.0
for case in sorted(problems.GOLDEN_CASES):
    factory, n_iter = problems.GOLDEN_CASES[case]
    prob = factory()
    with np.errstate(all="ignore"):
        res = oracle.solve_batch(prob.fg, prob.y0(), n_iter, variant="rl")
    gold = load_golden(case, "rl@" + %(core)r.lower())
    worst = max(worst, float(np.max(np.abs(res.y - gold["y"]))))
    assert np.array_equal([len(a) for a in res.active], gold["cnt"]), case
print("WORST %%.3e" %% worst)
"""


@pytest.mark.parametrize("core", ["Haswell", "Sandybridge"])
def test_oracle_follows_the_reference_under_another_kernel_family(core):
    """The restatement goes through the same NumPy/LAPACK calls as the reference, so with OpenBLAS forced to another
    kernel family it must land on that family's reference run (and not on the SkylakeX fixture of record)."""
    env = dict(os.environ, OPENBLAS_CORETYPE=core)
    code = _CHILD % dict(repo=REPO, tests=os.path.join(REPO, "tests"), core=core)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    if "SKIP" in out.stdout:
        pytest.skip("this host's OpenBLAS does not run the %s kernels: %s" % (core, out.stdout.strip()))
    assert float(out.stdout.split("WORST")[1]) <= 1e-12, out.stdout


@pytest.mark.parametrize("case", CASES)
def test_device_formulation_within_reference_spread(case):
    """oracle/device_model.py, variant rl (unpivoted elimination, exact zero pivot -> keep lam, as the HIP kernel):
    within max(1e-5, 2 x reference spread) of the fixture of record; identical discrete outcomes wherever the
    reference agrees with itself on them."""
    factory, n_iter = problems.GOLDEN_CASES[case]
    prob = factory()
    with np.errstate(all="ignore"):
        res = device_model.solve_batch_device(prob.fg, prob.y0(), n_iter, variant="rl")
    gold = load_golden(case, "rl")
    tol, same_counts, _ = rl_tolerance(case)
    dy = float(np.max(np.abs(res.y - gold["y"])))
    assert dy <= tol, "%s: %.3e > %.1e" % (case, dy, tol)
    assert np.array_equal(res.n_iters, gold["n_iters"])
    if same_counts:
        assert np.array_equal([len(a) for a in res.active], gold["cnt"])
    # per sample (the bar the GPU test holds the kernel to): 1e-5 to the nearest reference run wherever the reference
    # reproduces itself on that sample, one spread elsewhere; same active-set sizes on the reproducible samples
    excess, strict, _, cnt_agree = rl_sample_check(case, res.y)
    assert excess <= 1.0, "%s: %.2f x the per-sample tolerance" % (case, excess)
    assert np.array_equal(np.array([len(a) for a in res.active])[cnt_agree], gold["cnt"][cnt_agree])
    if case not in DEGENERATE:
        assert strict == prob.B


def test_fixture_families_are_complete():
    for case in CASES:
        for fam in RL_FAMILIES:
            assert str(load_golden(case, fam)["error"]) == ""
