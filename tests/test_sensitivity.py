"""CPU part of the tier-B parity argument: the oracle against ITSELF under two float32 summation orders of the PICNN.

BASELINE.json asks for y* within 1e-5 of the reference CPU solveBatch on identical inputs.  With the PICNN's
float32 sums in another order the inputs of the solver are no longer identical (|dg| ~ 3e-7 |g|), and the reference
algorithm turns that into differences far above 1e-5 on a small share of the samples.  These tests measure that
share with no GPU involved; tests/test_gpu_parity.py::test_fused_tail_is_inside_the_oracles_own_band then requires
the HIP path to stay inside it instead of inside hand-set thresholds.
"""
import numpy as np
import pytest

from sensitivity_util import bibtex_problem, oracle_pair, oracle_triple, per_sample, rounding_errors, tail


def test_gradients_of_the_two_orders_differ_only_by_float32_rounding():
    from oracle import picnn_oracle
    spec, params, ctx = bibtex_problem(32)
    y = np.random.RandomState(3).rand(32, spec.n_labels)
    f1, g1 = picnn_oracle.make_fg_from_context(params, ctx, list(spec.szs))(y)
    f2, g2 = picnn_oracle.make_fg_chain(params, ctx, list(spec.szs))(y)
    assert np.max(np.abs(g1 - g2)) <= 2e-6 * np.max(np.abs(g1))
    assert np.max(np.abs(f1 - f2)) <= 2e-6 * np.max(np.abs(f1))
    assert not np.array_equal(g1, g2), "the two orders are expected to differ in the last bits"


@pytest.mark.parametrize("B,n_iter", [(128, 10), (64, 30)])
def test_reference_moves_by_more_than_1e5_between_summation_orders(B, n_iter):
    """BASELINE.json configs[1] (B = 128, nIter = 10) and the nIter = 30 shape of configs[3]."""
    spec, params, ctx = bibtex_problem(B)
    a, b = oracle_pair(spec, params, ctx, n_iter)
    t = tail(per_sample(a.y, b.y))
    print("oracle(sgemm) vs oracle(chain), B=%d nIter=%d: %s" % (B, n_iter, t))
    # typical sample: float32-level agreement ...
    assert t["median"] <= (5e-6 if n_iter <= 10 else 5e-5)
    # ... but the tail is orders of magnitude above BASELINE's 1e-5: that is the reference's own sensitivity
    assert t["max"] > 1e-4
    assert t["frac_above_1e-5"] > 0
    # discrete outcomes differ on the same samples the tail comes from
    diff = [u for u in range(B) if list(a.active[u]) != list(b.active[u]) or a.n_iters[u] != b.n_iters[u]]
    worst = int(np.argmax(per_sample(a.y, b.y)))
    assert n_iter <= 10 or len(diff) > 0
    assert per_sample(a.y, b.y)[worst] == t["max"]


def test_band_is_the_same_between_any_two_of_three_summation_orders():
    """the band |y*(order A) - y*(order B)| was a two-point estimate (NumPy's sgemm order against the MFMA
    chain order the kernels use).  A third order that shares nothing with either (every product rounded to float32, NumPy's
    pairwise tree) gives three pairs.  If the kernel's order were a particularly bad -- or particularly lucky -- instance,
    the pairs that contain it would stand out; they do not: every pair has a float32-level median, a tail far above 1e-5,
    and the three medians / 90 % quantiles agree within a factor of four."""
    B, n_iter = 128, 10
    spec, params, ctx = bibtex_problem(B)
    a, b, c = oracle_triple(spec, params, ctx, n_iter)
    tails = {"sgemm-chain": tail(per_sample(a.y, b.y)), "sgemm-pairwise": tail(per_sample(a.y, c.y)),
             "chain-pairwise": tail(per_sample(b.y, c.y))}
    for k, t in tails.items():
        print("%-15s %s" % (k, t))
        assert t["median"] <= 5e-6 and t["max"] > 1e-5, (k, t)
    for q in ("median", "p90"):
        vals = [t[q] for t in tails.values()]
        assert max(vals) <= 4.0 * min(vals), (q, tails)


def test_mfma_chain_order_is_not_a_worse_float32_instance_than_sgemm():
    """VERDICT r3 5(b), CPU half (the GPU half asserts the same of the kernel's own output, which is bit-identical to the
    chain order): rounding error of E and dE/dy against the float64 evaluation of the same float32-parameter network.  The
    k-ordered fma chain of the MFMA is within 1.5 x of NumPy's sgemm order on the root-mean-square error and within 2 x on
    the worst element -- it is another instance of "the reference's float32 fg", not a sloppier one."""
    spec, params, ctx = bibtex_problem(64)
    for seed in (3, 4):
        y = np.random.RandomState(seed).rand(64, spec.n_labels)
        err, _ = rounding_errors(spec, params, ctx, y)
        print(seed, err)
        for j in (0, 2):                                   # rms of E, rms of dE/dy
            assert err["chain"][j] <= 1.5 * err["sgemm"][j], (seed, j, err)
        for j in (1, 3):                                   # worst element
            assert err["chain"][j] <= 2.0 * err["sgemm"][j], (seed, j, err)
        assert err["pairwise"][2] <= err["sgemm"][2]       # (the pairwise tree is the most accurate of the three)
