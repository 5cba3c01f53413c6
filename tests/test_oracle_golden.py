"""Pin the CPU oracle against vectors produced by the reference solver itself.

The fixtures in tests/golden/ were written by oracle/gen_golden.py, which imports
lib/bundle_entropy_dual.py, RL/src/bundle_entropy.py and lib/bundle_entropy.py
from the reference checkout and runs their solveBatch on tests/problems.py.
"""
import os

import numpy as np
import pytest

import problems
from golden_util import assert_matches_golden, flatten_slots, load_golden
from oracle import bundle_entropy_oracle as oracle

CASES = sorted(problems.GOLDEN_CASES)


@pytest.mark.parametrize("variant", ["dual", "rl", "pdipm"])
@pytest.mark.parametrize("case", CASES)
def test_oracle_reproduces_reference(case, variant):
    gold = load_golden(case, variant)
    assert str(gold["error"]) == "", "reference raised: %s" % gold["error"]
    factory, n_iter = problems.GOLDEN_CASES[case]
    prob = factory()
    y0 = prob.y0()
    with np.errstate(all="ignore"):
        res = oracle.solve_batch(prob.fg, y0, n_iter, variant=variant)
    assert res.y is y0, "solveBatch must update the caller's array in place"
    got = flatten_slots(res.y, res.G, res.h, res.ys, res.active, res.lam, res.n_iters, n_iter)
    # Same NumPy/LAPACK calls on the same data: agreement is expected to the last
    # few bits; 1e-12 leaves room for a different BLAS build on another machine.
    assert_matches_golden(got, gold, y_tol=1e-12, lam_tol=1e-10, what="%s/%s" % (case, variant))


@pytest.mark.parametrize("case", CASES)
def test_pdipm_variant_is_a_loose_cross_check(case):
    """lib/bundle_entropy.py (what the icnn_ebundle.py scripts literally import)
    solves the same subproblem by an interior-point method; it agrees with the
    oracle of record only loosely, which is why it is not the
    oracle.  This documents how loose."""
    gold = load_golden(case, "pdipm")
    dual = load_golden(case, "dual")
    # Measured when the fixtures were made: median |dy| 5e-10 .. 3e-5, but the
    # maximum reaches 2e-5 (lse_n159), 1.7e-3 (lse_n33) and 0.18 (action_box, where
    # the un-line-searched dual Newton of variant "dual" stalls) -- not a 1e-5 oracle.
    assert np.median(np.abs(gold["y"] - dual["y"])) < 1e-4


def test_reference_tuple_shape():
    prob = problems.max_affine(1, 4, 9, 6)
    y0 = prob.y0()
    y, A, b, lam, xs, n_iters = oracle.solveBatch(prob.fg, y0, nIter=5)
    assert y is y0
    for u in range(4):
        assert len(A[u]) == len(b[u]) == len(xs[u]) == len(lam[u])
        assert all(l > 0 for l in lam[u])
        assert all(a.shape == (9,) for a in A[u])


def test_callback_protocol():
    prob = problems.max_affine(1, 4, 9, 6)
    seen = []
    oracle.solve_batch(prob.fg, prob.y0(), 3, callback=lambda t, f, y: seen.append((t, f.shape, y.shape)))
    assert [s[0] for s in seen] == [0, 1, 2]
    seen = []
    oracle.solve_batch(prob.fg, prob.y0(), 3, callback=lambda t, f: seen.append(t), variant="rl")
    assert seen == [0, 1, 2]


def test_softplus_matches_definition():
    v = np.linspace(-40, 40, 161)
    assert np.allclose(oracle.softplus_stable(v), np.logaddexp(0, v), rtol=1e-14, atol=0)


@pytest.mark.parametrize("name", ["fc_multilabel", "fc_rl_leaky", "fc_rl_relu", "conv_completion"])
def test_picnn_oracles_reproduce_their_committed_fixtures(name):
    """tests/golden/picnn__*.npz (oracle/gen_picnn_fixtures.py): (params, x, y) -> (E, dE/dy) of the three PICNNs as the
    CPU oracles evaluate them.  The PICNN oracles are UNPINNED (TensorFlow r0.10 / tflearn are not installable here); the
    fixtures exist so that someone with that stack can pin them (oracle/pin_picnn_with_tflearn.py) -- this test only keeps
    oracle and fixtures in step, and checks that the RL network is stored under both readings of tflearn's leaky_relu."""
    import json
    from icnn_amd import picnn
    from oracle import picnn_conv_oracle, picnn_oracle
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "picnn__%s.npz" % name))
    meta = json.loads(str(z["meta"]))
    if name == "conv_completion":
        import torch
        cs = picnn.ConvSpec(H=meta["H"], W=meta["W"])
        params = picnn.init_conv_params(cs, 5, "spread")
        for k, (s1, s2) in meta["param_checksums"].items():
            v = np.asarray(params[k], dtype=np.float64)
            assert abs(v.sum() - s1) <= 1e-9 * (1 + abs(s1)) and abs(np.abs(v).sum() - s2) <= 1e-9 * (1 + s2), k
        ctx = picnn_conv_oracle.flat_context(picnn_conv_oracle.context(params, torch.from_numpy(z["x"])))
        E, g = picnn_conv_oracle.make_fg_from_context(params, ctx, cs.H, cs.W)(z["y"])
        tol = 1e-5            # torch's convolution kernels may sum in another order on another host
    else:
        params = {k[len("param:"):]: z[k] for k in z.files if k.startswith("param:")}
        fg = picnn_oracle.make_fg(params, z["x"], meta["layer_sizes"], meta["alpha"], meta["batchnorm"],
                                  "action" if meta["action_box"] else None)
        E, g = fg(z["y"])
        tol = 1e-6
    assert E.dtype == np.float32 and g.dtype == np.float32
    assert np.max(np.abs(E - z["E"])) <= tol * (1 + np.abs(z["E"]).max())
    assert np.max(np.abs(g - z["dE_dy"])) <= tol * (1 + np.abs(z["dE_dy"]).max())
    if name == "fc_rl_leaky":
        assert meta["alpha"] == 0.01
    if name == "fc_rl_relu":
        assert meta["alpha"] == 0.0


def test_single_sample_solve_of_the_reference_is_the_batch_of_one():
    """lib/bundle_entropy_dual.py::solve (:87-127), run by oracle/gen_golden.py on eight samples of four golden problems (one
    sample at a time): on problems whose bundles keep full rank it is the batch solver on a batch of one -- the restatement's
    solve_batch reproduces its output (the single-sample form has no rank test; none of these samples trips it).  The GPU half
    (tests/test_gpu_parity.py) holds icnn_amd.bundle_entropy.solve to the same fixture."""
    import os
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "solve__dual.npz"))
    for case in ("c1_quadratic", "lse_n159", "maxaffine_n159", "lse_n33"):
        factory, n_iter = problems.GOLDEN_CASES[case]
        prob = factory()
        y0 = prob.y0()
        for row, u in enumerate(gold[case + "__samples"]):
            def fg1(Y, u=u):
                full = np.array(y0, copy=True)
                full[u] = Y[0]
                f, g = prob.fg(full)
                return f[u:u + 1], g[u:u + 1]
            with np.errstate(all="ignore"):
                res = oracle.solve_batch(fg1, np.array(y0[u:u + 1], copy=True), n_iter)
            # float64 cuts: the same arithmetic, 1e-12.  float32 cuts: the single-sample form lets the dtype of the gradient
            # leak into the iterate -- x = 1/(1+exp(A[0])) stays a float32 ARRAY for the second iteration (:119), so that
            # iteration's offset b = f - dot(g, x) (:98) is a float32 dot product, where solveBatch stores the iterate in
            # the float64 batch array (:168) -- and lands 1e-7 from the batch form (measured 1.7e-7): BASELINE's 1e-5 applies
            tol = 1e-12 if prob.cut_dtype == np.float64 else (1e-4 if case == "lse_n33" else 1e-5)   # lse_n33 amplifies the 1e-7 tenfold per iteration (DUAL_Y_TOL)
            assert np.max(np.abs(res.y[0] - gold[case + "__y"][row])) <= tol, (case, u)
