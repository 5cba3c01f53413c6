"""The reference's own sensitivity to the float32 summation order of the PICNN (test infrastructure).

float32 dot products are order dependent and TensorFlow's order is unknowable, so "the reference" evaluated with
NumPy's sgemm order and with the k-ordered fma chain of v_mfma_f32_16x16x4_f32 (oracle/picnn_chain.c) are two
equally valid instances of it.  Their gradients differ by ~3e-7 relative; the bundle method amplifies that (pivot,
bound-set and pruning decisions are discontinuous, the un-line-searched Newton iteration has attracting limit
cycles).  `oracle_pair` runs the SAME oracle solver under both orders on identical inputs; the distribution of
|y*_sgemm - y*_chain| is the band inside which no implementation can be told from the reference.
"""
import numpy as np

from oracle import bundle_entropy_oracle as oracle
from oracle import picnn_oracle


def bibtex_problem(B, seed=0, regime="spread"):
    from icnn_amd import picnn
    spec = picnn.bibtex_spec()
    params = picnn.init_params(spec, seed, regime)
    rng = np.random.RandomState(seed + 100)
    x = (rng.rand(B, spec.n_features) < 0.04).astype(np.float32)
    fg = picnn_oracle.make_fg(params, x, list(spec.szs))
    ctx = picnn_oracle.flat_context(fg.ctx)
    return spec, params, ctx


def oracle_pair(spec, params, ctx, n_iter):
    """(result with sgemm-order PICNN, result with MFMA-chain-order PICNN) of the oracle solver, same context rows."""
    B = ctx.shape[0]
    out = []
    for make in (picnn_oracle.make_fg_from_context, picnn_oracle.make_fg_chain):
        fg = make(params, ctx, list(spec.szs), spec.alpha)
        with np.errstate(all="ignore"):
            out.append(oracle.solve_batch(fg, np.full((B, spec.n_labels), 0.5), n_iter))
    return out[0], out[1]


def tail(dy):
    """The quantiles every comparison is made on: per-sample max|dy| -> median, p90, p99, share above 1e-5, max."""
    dy = np.asarray(dy)
    return {"median": float(np.median(dy)), "p90": float(np.quantile(dy, 0.9)), "p99": float(np.quantile(dy, 0.99)),
            "frac_above_1e-5": float((dy > 1e-5).mean()), "max": float(dy.max())}


def per_sample(a, b):
    return np.max(np.abs(a - b), axis=1)


def assert_inside_band(dy_test, dy_band, B, what=""):
    """`dy_test` (implementation vs sgemm-order oracle) must not have a heavier tail than `dy_band` (chain-order oracle
    vs sgemm-order oracle): every quantile at most the band's, with room for one sample and for the float64 noise
    between the implementation and the chain-order oracle (1e-7, the tier-F tolerance)."""
    t, b = tail(dy_test), tail(dy_band)
    msg = "%s implementation %s vs oracle band %s" % (what, t, b)
    assert t["median"] <= 1.05 * b["median"] + 1e-7, msg
    assert t["p90"] <= 1.05 * b["p90"] + 1e-7, msg
    assert t["frac_above_1e-5"] <= b["frac_above_1e-5"] + 1.0 / B, msg
    assert t["max"] <= 1.05 * b["max"] + 1e-7, msg
    return t, b
