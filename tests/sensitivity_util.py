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


def oracle_triple(spec, params, ctx, n_iter):
    """oracle_pair plus a third, unrelated float32 summation order of the PICNN (products rounded to float32, NumPy's
    pairwise tree: oracle/picnn_oracle.py energy_and_grad_pairwise), so that the band is not a two-point estimate."""
    B = ctx.shape[0]
    a, b = oracle_pair(spec, params, ctx, n_iter)
    fg = picnn_oracle.make_fg_pairwise(params, ctx, list(spec.szs), spec.alpha)
    with np.errstate(all="ignore"):
        c = oracle.solve_batch(fg, np.full((B, spec.n_labels), 0.5), n_iter)
    return a, b, c


def rounding_errors(spec, params, ctx, y):
    """Error of the float32 PICNN against its float64 evaluation for the three CPU summation orders:
    {order: (rms error of E, max error of E, rms error of dE/dy, max error of dE/dy)} and the float64 (E, dE/dy)."""
    szs = list(spec.szs)
    layers = picnn_oracle.unflatten_context(ctx, spec.n_labels, szs + [1])
    E64, g64 = picnn_oracle.energy_and_grad_f64(params, layers, y, szs, spec.alpha)
    out = {}
    for name, (E, g) in (("sgemm", picnn_oracle.energy_and_grad(params, layers, np.asarray(y, np.float32), szs, spec.alpha)),
                         ("chain", picnn_oracle.energy_and_grad_chain(params, ctx, y, szs, spec.alpha)),
                         ("pairwise", picnn_oracle.energy_and_grad_pairwise(params, layers, y, szs, spec.alpha))):
        out[name] = error_stats(E, g, E64, g64)
    return out, (E64, g64)


def error_stats(E, g, E64, g64):
    dE, dg = np.asarray(E, np.float64) - E64, np.asarray(g, np.float64) - g64
    return (float(np.sqrt(np.mean(dE ** 2))), float(np.abs(dE).max()), float(np.sqrt(np.mean(dg ** 2))), float(np.abs(dg).max()))


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
