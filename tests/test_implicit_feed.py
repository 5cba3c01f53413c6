"""Implicit-differentiation feed: oracle pinned against the reference's own
crossEntrGrad / mseGrad (CPU), HIP kernel against the oracle and the same golden vectors (GPU)."""
import os

import numpy as np
import pytest

import problems
from golden_util import GOLDEN_DIR
from oracle import bundle_entropy_oracle as oracle
from oracle import implicit_feed_oracle as feed_oracle

CASES = {"maxaffine_n159": 10, "lse_n33": 12, "zero_gradient": 6}


def _load(case, loss):
    z = np.load(os.path.join(GOLDEN_DIR, "feed__%s__%s.npz" % (case, loss)))
    return {k: z[k] for k in z.files}


@pytest.mark.parametrize("loss", ["xent", "mse"])
@pytest.mark.parametrize("case", sorted(CASES))
def test_feed_oracle_reproduces_reference(case, loss):
    gold = _load(case, loss)
    prob = problems.GOLDEN_CASES[case][0]()
    with np.errstate(all="ignore"):
        y, A, b, lam, xs, _ = oracle.solveBatch(prob.fg, prob.y0(), nIter=CASES[case])
        idx, rows_y, rows_v, rows_c = feed_oracle.feed_rows(y, gold["labels"], A, xs, lam, loss)
    assert np.array_equal(idx, gold["idx"])
    assert np.allclose(rows_c, gold["c"], rtol=1e-9, atol=1e-12)
    assert np.allclose(rows_v, gold["v"], rtol=1e-9, atol=1e-11)


@pytest.mark.gpu
@pytest.mark.parametrize("loss", ["xent", "mse"])
@pytest.mark.parametrize("case", sorted(CASES))
def test_feed_kernel_matches_reference_golden(case, loss):
    from icnn_amd import bundle_entropy
    gold = _load(case, loss)
    prob = problems.GOLDEN_CASES[case][0]()
    res = bundle_entropy.solveBatch(prob.fg, prob.y0(), nIter=CASES[case], native=True)
    feed = bundle_entropy.implicit_feed(res, gold["labels"], loss)
    idx = feed.sample.cpu().numpy()
    assert np.array_equal(idx, gold["idx"])
    # lse_n33 is the ill-conditioned smooth case: y* itself agrees to 5e-7 only and the
    # KKT solve amplifies that
    tol = 1e-3 if case == "lse_n33" else 1e-6
    scale_c = 1.0 + np.abs(gold["c"])
    assert np.all(np.abs(feed.c.cpu().numpy() - gold["c"]) <= tol * scale_c)
    scale_v = 1.0 + np.abs(gold["v"])
    assert np.all(np.abs(feed.v.cpu().numpy() - gold["v"]) <= tol * scale_v)
    # the y rows are the points the active cuts were taken at
    host_ys = res.ys.cpu().numpy()
    act, cnt = res.active.cpu().numpy(), res.count.cpu().numpy()
    want = np.concatenate([host_ys[j, act[j, :cnt[j]]] for j in range(prob.B)], axis=0)
    assert np.array_equal(feed.y.cpu().numpy(), want)


@pytest.mark.gpu
def test_feed_kernel_wide_rows_with_more_slots_than_lds_rows():
    """The completion model's feed (mseGrad, completion/icnn_ebundle.py:493-522) after 31 bundle iterations: n = 2048 with
    31 slots does not fit the LDS of the feed kernel's workgroup, the bundle is staged in st.scratch like the dual step's
    (be_dual_dev.h GLB).  Against the oracle's feed rows computed from the same solve."""
    import torch
    from icnn_amd import bundle_entropy, picnn
    spec = picnn.ConvSpec()
    B, n_iter = 3, 31
    params = picnn.init_conv_params(spec, 5, "spread")
    x = np.random.RandomState(55).rand(B, spec.H, spec.W, 1).astype(np.float32)
    model = picnn.ConvModel(spec, params)
    ctx = model.context(torch.from_numpy(x))
    y0 = np.repeat((0.2 + 0.6 * np.random.RandomState(9).rand(spec.n_labels))[None], B, axis=0)
    res = bundle_entropy.FusedSolver(model, B, n_iter, "dual").solve(ctx, torch.from_numpy(y0).cuda())
    labels = np.random.RandomState(4).rand(B, spec.n_labels)
    feed = bundle_entropy.implicit_feed(res, labels, "mse")
    y = res.y.cpu().numpy()
    G, ys, lam = res.state.G.cpu().numpy(), res.state.ys.cpu().numpy(), res.lam.cpu().numpy()
    act, cnt = res.active.cpu().numpy(), res.count[:B].cpu().numpy()
    assert cnt.max() > 12
    A = [[G[j, s].astype(np.float64) for s in act[j, :cnt[j]]] for j in range(B)]
    xs = [[ys[j, s] for s in act[j, :cnt[j]]] for j in range(B)]
    lams = [lam[j, :cnt[j]] for j in range(B)]
    idx, rows_y, rows_v, rows_c = feed_oracle.feed_rows(y, labels, A, xs, lams, "mse")
    assert np.array_equal(feed.sample.cpu().numpy(), idx)
    assert np.array_equal(feed.y.cpu().numpy(), rows_y)
    assert np.all(np.abs(feed.c.cpu().numpy() - rows_c) <= 1e-6 * (1.0 + np.abs(rows_c)))
    assert np.all(np.abs(feed.v.cpu().numpy() - rows_v) <= 1e-6 * (1.0 + np.abs(rows_v)))
