"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle.

Tolerances
  generic mode (same f, g handed to both sides): |y - y_oracle| <= 1e-9 and identical
      discrete outcomes (active slots, nIters).  Remaining differences are float64
      summation order (MFMA contraction vs BLAS) only.
  fused mode (PICNN evaluated in float32 on both sides, different summation order):
      BASELINE.json's tolerance, |y* - y*_ref| <= 1e-5.
"""
import numpy as np
import pytest
import torch

import problems
from golden_util import assert_matches_golden, load_golden, rl_sample_check, rl_tolerance
from gpu_util import compare_with_oracle, flatten_result, result_to_host
from oracle import bundle_entropy_oracle as oracle
from oracle import picnn_oracle

pytestmark = pytest.mark.gpu

DUAL_CASES = sorted(problems.GOLDEN_CASES)
# Smooth energies: the bundle's cuts become nearly parallel near convergence and the
# reference algorithm itself amplifies 1e-16 perturbations by about 10x per outer iteration
# (measured: 3e-16 at t=1 -> 5e-7 at t=11 on lse_n33), so only a looser bound is meaningful.
DUAL_Y_TOL = {"lse_n33": 5e-6, "lse_n159": 1e-7}
def _solve(prob, n_iter, variant, **kw):
    from icnn_amd import bundle_entropy
    y0 = prob.y0()
    res = bundle_entropy.solveBatch(prob.fg, y0, nIter=n_iter, variant=variant, native=True, **kw)
    return y0, res


@pytest.mark.parametrize("case", DUAL_CASES)
def test_dual_variant_matches_reference_golden(case):
    factory, n_iter = problems.GOLDEN_CASES[case]
    prob = factory()
    y0, res = _solve(prob, n_iter, "dual")
    got, host = flatten_result(res, n_iter)
    assert np.array_equal(y0, host["y"]), "initXs must be updated in place"
    gold = load_golden(case, "dual")
    tol = DUAL_Y_TOL.get(case, 1e-9)
    assert_matches_golden(got, gold, y_tol=tol, lam_tol=max(1e-7, 1e3 * tol), chk_rtol=max(1e-7, 100 * tol),
                          what=case)


@pytest.mark.parametrize("case", DUAL_CASES)
def test_rl_variant_matches_reference_golden(case):
    """RL variant (RL/src/bundle_entropy.py: no rank test), ALL ten reference-generated problems.  Tolerance:
    1e-5 on y* with identical nIters and active-set sizes wherever the reference reproduces itself to 1e-5 across
    OpenBLAS kernel families (six problems); on the four degenerate ones (repeated cuts or n = 1: singular Newton
    systems whose LAPACK solution is rounding noise) twice the reference's own spread -- measured by running the
    reference itself under OPENBLAS_CORETYPE = Haswell / Sandybridge / Nehalem, fixtures `*__rl@<family>.npz`,
    tests/test_rl_sensitivity.py."""
    factory, n_iter = problems.GOLDEN_CASES[case]
    prob = factory()
    y0, res = _solve(prob, n_iter, "rl", check=False)
    got, host = flatten_result(res, n_iter)
    gold = load_golden(case, "rl")
    tol, same_counts, _ = rl_tolerance(case)
    assert np.array_equal(got["n_iters"], gold["n_iters"])
    dy = np.max(np.abs(got["y"] - gold["y"]))
    # per sample: 1e-5 to the nearest of the reference's four runs wherever the reference reproduces itself on THAT
    # sample, one spread (not two) elsewhere; identical active-set sizes on the reproducible samples
    singular = (host["status"] & 1) != 0                # ICNN_BE_ST_SINGULAR: an exactly zero pivot, lam kept (rl :55-62)
    excess, strict, d_u, cnt_agree = rl_sample_check(case, host["y"])
    print("%s: max|y - y_ref| = %.3e (case-level tolerance %.1e); per sample: %d of %d held to 1e-5, worst distance / "
          "tolerance %.2f" % (case, dy, tol, strict, prob.B, excess))
    assert dy <= tol, "%s: max|y - y_ref| = %.3e > %.1e" % (case, dy, tol)
    assert excess <= 1.0, "%s: a sample is %.2f x its tolerance away from every reference run" % (case, excess)
    assert np.array_equal(got["cnt"][cnt_agree], gold["cnt"][cnt_agree])
    assert np.isfinite(host["y"]).all()
    assert (host["y"] >= 0.03 - 1e-15).all() and (host["y"] <= 0.97 + 1e-15).all()
    # per sample, whatever the case-level band says about y: the VALUE of the objective f(y) - H(y) at the
    # result must be one the reference itself reaches on that sample (nearest of its four runs, to one spread of their
    # objective values or 1e-7 relative) -- where y* is ill-determined (n = 1 with singular Newton systems, nearly parallel
    # cuts) the objective is flat, and a result that merely sits inside the y-band but on a worse level set fails here
    from golden_util import rl_objective_check
    worst_obj, held = rl_objective_check(case, prob, host["y"], exempt=singular)
    print("%s: objective value per sample: worst distance / tolerance %.2f, %d of %d samples held to 1e-7 relative"
          % (case, worst_obj, held, prob.B))
    assert worst_obj <= 1.0, "%s: a sample's objective value is %.2f x its tolerance away from every reference run" % (case, worst_obj)
    # invariants that survive the degeneracy: the multipliers are a simplex point, y is the clipped entropy-dual image of
    # the sample's OWN final bundle (rl :117-123), and every stored cut is a cut of the problem's f at its point
    # (h = f(ys) - <g(ys), ys>, rl :102-110)
    for u in range(prob.B):
        lam, act = host["lam"][u], host["active"][u]
        assert lam is not None and np.all(lam > 0) and abs(lam.sum() - 1) < 1e-6
        Gu = host["G"][u, act].astype(np.float64)
        img = np.clip(1.0 / (1.0 + np.exp(Gu.T.dot(lam))), 0.03, 0.97)
        assert np.max(np.abs(img - host["y"][u])) <= 1e-9
    pts = host["ys"][:, 0, :].copy()                     # slot 0 is filled for every sample (no rank test in this variant)
    f0, g0 = prob.fg(pts.copy())
    h0 = np.asarray(f0, dtype=np.float64) - np.sum(np.asarray(g0, dtype=np.float64) * pts, axis=1)
    assert np.allclose(host["h"][:, 0], h0, rtol=1e-6, atol=1e-6)
    assert np.allclose(host["G"][:, 0, :], np.asarray(g0), rtol=1e-6, atol=1e-7)
    if same_counts:
        assert np.array_equal(got["cnt"], gold["cnt"])
        assert np.max(np.abs(got["lam"] - gold["lam"])) <= 1e-2   # lam is far worse conditioned than y


def _all_outputs(res, B):
    return [t.cpu().numpy().copy() for t in (res.y, res.lam, res.active, res.count[:B], res.n_iters[:B], res.newton_iters[:B],
                                             res.state.G, res.state.h, res.state.ys, res.finished[:B], res.status[:B])]


SMALL_ROW_PROBLEMS = {
    "c1_quadratic": None, "zero_gradient": None, "action_box": None, "single_sample": None, "n_equals_1": None,   # goldens, n <= 16
    "maxaffine_n8": (lambda: problems.max_affine(21, 37, 8, 6, 1.0), 5),        # n = 8: NumPy's eight-accumulator sum, no tail
    "maxaffine_n12": (lambda: problems.max_affine(22, 50, 12, 9, 1.0), 7),      # ... with a tail
    "maxaffine_n16_f64": (lambda: problems.max_affine(23, 21, 16, 12, 1.0, np.float64), 12),   # two accumulator passes, float64 cuts
    "lse_n13_long": (lambda: problems.log_sum_exp(24, 33, 13, 6, 1.0), 15),     # 15 slots: bundles past eight cuts (16x16 MFMA form)
    "lse_n3": (lambda: problems.log_sum_exp(25, 130, 3, 5, 1.5), 5),
}


@pytest.mark.parametrize("case", sorted(SMALL_ROW_PROBLEMS))
def test_small_rows_four_samples_per_wave_equals_wave_per_sample(case):
    """Narrow rows (n <= 16), variant rl: the default dual step packs four samples into a wave, one per 16-lane DPP row,
    with the bundle in registers and the MFMA contractions replayed as fused multiply-add chains on the VALU
    (be_dual_small_dev.h); ICNN_BE_FLAG_WAVE_PER_SAMPLE keeps the wave-per-sample kernel.  Same operations in the same
    order: every output bit-identical -- on the five reference-generated golden problems with n <= 16 (float64 cuts,
    n = 1 and repeated cuts among them) and on shapes that walk NumPy's pairwise sum through all its branches."""
    from icnn_amd import _lib
    factory, n_iter = SMALL_ROW_PROBLEMS[case] or problems.GOLDEN_CASES[case]
    prob = factory()
    outs = []
    for flags in (0, _lib.FLAG_WAVE_PER_SAMPLE):
        _, res = _solve(prob, n_iter, "rl", check=False, flags=flags)
        outs.append(_all_outputs(res, prob.B))
    for i, (a, b) in enumerate(zip(*outs)):
        assert np.array_equal(a, b), "output %d differs (max |d| = %.3e)" % (i, np.max(np.abs(a.astype(np.float64) - b)))


@pytest.mark.parametrize("B,n_iter", [(210, 5), (1, 5), (1027, 5), (8192, 5), (333, 12), (64, 3)])
def test_small_rows_fused_halfcheetah_equals_wave_per_sample(B, n_iter):
    """The same through the fused solve of the RL agent's network (HalfCheetah, 6 actions, BASELINE.json configs[4]'s
    model; its full replay batch of 8192 included)."""
    from icnn_amd import _lib, bundle_entropy, picnn
    spec = picnn.halfcheetah_spec()
    params, x = _picnn_problem(spec, max(B, 64), 3, "spread", yu_bias=1.0, gate_bias=1.0)
    model = picnn.FCModel(spec, params)
    ctx = model.context(torch.from_numpy(x))[:B].contiguous()
    outs = []
    for flags in (_lib.FLAG_TWO_KERNELS, _lib.FLAG_TWO_KERNELS | _lib.FLAG_WAVE_PER_SAMPLE):
        res = bundle_entropy.FusedSolver(model, B, n_iter, "rl", flags=flags).solve(ctx, 0.5)
        outs.append(_all_outputs(res, B))
    for i, (a, b) in enumerate(zip(*outs)):
        assert np.array_equal(a, b), "output %d differs" % i
    assert outs[0][5].max() > 0


@pytest.mark.parametrize("case", DUAL_CASES)
def test_pdipm_variant_matches_reference_golden(case):
    """Interior-point variant (lib/bundle_entropy.py, solver='pc' -- the module the icnn_ebundle.py scripts import)
    against the outputs of the reference itself on all ten problems: y* within 1e-5 (BASELINE.json's tolerance;
    measured ~1e-10), identical nIters, active-set sizes and lam None-ness, multipliers and cut offsets to 1e-6."""
    factory, n_iter = problems.GOLDEN_CASES[case]
    prob = factory()
    from icnn_amd import bundle_entropy
    y0 = prob.y0()
    res = bundle_entropy.solveBatch(prob.fg, y0, n_iter, None, "pc", native=True)
    got, host = flatten_result(res, n_iter)
    assert np.array_equal(y0, host["y"]), "initXs must be updated in place"
    gold = load_golden(case, "pdipm")
    # lse_n33: nearly parallel cuts -- M + diag(s/z) of the active cuts has a condition number ~1e10, so the MULTIPLIERS carry
    # the rounding noise of whatever arithmetic built the system at the 1e-6 level (how lam is spread over parallel cuts does
    # not matter to y: y* passes the 1e-7 bound below like every other case).  Round 3's arithmetic landed 2e-7 from the
    # reference's OpenBLAS result, round 4's 5.47e-6.  Measured in round 5: IEEE division for every O(k) quantity (s / z,
    # 1 / sum(m1), rc / s, the pivots) leaves the 5.47e-6 unchanged to all printed digits -- the distance comes from the column
    # arithmetic (Hinv = y (1 - y), one logarithm, the summation order of M) -- so the scalars went back to reciprocals (the
    # dual phase is bound by instruction issue), the bound for this case stays 1e-5, and what IS asserted exactly is the
    # discrete outcome: the same active cuts (lam > 1e-8 prunes).
    lam_tol = 1e-5 if case == "lse_n33" else 1e-6
    assert np.array_equal(got["lam"] > 0, gold["lam"] > 0), "active sets differ from the reference's"
    dy = assert_matches_golden(got, gold, y_tol=1e-5, lam_tol=lam_tol, chk_rtol=1e-7, what=case + "/pdipm")
    print("%s: max|y - y_ref| = %.3e" % (case, dy))
    assert dy <= 1e-7, "the interior-point iteration is well conditioned: expected far inside the 1e-5 tolerance"


def test_pdipm_reference_tuple_and_dropin_module():
    """dropin/bundle_entropy.py is what `import bundle_entropy` resolves to for the icnn_ebundle.py scripts: default
    solver 'pc', the reference's 6-tuple with lam pruned at 1e-8 (lib/bundle_entropy.py:234-237)."""
    import importlib.util
    import os
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("dropin_be", os.path.join(repo, "dropin", "bundle_entropy.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    prob = problems.log_sum_exp(4, 8, 21, 6)
    y0 = prob.y0()
    y, A, b, lam, xs, n_iters = mod.solveBatch(prob.fg, y0, nIter=6)
    assert y is y0
    ora = oracle.solveBatch(prob.fg, prob.y0(), 6, variant="pdipm")
    assert n_iters == ora[5]
    assert np.max(np.abs(y - ora[0])) <= 1e-7
    for u in range(8):
        assert len(A[u]) == len(b[u]) == len(xs[u]) == len(lam[u]) == len(ora[1][u])
        assert np.all(lam[u] > 1e-8) and np.allclose(lam[u], ora[3][u], atol=1e-7)


@pytest.mark.parametrize("which,B,n_iter", [("bibtex", 128, 10), ("bibtex", 48, 20)])
def test_fused_pdipm_matches_chain_order_oracle(which, B, n_iter):
    """The interior-point variant with the PICNN evaluated on the device (icnn_be_solve_fc, one launch per phase and
    round) against the oracle's restatement of lib/bundle_entropy.py fed by the order-matched PICNN; nIter = 20 runs
    the 32-slot kernels.  (On the RL agent's action-box network the reference's own pdipm_pc produces NaNs -- scipy's
    cho_solve raises ValueError --, and the reference never pairs the two: RL/src has its own module.)"""
    from icnn_amd import bundle_entropy, picnn
    spec = picnn.bibtex_spec() if which == "bibtex" else picnn.halfcheetah_spec()
    kw = {} if which == "bibtex" else dict(yu_bias=1.0, gate_bias=1.0)
    params, x = _picnn_problem(spec, B, 2, "spread", **kw)
    model = picnn.FCModel(spec, params)
    ctx = model.context(torch.from_numpy(x))
    fg = picnn_oracle.make_fg_chain(params, ctx.cpu().numpy(), list(spec.szs), spec.alpha, spec.action_box)
    y0 = np.full((B, spec.n_labels), 0.5)
    res = bundle_entropy.solveBatch(f=model, ctx=ctx, y0=y0, nIter=n_iter, variant="pdipm", native=True)
    with np.errstate(all="ignore"):
        ora = oracle.solve_batch(fg, np.full((B, spec.n_labels), 0.5), n_iter, variant="pdipm")
    host = result_to_host(res)
    dy, discrete = compare_with_oracle(host, ora)
    print("fused pdipm %s: max|dy| = %.3e, %d discrete differences" % (which, dy.max(), len(discrete)))
    assert not discrete, discrete
    assert dy.max() <= 1e-7


@pytest.mark.parametrize("B,n_iter", [(100, 10), (300, 6), (1100, 10), (1100, 20), (40, 31)])
def test_pdipm_persistent_kernels_equal_two_kernel_rounds(B, n_iter):
    """Round 4: the interior-point variant runs through the persistent kernels as well (a workgroup per 1-4 samples up to four
    samples per CU, a workgroup per 16-sample tile beyond, the dual phase in LDS groups when nIter > 15): same device
    functions as one launch per phase and round, so every output must be bit-identical."""
    from icnn_amd import _lib, bundle_entropy, picnn
    spec = picnn.bibtex_spec()
    params, x = _picnn_problem(spec, max(B, 64), 5, "spread")
    model = picnn.FCModel(spec, params)
    ctx = model.context(torch.from_numpy(x))[:B].contiguous()
    outs, rounds = [], []
    for flags in (0, _lib.FLAG_TWO_KERNELS):
        res = bundle_entropy.FusedSolver(model, B, n_iter, "pdipm", flags=flags).solve(ctx, 0.5)
        outs.append(_all_outputs(res, B))
        rounds.append(res.state.rounds)
    for i, (a, b) in enumerate(zip(*outs)):
        assert np.array_equal(a, b), "output %d differs between the persistent kernel and launch pairs" % i
    assert (outs[0][10] == 0).all() and outs[0][3].max() > 1
    assert rounds == [n_iter, n_iter], rounds


@pytest.mark.parametrize("n,variant,n_iter", [(300, "dual", 8), (700, "dual", 8), (300, "rl", 8), (300, "pdipm", 8),
                                              (2048, "pdipm", 5), (1100, "pdipm", 6), (2048, "pdipm", 12)])
def test_mid_width_rows_match_oracle(n, variant, n_iter):
    """256 < n < 1024: still one wave per sample, but a bundle row spans more than four 64-lane chunks (the generic
    staging loops of the dual step instead of the register-batched ones; the interior-point variant: the column-chunked
    passes on one wave, ipm_solve_wide_one_fn); the interior-point variant also at n >= 1024, where round 5 splits the columns
    over eight waves (ipm_solve_waves): the completion model's n = 2048, n = 1100 (two waves without a chunk), and n = 2048
    past the LDS capacity (rounds whose bundle is staged in device memory: the GSRC instances)."""
    prob = problems.log_sum_exp(31, 6, n, 9, 0.5)
    y0, res = _solve(prob, n_iter, variant, check=False)
    with np.errstate(all="ignore"):
        ora = oracle.solve_batch(prob.fg, prob.y0(), n_iter, variant=variant)
    host = result_to_host(res)
    dy, discrete = compare_with_oracle(host, ora)
    assert not discrete, discrete
    assert dy.max() <= 1e-6, dy.max()


@pytest.mark.parametrize("n,n_iter", [(1040, 16), (2048, 24), (2560, 31)])
def test_pdipm_wide_rows_beyond_the_eight_wave_solver_match_oracle(n, n_iter):
    """The interior-point variant on wide rows where ipm_solve_waves does NOT apply: a piecewise-linear energy of
    sixty pieces keeps every cut active (a bundle of t cuts in round t), so n = 1040 at 16 iterations passes 13 cuts (eight
    waves x the padded sums no longer fit the row: wave 0 solves alone, the status relayed through the slot list), n = 2048
    at 24 iterations passes IPM_KMAX_WAVES = 20 cuts (the same fallback, bundle staged in device memory), and n = 2560 at 31
    iterations is, from 28 rows on, past what the eight-wave carve-up holds next to the variant's five column buffers (8 x 28 x
    29 per-wave doubles + 5 x 2560 x 8 B > 160 KB: launch_dual_step falls back to the one-wave device-memory instance; before
    round 6: hipErrorInvalidValue mid-solve; the interior-point variant's own width limit, five column buffers + four staged
    rows in 160 KB, is n = 2925).  Against the NumPy restatement of lib/bundle_entropy.py on the same cuts."""
    from icnn_amd import bundle_entropy
    prob = problems.max_affine(31 + n, 3, n, 60, 1.0)
    res = bundle_entropy.solveBatch(prob.fg, prob.y0(), nIter=n_iter, variant="pdipm", native=True)
    with np.errstate(all="ignore"):
        ora = oracle.solve_batch(prob.fg, prob.y0(), n_iter, variant="pdipm")
    host = result_to_host(res)
    dy, discrete = compare_with_oracle(host, ora)
    print("pdipm n=%d nIter=%d: cuts max %d, max|dy| %.2e, discrete %s" % (n, n_iter, max(len(a) for a in host["active"]), dy.max(), discrete))
    assert (host["status"] == 0).all()
    assert max(len(a) for a in host["active"]) == n_iter
    assert not discrete, discrete
    assert dy.max() <= 1e-7, dy.max()


def test_single_sample_solve_matches_the_reference():
    """`solve(fg, initX, nIter, callback)` (lib/bundle_entropy_dual.py:87-127) through dropin/bundle_entropy_dual.py against
    outputs of the reference's own function (tests/golden/solve__dual.npz, oracle/gen_golden.py): a new array is returned,
    `initX` is left alone, the callback sees scalar energies and 1-D iterates."""
    import importlib.util
    import os
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("dropin_be_dual", os.path.join(repo, "dropin", "bundle_entropy_dual.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    gold = np.load(os.path.join(repo, "tests", "golden", "solve__dual.npz"))
    for case in ("c1_quadratic", "lse_n159", "maxaffine_n159", "lse_n33"):
        factory, n_iter = problems.GOLDEN_CASES[case]
        prob = factory()
        y0 = prob.y0()
        for row, u in enumerate(gold[case + "__samples"]):
            def fg1(x, u=u):
                full = np.array(y0, copy=True)
                full[u] = x
                f, g = prob.fg(full)
                return f[u], g[u]
            seen = []
            x0 = np.array(y0[u], copy=True)
            x = mod.solve(fg1, x0, n_iter, lambda t, f, xx: seen.append((t, np.ndim(f), xx.shape)))
            assert x is not x0 and np.array_equal(x0, y0[u])
            assert seen[0] == (0, 0, (prob.n,)) and len(seen) == n_iter
            # float64 cuts 1e-9 like the batch goldens; float32 cuts: the reference's single-sample form keeps a float32
            # iterate for one iteration (tests/test_oracle_golden.py), 1.7e-7 from the batch algorithm: BASELINE's 1e-5
            tol = 1e-9 if prob.cut_dtype == np.float64 else (1e-4 if case == "lse_n33" else 1e-5)    # lse_n33 amplifies the 1e-7 tenfold per iteration (DUAL_Y_TOL)
            assert np.max(np.abs(x - gold[case + "__y"][row])) <= tol, (case, u)


def test_reference_tuple_types():
    from icnn_amd import bundle_entropy
    prob = problems.max_affine(1, 8, 21, 6)
    y0 = prob.y0()
    y, A, b, lam, xs, n_iters = bundle_entropy.solveBatch(prob.fg, y0, nIter=6)
    assert y is y0 and y.dtype == np.float64
    ora = oracle.solveBatch(prob.fg, prob.y0(), nIter=6)
    for u in range(8):
        assert len(A[u]) == len(b[u]) == len(xs[u]) == len(lam[u]) == len(ora[1][u])
        assert all(a.dtype == np.float32 and a.shape == (21,) for a in A[u])
        assert np.allclose(lam[u], ora[3][u], atol=1e-8)
    assert n_iters == ora[5]


def test_reference_tuple_at_headline_size_is_the_device_state():
    """BundleResult.as_reference_tuple at 4096 x 10 (what an unmodified multi-label-cls/icnn_ebundle.py:225-226 receives):
    icnn_be_export_active + ONE pinned copy + lists built on demand must hold exactly the device state -- every sample's
    active rows in bundle order, offsets, points, multipliers, nIters, and y."""
    from icnn_amd import bundle_entropy, picnn
    spec = picnn.bibtex_spec()
    B, n_iter = 4096, 10
    params, x = _picnn_problem(spec, B, 1000, "spread")
    model = picnn.FCModel(spec, params)
    ctx = model.context(torch.from_numpy(x))
    y0 = np.full((B, spec.n_labels), 0.5)
    res = bundle_entropy.solveBatch(f=model, ctx=ctx, y0=y0, nIter=n_iter, native=True)
    y, A, b, lam, xs, n_iters = res.as_reference_tuple()
    assert y is y0 and isinstance(A, list) and isinstance(lam, list) and len(A) == len(b) == len(lam) == len(xs) == B
    cnt = res.count[:B].cpu().numpy()
    act = res.active[:B].cpu().numpy()
    G, h, ys, lm = res.G.cpu().numpy(), res.h.cpu().numpy(), res.ys.cpu().numpy(), res.lam.cpu().numpy()
    assert np.array_equal(y, res.y.cpu().numpy()) and n_iters == res.n_iters[:B].cpu().numpy().tolist()
    assert [len(a) for a in A] == cnt.tolist()                                  # icnn_ebundle.py:235
    for u in range(B):
        k = cnt[u]
        assert len(b[u]) == len(xs[u]) == len(lam[u]) == k
        assert isinstance(A[u], list) and all(a.dtype == np.float32 and a.shape == (spec.n_labels,) for a in A[u])
        if k:
            assert np.array_equal(np.array(A[u]), G[u, act[u, :k]]) and np.array_equal(np.array(xs[u]), ys[u, act[u, :k]])
            assert np.array_equal(np.array(b[u]), h[u, act[u, :k]]) and np.array_equal(lam[u], lm[u, :k])
    # a second call while the first result is alive must not overwrite it (the pinned block is not shared)
    keep = [np.array(A[u]) for u in (0, B - 1)]
    res.as_reference_tuple()
    assert np.array_equal(keep[0], np.array(A[0])) and np.array_equal(keep[1], np.array(A[B - 1]))


def test_tensor_start_point_with_numpy_fg():
    """initXs as a torch tensor while fg is a NumPy callable: fg must see the start point at t = 0."""
    from icnn_amd import bundle_entropy
    prob = problems.log_sum_exp(3, 8, 17, 5)
    seen = []

    def fg(y):
        seen.append(y.copy())
        return prob.fg(y)

    res = bundle_entropy.solveBatch(fg, torch.full((8, 17), 0.5, dtype=torch.float64), nIter=4, native=True)
    assert np.array_equal(seen[0], np.full((8, 17), 0.5))
    ora = oracle.solve_batch(prob.fg, prob.y0(), 4)
    assert np.max(np.abs(res.y.cpu().numpy() - ora.y)) <= 1e-9


def test_empty_batch_returns_empty_lists():
    from icnn_amd import bundle_entropy
    y0 = np.zeros((0, 5))
    y, A, b, lam, xs, n_iters = bundle_entropy.solveBatch(lambda y: (np.zeros(0, np.float32), np.zeros((0, 5), np.float32)),
                                                          y0, nIter=3)
    assert y is y0 and A == [] and b == [] and lam == [] and xs == [] and n_iters == []


def test_float64_energy_with_float32_gradient_keeps_its_precision():
    """bi = fi - np.sum(gi * x) keeps fi's dtype (dual :143): an fg that returns float64 energies next to float32
    gradients must not have them rounded to float32 on the way to the device."""
    from icnn_amd import bundle_entropy
    base = problems.log_sum_exp(5, 8, 21, 6)

    def fg(y):
        f, g = base.fg(y)
        return f.astype(np.float64) + 1e-9 * np.arange(8), g          # bits below float32 resolution

    y0 = base.y0()
    res = bundle_entropy.solveBatch(fg, y0, nIter=5, native=True)
    ora = oracle.solve_batch(fg, base.y0(), 5)
    host = result_to_host(res)
    assert np.max(np.abs(host["h"][:, :5] - ora.h[:, :5])) <= 1e-13
    assert np.max(np.abs(host["y"] - ora.y)) <= 1e-9


def test_callback_protocol_and_in_place_iterates():
    from icnn_amd import bundle_entropy
    prob = problems.log_sum_exp(2, 8, 17, 5)
    seen = []
    y0 = prob.y0()
    bundle_entropy.solveBatch(prob.fg, y0, nIter=4,
                              callback=lambda t, f, y: seen.append((t, f.copy(), y.copy(), y is y0)))
    assert [s[0] for s in seen] == [0, 1, 2, 3]
    assert all(s[3] for s in seen), "callback must see the live initXs array"
    ref_seen = []
    oracle.solve_batch(prob.fg, prob.y0(), 4, callback=lambda t, f, y: ref_seen.append((f.copy(), y.copy())))
    for (t, f, y, _), (rf, ry) in zip(seen, ref_seen):
        assert np.allclose(f, rf, rtol=1e-6, atol=1e-6) and np.max(np.abs(y - ry)) < 1e-9
    seen = []
    bundle_entropy.solveBatch(prob.fg, prob.y0(), nIter=3, variant="rl", callback=lambda t, f: seen.append(t))
    assert seen == [0, 1, 2]


def test_cycle_shortcut_equals_full_newton_cap():
    """The limit-cycle shortcut must not change results beyond float64 jitter."""
    from icnn_amd import _lib
    factory, n_iter = problems.GOLDEN_CASES["action_box"]     # contains capped Newton solves
    _, fast = _solve(factory(), n_iter, "dual")
    _, full = _solve(factory(), n_iter, "dual", flags=_lib.FLAG_NO_CYCLE_SHORTCUT)
    a, b = result_to_host(fast), result_to_host(full)
    assert np.max(np.abs(a["y"] - b["y"])) < 1e-10
    assert a["active"] == b["active"]
    assert b["newton"].max() >= 100 and a["newton"].max() < b["newton"].max()


@pytest.mark.parametrize("B,n_iter", [(100, 10), (1100, 6), (16, 15), (1100, 12), (300, 14), (530, 24), (40, 31)])
def test_persistent_tile_kernel_equals_two_kernel_rounds(B, n_iter):
    """be_fused.hip runs the same device functions as the two-kernel rounds, 16 samples per workgroup for all
    rounds: every output must be bit-identical (partial last tile, more rounds than cuts, both included).  From
    nIter = 11 on the sixteen bundles of a tile do not fit the LDS together and the dual phase runs in groups sized by
    the cuts the samples hold; beyond 15 the 32-slot instance with the per-round update budget and the finishing
    launch (the two-kernel side then runs its time-sliced rounds)."""
    from icnn_amd import _lib, bundle_entropy, picnn
    spec = picnn.bibtex_spec()
    params = picnn.init_params(spec, 0, "spread")
    x = (np.random.RandomState(77).rand(B, spec.n_features) < 0.04).astype(np.float32)
    model = picnn.FCModel(spec, params)
    ctx = model.context(torch.from_numpy(x))
    outs = []
    for flags in (_lib.FLAG_PERSISTENT, _lib.FLAG_TWO_KERNELS):
        res = bundle_entropy.FusedSolver(model, B, n_iter, "dual", flags=flags).solve(ctx, 0.5)
        outs.append([t.cpu().numpy().copy() for t in (res.y, res.lam, res.active, res.count[:B], res.n_iters[:B],
                                                       res.newton_iters[:B], res.state.G, res.state.h, res.state.ys,
                                                       res.finished[:B], res.status[:B])])
    for a, b in zip(*outs):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("which,B,n_iter", [("bibtex", 100, 10), ("bibtex", 1, 7), ("bibtex", 256, 4), ("halfcheetah", 210, 5),
                                            ("halfcheetah", 1, 5), ("bibtex", 301, 6), ("bibtex", 70, 30), ("bibtex", 400, 22),
                                            ("halfcheetah", 333, 17), ("bibtex", 700, 5), ("bibtex", 1001, 4), ("halfcheetah", 1024, 5)])
def test_persistent_per_sample_kernel_equals_two_kernel_rounds(which, B, n_iter):
    """Batches of at most four samples per CU run a persistent workgroup per 1-4 samples by default
    (be_fused.hip, fused_rows_solve_kernel: VALU evaluation + the samples' dual steps, every sample at its own pace,
    early leavers free their CU; any nIter up to 31, no time slicing needed): every output bit-identical to one
    launch per phase and round (which time-slices for nIter > 15)."""
    from icnn_amd import _lib, bundle_entropy, picnn
    if which == "bibtex":
        spec, variant = picnn.bibtex_spec(), "dual"
        params = picnn.init_params(spec, 0, "spread")
        x = (np.random.RandomState(79).rand(max(B, 64), spec.n_features) < 0.04).astype(np.float32)
    else:
        spec, variant = picnn.halfcheetah_spec(), "rl"
        params = picnn.init_params(spec, 0, "spread", yu_bias=1.0, gate_bias=1.0)
        x = np.random.RandomState(80).randn(max(B, 64), spec.n_features).astype(np.float32)
    model = picnn.FCModel(spec, params)
    ctx = model.context(torch.from_numpy(x))[:B].contiguous()
    outs = []
    for flags in (0, _lib.FLAG_TWO_KERNELS):
        res = bundle_entropy.FusedSolver(model, B, n_iter, variant, flags=flags).solve(ctx, 0.5)
        outs.append([t.cpu().numpy().copy() for t in (res.y, res.lam, res.active, res.count[:B], res.n_iters[:B],
                                                       res.newton_iters[:B], res.state.G, res.state.h, res.state.ys,
                                                       res.finished[:B], res.status[:B])])
    for a, b in zip(*outs):
        assert np.array_equal(a, b)


def test_stragglers_of_time_sliced_rounds_finish_in_one_persistent_launch():
    """nIter > 15 on a batch beyond the per-sample kernel (more than four samples per CU): nIter time-sliced rounds
    (eight Newton updates per sample and round, then parked), after which the samples that are behind are finished by ONE
    launch of the persistent per-sample kernel in resume mode instead of whole rounds for the stragglers: every output
    bit-identical to the two-kernel rounds with their blind straggler rounds, and some samples really were behind."""
    from icnn_amd import _lib, bundle_entropy, picnn
    spec = picnn.bibtex_spec()
    params = picnn.init_params(spec, 0, "spread")
    B, n_iter = 1100, 20
    x = (np.random.RandomState(81).rand(B, spec.n_features) < 0.04).astype(np.float32)
    model = picnn.FCModel(spec, params)
    ctx = model.context(torch.from_numpy(x))
    outs = []
    for flags in (0, _lib.FLAG_TWO_KERNELS):
        res = bundle_entropy.FusedSolver(model, B, n_iter, "dual", flags=flags).solve(ctx, 0.5)
        outs.append([t.cpu().numpy().copy() for t in (res.y, res.lam, res.active, res.count[:B], res.n_iters[:B],
                                                       res.newton_iters[:B], res.state.G, res.state.h, res.state.ys,
                                                       res.finished[:B], res.status[:B])])
    for a, b in zip(*outs):
        assert np.array_equal(a, b)
    assert outs[0][5].max() > 8 * n_iter / 2          # a sample with that many updates was parked along the way


@pytest.mark.parametrize("B,n_iter", [(1100, 20), (700, 10), (4096, 30)])
def test_budgeted_tile_kernel_with_finishing_launch_equals_lockstep_tiles(B, n_iter):
    """The persistent per-tile kernel in its budgeted form (ICNN_BE_FLAG_PERSISTENT | ICNN_BE_FLAG_TIME_SLICE: eight Newton
    updates per sample and round, a parked sample skips phase A and resumes in its tile's next dual phase, ONE finishing launch
    of the per-sample kernel for the samples that are behind) against the default lockstep tiles: every output bit-identical,
    and the budget really parked something."""
    from icnn_amd import _lib, bundle_entropy, picnn
    spec = picnn.bibtex_spec()
    params, x = _picnn_problem(spec, B, 9, "spread")
    model = picnn.FCModel(spec, params)
    ctx = model.context(torch.from_numpy(x))
    outs, rounds = [], []
    for flags in (_lib.FLAG_PERSISTENT, _lib.FLAG_PERSISTENT | _lib.FLAG_TIME_SLICE):
        res = bundle_entropy.FusedSolver(model, B, n_iter, "dual", flags=flags).solve(ctx, 0.5)
        outs.append(_all_outputs(res, B))
        rounds.append(res.state.rounds)
    for i, (a, b) in enumerate(zip(*outs)):
        assert np.array_equal(a, b), "output %d differs" % i
    assert rounds == [n_iter, n_iter + 1], rounds


def test_wide_rows_more_iterations_than_lds_rows():
    """n = 2048 (the completion model's width): the staging area of a workgroup holds 12 cuts, fewer than the
    reference's default of 30 bundle iterations (completion/icnn_ebundle.py:41).  The iteration count is not limited by
    that -- only the ACTIVE bundle is staged --: a log-sum-exp of a few pieces keeps few cuts active and runs all 20
    iterations, within 1e-9 of the oracle; a convex quadratic with float64 cuts (half the capacity) grows its bundle past it, and
    those samples stop with ICNN_BE_ST_OVERFLOW, which the host reports as MemoryError (the reference has no such limit)."""
    from icnn_amd import _lib, bundle_entropy
    lib = _lib.load()
    cap = lib.icnn_be_bundle_capacity(2048, 20, _lib.CUT_F32, _lib.VARIANT["dual"])
    assert 8 <= cap < 20
    prob = problems.log_sum_exp(seed=12, B=6, n=2048, pieces=4, scale=0.05)
    y0 = prob.y0()
    res = bundle_entropy.solveBatch(prob.fg, y0, nIter=20, native=True)
    with np.errstate(all="ignore"):
        ora = oracle.solve_batch(prob.fg, prob.y0(), 20)
    assert int(res.count[:6].max().item()) < cap
    assert np.max(np.abs(res.y.cpu().numpy() - ora.y)) <= 1e-9
    # float64 cuts take twice the room: a convex quadratic that keeps eight cuts active outgrows the LDS.  Without the
    # staging area in device memory (a C caller that passes scratch = NULL) those samples stop with OVERFLOW ...
    many = problems.quadratic(seed=5, B=3, n=2048)
    cap64 = lib.icnn_be_bundle_capacity(2048, 25, _lib.CUT_F64, _lib.VARIANT["dual"])
    assert 2 <= cap64 < 8
    bundle_entropy.ALLOW_SCRATCH = False
    try:
        with pytest.raises(MemoryError):
            bundle_entropy.solveBatch(many.fg, many.y0(), nIter=25, native=True)
        res = bundle_entropy.solveBatch(many.fg, many.y0(), nIter=25, native=True, check=False)
    finally:
        bundle_entropy.ALLOW_SCRATCH = True
    st = res.status[:3].cpu().numpy()
    assert (st & _lib.ST_OVERFLOW).all() and np.isfinite(res.y.cpu().numpy()).all()
    assert int(res.count[:3].max().item()) <= cap64
    # ... with it (the default of the Python host) the same solve runs all its iterations (this quadratic is one of the
    # problems whose iterates are not comparable between two float64 implementations, so no oracle comparison here)
    res = bundle_entropy.solveBatch(many.fg, many.y0(), nIter=25, native=True)
    assert not res.status[:3].cpu().numpy().any() and (res.n_iters[:3].cpu().numpy() == 25).all()
    # float64 cuts through the staging area in device memory: forced for every round on a solve that fits LDS, same bits
    p64 = problems.log_sum_exp(seed=12, B=4, n=2048, pieces=4, scale=0.05, cut_dtype=np.float64)
    a = bundle_entropy.solveBatch(p64.fg, p64.y0(), nIter=16, native=True)
    b = bundle_entropy.solveBatch(p64.fg, p64.y0(), nIter=16, native=True, flags=_lib.FLAG_GLOBAL_BUNDLE)
    assert torch.equal(a.y, b.y) and torch.equal(a.lam, b.lam) and torch.equal(a.active, b.active)
    with np.errstate(all="ignore"):
        ora = oracle.solve_batch(p64.fg, p64.y0(), 16)
    assert np.max(np.abs(b.y.cpu().numpy() - ora.y)) <= 1e-6       # (a smooth log-sum-exp: the ill-conditioned class)


@pytest.mark.parametrize("kind,n,n_iter", [("max_affine", 1040, 12), ("max_affine", 2000, 20), ("many_pieces", 1536, 24), ("many_pieces", 1040, 31)])
def test_wide_rows_of_other_widths_match_oracle(kind, n, n_iter):
    """The eight-wave dual step with its fused VALU pass on a generic fg (float32 cuts through icnn_be_dual_step) at widths other
    than the completion model's 2048: widths that are no multiple of 512 (columns beyond n masked), piecewise-linear energies
    with few pieces (repeated cuts: the rank test ends the solve) and with sixty (every cut stays active: a bundle of t cuts in
    round t) -- at n = 1536 past the LDS capacity (split staging in dual_step_wide_kernel, 24 cuts: the pass up to 20, then the
    plain device-memory body with the MFMA sweep), at n = 1040 up to 31 cuts in LDS.  Against the NumPy oracle on the same cuts:
    identical iteration counts and active sets, y* to 1e-9; and bit-identical to forced device-memory staging."""
    from icnn_amd import _lib, bundle_entropy
    prob = problems.max_affine(31 + n, 6 if kind == "max_affine" else 3, n, 9 if kind == "max_affine" else 60, 1.0)
    res = bundle_entropy.solveBatch(prob.fg, prob.y0(), nIter=n_iter, native=True)
    with np.errstate(all="ignore"):
        ora = oracle.solve_batch(prob.fg, prob.y0(), n_iter)
    host = result_to_host(res)
    dy, discrete = compare_with_oracle(host, ora)
    print("%s n=%d nIter=%d: cuts max %d, max|dy| %.2e, discrete %s" % (kind, n, n_iter, max(len(a) for a in host["active"]), dy.max(), discrete))
    assert (host["status"] == 0).all()
    assert not discrete, discrete
    assert dy.max() <= 1e-9, dy.max()
    if kind == "many_pieces":
        assert max(len(a) for a in host["active"]) >= 24
    forced = bundle_entropy.solveBatch(prob.fg, prob.y0(), nIter=n_iter, native=True, flags=_lib.FLAG_GLOBAL_BUNDLE)
    assert torch.equal(res.y, forced.y) and torch.equal(res.count, forced.count) and torch.equal(res.lam, forced.lam)


@pytest.mark.parametrize("B,n_iter", [(100, 10), (1100, 10), (300, 14)])
def test_one_wave_valu_contraction_agrees_with_mfma_sweep(B, n_iter):
    """Round 4: one-wave samples (float32 rows of up to 192 columns) form H = A diag(w) A^T | A z of bundles of up to 8 cuts by
    the fused VALU pass, larger bundles by the float64 MFMA sweep; ICNN_BE_FLAG_MFMA_CONTRACTION keeps the sweep throughout.
    Same sums in another order: y* agrees to float64 solver noise with identical discrete outcomes, on the per-sample kernel
    (B = 100, 300) and the per-tile kernel (B = 1100); and in EITHER mode the persistent kernels are bit-identical to launch
    pairs (every kernel takes the same decision)."""
    from icnn_amd import _lib, bundle_entropy, picnn
    spec = picnn.bibtex_spec()
    params, x = _picnn_problem(spec, max(B, 64), 4, "spread")
    model = picnn.FCModel(spec, params)
    ctx = model.context(torch.from_numpy(x))[:B].contiguous()
    outs = {}
    for mode in (0, _lib.FLAG_MFMA_CONTRACTION):
        for path in (0, _lib.FLAG_TWO_KERNELS):
            res = bundle_entropy.FusedSolver(model, B, n_iter, "dual", flags=mode | path).solve(ctx, 0.5)
            outs[(mode, path)] = _all_outputs(res, B)
        for i, (a, b) in enumerate(zip(outs[(mode, 0)], outs[(mode, _lib.FLAG_TWO_KERNELS)])):
            assert np.array_equal(a, b), "mode %d: output %d differs between the dispatch paths" % (mode, i)
    v, m = outs[(0, 0)], outs[(_lib.FLAG_MFMA_CONTRACTION, 0)]
    assert not np.array_equal(v[0], m[0]), "the two contractions are expected to differ in the last bits"
    assert np.max(np.abs(v[0] - m[0])) <= 1e-9                     # y*
    for i in (2, 3, 4, 9, 10):                                     # active slots, counts, nIters, finished, status
        assert np.array_equal(v[i], m[i]), "discrete output %d differs between the contractions" % i


@pytest.mark.parametrize("B,n_iter,seed", [(19, 5, 2), (5, 9, 4), (4, 20, 6)])
def test_fused_valu_contraction_agrees_with_mfma_sweep(B, n_iter, seed):
    """Wide rows (eight waves per sample), bundles of up to 7 cuts: H = A diag(w) A^T and A z are formed in the column pass
    itself (be_dual_valu_dev.h: lane = column, transposing wave butterfly, one barrier per Newton update);
    ICNN_BE_FLAG_MFMA_CONTRACTION keeps the float64-MFMA sweep.  Same sums, another order: active sets and iteration counts
    are identical, y* agrees to rounding -- completion model at nIter 5 (bundles of up to 6 cuts: the inlined instances), 9 (the
    instances of 8 and more cuts, functions) and 20 (from round 12 on dual_step_wide_kernel: device-memory staging with the 12
    oldest rows mirrored in LDS; with the flag the plain device-memory kernel and its MFMA sweep)."""
    from icnn_amd import _lib, bundle_entropy, picnn
    spec, params, x = _conv_problem(B, seed, "spread")
    model = picnn.ConvModel(spec, params)
    ctx = model.context(torch.from_numpy(x))
    y0 = torch.from_numpy(np.repeat((0.2 + 0.6 * np.random.RandomState(9).rand(spec.n_labels))[None], B, axis=0)).cuda()
    outs = []
    for flags in (0, _lib.FLAG_MFMA_CONTRACTION):
        res = bundle_entropy.FusedSolver(model, B, n_iter, "dual", flags=flags).solve(ctx, y0)
        outs.append([t.cpu().numpy().copy() for t in (res.y, res.count[:B], res.n_iters[:B], res.active, res.lam)])
    (ya, ca, na, aa, la), (yb, cb, nb, ab_, lb) = outs
    assert np.array_equal(ca, cb) and np.array_equal(na, nb)
    for u in range(B):
        assert np.array_equal(aa[u, :ca[u]], ab_[u, :cb[u]])
    assert np.abs(ya - yb).max() <= 1e-9, np.abs(ya - yb).max()
    assert not np.array_equal(ya, yb)                          # (the flag really switches the contraction)


def test_wide_rows_bundle_staged_in_device_memory():
    """n = 2048: from the round on in which a bundle could outgrow the 12 cuts the LDS holds, the dual step stages it in
    st.scratch (device memory) -- same kernel code, the sweeps at L2 latency.  (a) Forced for every round
    (ICNN_BE_FLAG_GLOBAL_BUNDLE) on a solve that also fits LDS: every output bit-identical to the LDS staging.  (b) The
    completion model at 31 bundle iterations (reference default: 30): bundles grow to 17 cuts, nothing overflows, the
    active sets equal the oracle's (run on the kernel-order PICNN: identical cuts on both sides) and y* agrees to 1e-9."""
    from icnn_amd import _lib, bundle_entropy, picnn
    from oracle import picnn_conv_oracle as co
    spec, params, x = _conv_problem(6, 3, "spread")
    model = picnn.ConvModel(spec, params)
    ctx = model.context(torch.from_numpy(x))
    y0 = np.repeat((0.2 + 0.6 * np.random.RandomState(9).rand(spec.n_labels))[None], 6, axis=0)
    outs = []
    for flags in (0, _lib.FLAG_GLOBAL_BUNDLE):
        # (16 iterations: both runs use the 32-slot kernel, whose sweep sums in another order than the 16-slot one's;
        #  without the flag rounds 0-11 stage in LDS, with it every round stages in device memory)
        res = bundle_entropy.FusedSolver(model, 6, 16, "dual", flags=flags).solve(ctx, torch.from_numpy(y0).cuda())
        outs.append([t.cpu().numpy().copy() for t in (res.y, res.lam, res.active, res.count[:6], res.n_iters[:6],
                                                       res.newton_iters[:6], res.state.G, res.state.h, res.finished[:6])])
    for a, b in zip(*outs):
        assert np.array_equal(a, b)
    # (b) the completion model at the largest iteration count (reference default: 30, completion/icnn_ebundle.py:41)
    B, n_iter = 4, 31
    spec, params, x = _conv_problem(B, 5, "spread")
    model = picnn.ConvModel(spec, params)
    ctx = model.context(torch.from_numpy(x))
    solver = bundle_entropy.FusedSolver(model, B, n_iter, "dual")
    assert solver.state.scratch is not None
    res = solver.solve(ctx, torch.from_numpy(y0[:B]).cuda())
    cnt = res.count[:B].cpu().numpy()
    assert not (res.status[:B].cpu().numpy() & _lib.ST_OVERFLOW).any()
    cap = _lib.load().icnn_be_bundle_capacity(spec.n_labels, n_iter, _lib.CUT_F32, _lib.VARIANT["dual"])
    assert cnt.max() + 1 > cap, (cnt, cap)                  # the solve really needed the staging area
    fg = co.make_fg_chain(params, ctx.cpu().numpy(), spec.H, spec.W)
    with np.errstate(all="ignore"):
        ora = oracle.solve_batch(fg, y0[:B].copy(), n_iter)
    assert np.array_equal(cnt, np.array([len(l) for l in ora.lam]))
    assert np.max(np.abs(res.y.cpu().numpy() - ora.y)) <= 1e-9
    # (c) the interior-point variant -- the module the completion script imports (lib/bundle_entropy.py) -- stages fewer
    # cuts in LDS still (three more column buffers) and takes the same route
    n_iter = 20
    cap_ipm = _lib.load().icnn_be_bundle_capacity(spec.n_labels, n_iter, _lib.CUT_F32, _lib.VARIANT["pdipm"])
    res = bundle_entropy.FusedSolver(model, B, n_iter, "pdipm").solve(ctx, torch.from_numpy(y0[:B]).cuda())
    with np.errstate(all="ignore"):
        ora = oracle.solve_batch(fg, y0[:B].copy(), n_iter, variant="pdipm")
    cnt = res.count[:B].cpu().numpy()
    assert not res.status[:B].cpu().numpy().any() and cnt.max() + 1 > cap_ipm, (cnt, cap_ipm)
    assert np.array_equal(cnt, np.array([len(l) for l in ora.lam]))
    assert np.max(np.abs(res.y.cpu().numpy() - ora.y)) <= 1e-7


def test_persistent_tile_kernel_rl_variant_equals_two_kernel_rounds():
    """The RL variant (clip, Armijo search, early stop, no rank test) through the persistent kernel when forced
    (by default it keeps the two-kernel rounds, which measured faster): bit-identical outputs."""
    from icnn_amd import _lib, bundle_entropy, picnn
    spec = picnn.halfcheetah_spec()
    params = picnn.init_params(spec, 0, "spread", yu_bias=1.0, gate_bias=1.0)
    B, n_iter = 210, 5
    x = np.random.RandomState(78).randn(B, spec.n_features).astype(np.float32)
    model = picnn.FCModel(spec, params)
    ctx = model.context(torch.from_numpy(x))
    outs = []
    for flags in (_lib.FLAG_PERSISTENT, _lib.FLAG_TWO_KERNELS):
        res = bundle_entropy.FusedSolver(model, B, n_iter, "rl", flags=flags).solve(ctx, 0.5)
        outs.append([t.cpu().numpy().copy() for t in (res.y, res.lam, res.active, res.count[:B], res.n_iters[:B],
                                                       res.finished[:B], res.status[:B])])
    for a, b in zip(*outs):
        assert np.array_equal(a, b)


def test_period3_cycle_shortcut_on_benchmark_batch():
    """About one Newton solve in 40 000 of the Bibsonomy-shaped workload ends in a 3-cycle; the benchmark
    batch (seed 1000) contains one.  With the shortcut the solve stops after ~30 updates instead of 100 and
    the result stays within float64 jitter of running the full cap (which the reference always does)."""
    from icnn_amd import _lib, bundle_entropy, picnn
    spec = picnn.bibtex_spec()
    params = picnn.init_params(spec, 0, "spread")
    B, n_iter = 4096, 10
    x = (np.random.RandomState(1000).rand(B, spec.n_features) < 0.04).astype(np.float32)
    model = picnn.FCModel(spec, params)
    ctx = model.context(torch.from_numpy(x))
    out = {}
    for name, flags in (("fast", 0), ("full", _lib.FLAG_NO_CYCLE_SHORTCUT)):
        res = bundle_entropy.FusedSolver(model, B, n_iter, "dual", flags=flags).solve(ctx, 0.5)
        out[name] = (res.y.cpu().numpy().copy(), res.newton_iters[:B].cpu().numpy().copy(),
                     res.count[:B].cpu().numpy().copy(), res.active.cpu().numpy().copy())
    (yf, nf, cf, af), (yr, nr, cr, ar) = out["fast"], out["full"]
    assert np.max(np.abs(yf - yr)) < 1e-8
    assert (cf == cr).all() and (af == ar).all()
    # the full run has solves that hit the 100-update cap every outer iteration they cycle in; the shortcut
    # never needs anywhere near that in total
    saved = nr.astype(np.int64) - nf
    assert saved.max() >= 60 and (saved >= 0).all()


def _picnn_problem(spec, B, seed, regime, **init_kw):
    from icnn_amd import picnn
    params = picnn.init_params(spec, seed, regime, **init_kw)
    rng = np.random.RandomState(seed + 100)
    if spec.n_features > 100:
        x = (rng.rand(B, spec.n_features) < 0.04).astype(np.float32)   # sparse binary, like BibTeX
    else:
        x = rng.randn(B, spec.n_features).astype(np.float32)
    return params, x


@pytest.mark.parametrize("which,B", [("bibtex", 100), ("bibtex", 16), ("halfcheetah", 257)])
def test_fc_energy_and_gradient(which, B):
    from icnn_amd import picnn
    spec = picnn.bibtex_spec() if which == "bibtex" else picnn.halfcheetah_spec()
    kw = {} if which == "bibtex" else dict(yu_bias=1.0, gate_bias=1.0)
    params, x = _picnn_problem(spec, B, 0, "spread", **kw)
    model = picnn.FCModel(spec, params)
    ctx = model.context(torch.from_numpy(x))
    fg = picnn_oracle.make_fg(params, x, list(spec.szs), spec.alpha, spec.batchnorm,
                              "action" if spec.action_box else None)
    ctx_ref = picnn_oracle.flat_context(fg.ctx)
    scale = np.abs(ctx_ref).max()
    assert np.max(np.abs(ctx.cpu().numpy() - ctx_ref)) <= 2e-5 * scale, "x-only context"
    rng = np.random.RandomState(5)
    y = rng.rand(B, spec.n_labels)
    # same context on both sides so that only the y-path kernel is compared
    ctx_dev = torch.from_numpy(ctx_ref).cuda()
    f, g = model.fg(ctx_dev, torch.from_numpy(y).cuda())
    f_ref, g_ref = fg(y)
    gs = np.abs(g_ref).max()
    assert np.max(np.abs(g.cpu().numpy() - g_ref)) <= 2e-5 * gs
    assert np.max(np.abs(f.cpu().numpy() - f_ref)) <= 2e-5 * max(1.0, np.abs(f_ref).max())


@pytest.mark.parametrize("which,B", [("bibtex", 4096), ("bibtex", 77), ("halfcheetah", 257), ("halfcheetah", 1),
                                     ("small3", 130)])
def test_context_kernels_match_oracle(which, B):
    """x-only context producer on the device (be_context.hip: one MFMA GEMM per stage with routed epilogue,
    batch-statistics BatchNorm) against oracle/picnn_oracle.context, the NumPy restatement of
    multi-label-cls/icnn_ebundle.py:339-374 / RL/src/icnn.py:339-385.  float32 tolerance 2e-5 of the context's scale;
    batch sizes off the 64-row tile, K off the 16-deep k-block (159, 17), three hidden layers."""
    from icnn_amd import picnn
    if which == "bibtex":
        spec, kw = picnn.bibtex_spec(), {}
    elif which == "halfcheetah":
        spec, kw = picnn.halfcheetah_spec(), dict(yu_bias=1.0, gate_bias=1.0)
    else:
        spec, kw = picnn.FCSpec(45, 11, (70, 33, 18)), {}
    params, x = _picnn_problem(spec, B, 4, "spread", **kw)
    model = picnn.FCModel(spec, params)
    ctx = model.context(torch.from_numpy(x)).cpu().numpy()
    if B == 1 and spec.batchnorm:
        pytest.skip("batch statistics of one row")
    ref = picnn_oracle.flat_context(picnn_oracle.context(params, x, list(spec.szs), spec.batchnorm))
    assert ctx.shape == ref.shape == (B, spec.ctx_width)
    scale = np.abs(ref).max()
    err = np.max(np.abs(ctx - ref))
    print("%s B=%d: max|ctx - ref| = %.2e of scale %.2e" % (which, B, err, scale))
    assert err <= 2e-5 * scale
    # the torch statement used on the host side (CPU tests, sharding) agrees too
    host = picnn.context(spec, params, torch.from_numpy(x)).numpy()
    assert np.max(np.abs(ctx - host)) <= 2e-5 * scale


def test_repack_then_context_and_adam_use_the_new_parameters():
    """model.repack(params) -- the per-update call of INTEGRATION.md -- followed by model.context(x) and by the one-launch
    act() path (rl_adam, icnn_be_adam_fc_obs reads struct icnn_be_fc_ctx): both must work and see the new parameters."""
    import dataclasses
    from icnn_amd import picnn, rl_adam
    spec = dataclasses.replace(picnn.halfcheetah_spec(), action_box=False)
    p0 = picnn.init_params(spec, 0, "spread", yu_bias=1.0, gate_bias=1.0)
    p1 = picnn.init_params(spec, 5, "spread", yu_bias=1.0, gate_bias=1.0)
    x = np.random.RandomState(3).randn(4, spec.n_features).astype(np.float32)
    model = picnn.FCModel(spec, p0)
    c0 = model.context(torch.from_numpy(x)).cpu().numpy()
    model.repack(p1)
    c1 = model.context(torch.from_numpy(x)).cpu().numpy()
    fresh = picnn.FCModel(spec, p1)
    assert np.array_equal(c1, fresh.context(torch.from_numpy(x)).cpu().numpy())
    assert not np.array_equal(c0, c1)
    a = rl_adam.adam(model, torch.from_numpy(x), one_launch=True).cpu().numpy()
    b = rl_adam.adam(fresh, torch.from_numpy(x), one_launch=True).cpu().numpy()
    assert np.array_equal(a, b)


@pytest.mark.parametrize("B,split", [(4096, 512), (77, 40), (64, 0)])
def test_sharded_context_uses_global_batchnorm_statistics(B, split):
    """Data-parallel ranks: every rank computes the context rows of ITS shard only, the u-path BatchNorm
    statistics of the global batch come from one all-reduce of 2 x 600 float64 sums (icnn_be_fc_context_stage / _norm,
    FCModel.context_sharded).  Two shards of one minibatch are run one after the other here with an all-reduce that adds
    the other shard's sums (the Bibsonomy network normalises stage 0 only, whose sums depend on x alone); a world of one
    rank (split = 0) runs the same code.  Rows must match the single-device context of the whole batch to float32 rounding
    (the variance is E[u^2] - mean^2 in float64 here, two float32 passes there)."""
    import ctypes as C
    from icnn_amd import picnn
    spec = picnn.bibtex_spec()
    params, x = _picnn_problem(spec, B, 2, "spread")
    model = picnn.FCModel(spec, params)
    xd = torch.from_numpy(x).cuda()
    full = model.context(xd).cpu().numpy()
    scale = np.abs(full).max()
    if split == 0:
        got = model.context_sharded(xd).cpu().numpy()
        assert np.max(np.abs(got - full)) <= 2e-5 * scale
        return
    shards = [xd[:split].contiguous(), xd[split:].contiguous()]

    def stage0_sums(xs):
        w = spec.widths[0]
        ctx = torch.empty(xs.shape[0], spec.ctx_width, dtype=torch.float32, device="cuda")
        work = torch.empty(int(model._lib.icnn_be_fc_context_work_floats(C.byref(model.c_ctx), xs.shape[0])), dtype=torch.float32,
                           device="cuda")
        stats = torch.zeros(2 * w, dtype=torch.float64, device="cuda")
        rc = model._lib.icnn_be_fc_context_stage(C.byref(model.c_ctx), 0, xs.data_ptr(), xs.shape[0], ctx.data_ptr(),
                                                 spec.ctx_width, work.data_ptr(), stats.data_ptr(), None)
        assert rc == 1
        torch.cuda.synchronize()
        return stats

    sums = [stage0_sums(s) for s in shards]
    rows = []
    for me, xs in enumerate(shards):
        calls = []

        def all_reduce(t, other=sums[1 - me], calls=calls):
            assert t.numel() == other.numel()
            calls.append(t.numel())
            t += other

        rows.append(model.context_sharded(xs, batch_total=float(B), all_reduce=all_reduce).cpu().numpy())
        assert calls == [2 * spec.widths[0]], "ONE all-reduce of 2 x 600 doubles"
    got = np.concatenate(rows)
    err = np.max(np.abs(got - full))
    print("sharded context B=%d split %d: max |d| = %.2e of scale %.2e" % (B, split, err, scale))
    assert err <= 2e-5 * scale
    alone = model.context(shards[0]).cpu().numpy()            # (statistics of the shard alone give different rows: the test bites)
    assert np.max(np.abs(alone - full[:split])) > 1e-3 * scale


@pytest.mark.parametrize("mode", ["makeCvx", "proj"])
def test_weight_clamps_on_the_device(mode):
    """makeCvx / proj (multi-label-cls/icnn_ebundle.py:143-144) applied to the device-resident packed weights equal
    packing the clamped weights on the host; the unconstrained 'yu' operands are untouched."""
    from icnn_amd import picnn
    spec = picnn.FCSpec(30, 9, (40, 24))
    params = picnn.init_params(spec, 5, "spread")
    rng = np.random.RandomState(9)
    for k in params:                                  # signs on the 'proj' weights so that the clamp has work to do
        if "proj" in k:
            params[k] = (params[k] * np.sign(rng.randn(*params[k].shape))).astype(np.float32)
    model = picnn.FCModel(spec, dict(params))
    model.clamp(mode)
    want = picnn.make_convex(dict(params)) if mode == "makeCvx" else picnn.project(dict(params))
    ref = picnn.FCModel(spec, want)
    assert torch.equal(model.wpack, ref.wpack)
    y = torch.from_numpy(rng.rand(12, spec.n_labels)).cuda()
    ctx = ref.context(torch.from_numpy(rng.randn(12, 30).astype(np.float32)))
    f1, g1 = model.fg(ctx, y)
    f2, g2 = ref.fg(ctx, y)
    assert torch.equal(f1, f2) and torch.equal(g1, g2)


@pytest.mark.parametrize("regime,B,n_iter", [("spread", 128, 10), ("init", 128, 10), ("spread", 64, 30)])
def test_fused_bibtex_matches_oracle(regime, B, n_iter):
    """BASELINE.json configs[1]: Bibsonomy PICNN, y-dim 159, batch 128, nIter 10."""
    from icnn_amd import bundle_entropy, picnn
    spec = picnn.bibtex_spec()
    params, x = _picnn_problem(spec, B, 0, regime)
    model = picnn.FCModel(spec, params)
    fg = picnn_oracle.make_fg(params, x, list(spec.szs))
    ctx = torch.from_numpy(picnn_oracle.flat_context(fg.ctx)).cuda()
    y0 = np.full((B, spec.n_labels), 0.5)
    res = bundle_entropy.solveBatch(f=model, ctx=ctx, y0=y0, nIter=n_iter, native=True)
    assert res.y.shape == (B, spec.n_labels)
    with np.errstate(all="ignore"):
        ora = oracle.solve_batch(fg, np.full((B, spec.n_labels), 0.5), n_iter)
    host = result_to_host(res)
    dy, discrete = compare_with_oracle(host, ora)
    print("fused %s B=%d nIter=%d: max|dy|=%.3e median %.1e, %d/%d above 1e-5, %d different discrete outcomes"
          % (regime, B, n_iter, dy.max(), np.median(dy), int((dy > 1e-5).sum()), B, len(discrete)))
    assert np.array_equal(host["y"], y0), "y0 must be updated in place"
    # Against an oracle whose float32 PICNN sums in a different order (NumPy sgemm vs the MFMA chain) "within 1e-5
    # on every sample" is not attainable by ANY implementation; what is asserted is that the HIP path's tail is no
    # heavier than the oracle's own tail between the two orders (test_fused_tail_is_inside_the_oracles_own_band;
    # test_fused_matches_chain_order_oracle is the bit-tight check).
    if regime == "init":
        assert dy.max() <= 1e-5
    else:
        # the band, asserted (this branch used to print only): the same oracle solver with the PICNN in the
        # kernel's summation order gives the distribution of |y*(chain) - y*(sgemm)| on THESE inputs; the HIP path must
        # have no heavier tail against the sgemm-order oracle, and must sit on the chain-order oracle to solver noise
        from sensitivity_util import assert_inside_band, per_sample
        fg_chain = picnn_oracle.make_fg_chain(params, ctx.cpu().numpy(), list(spec.szs))
        with np.errstate(all="ignore"):
            ora_chain = oracle.solve_batch(fg_chain, np.full((B, spec.n_labels), 0.5), n_iter)
        assert_inside_band(dy, per_sample(ora_chain.y, ora.y), B, "%s B=%d nIter=%d:" % (regime, B, n_iter))
        assert per_sample(host["y"], ora_chain.y).max() <= 1e-7


@pytest.mark.parametrize("B,n_iter", [(128, 10), (64, 30), (4096, 10)])
def test_fused_tail_is_inside_the_oracles_own_band(B, n_iter):
    """Tier B, without free parameters: |y_hip - y_oracle(sgemm order)| per sample must have no heavier tail (median,
    p90, share above 1e-5, max) than |y_oracle(chain order) - y_oracle(sgemm order)|, the oracle against itself under
    two equally valid float32 summation orders of the PICNN (tests/test_sensitivity.py measures that band on CPU).
    B = 128 / nIter = 10 is BASELINE.json configs[1], B = 4096 the headline batch, nIter = 30 the shape of configs[3]."""
    from icnn_amd import bundle_entropy, picnn
    from sensitivity_util import assert_inside_band, bibtex_problem, oracle_pair, per_sample
    spec, params, ctx = bibtex_problem(B)
    model = picnn.FCModel(spec, params)
    y0 = np.full((B, spec.n_labels), 0.5)
    res = bundle_entropy.solveBatch(f=model, ctx=torch.from_numpy(ctx).cuda(), y0=y0, nIter=n_iter, native=True)
    ora_sgemm, ora_chain = oracle_pair(spec, params, ctx, n_iter)
    y_hip = res.y.cpu().numpy()
    t, b = assert_inside_band(per_sample(y_hip, ora_sgemm.y), per_sample(ora_chain.y, ora_sgemm.y), B,
                              "B=%d nIter=%d:" % (B, n_iter))
    print("B=%d nIter=%d  HIP vs sgemm-order oracle %s\n                 oracle vs oracle          %s" % (B, n_iter, t, b))
    # and against the order-matched oracle the HIP path is tight on every sample
    assert per_sample(y_hip, ora_chain.y).max() <= 1e-7


@pytest.mark.parametrize("which,B", [("bibtex", 100), ("halfcheetah", 257)])
def test_fc_energy_and_gradient_bit_exact_vs_mfma_order_oracle(which, B):
    """oracle/picnn_chain.c evaluates the same float32 network in the MFMA's accumulation order:
    the kernel must agree with it bit for bit (E and dE/dy), not just to rounding."""
    from icnn_amd import picnn
    spec = picnn.bibtex_spec() if which == "bibtex" else picnn.halfcheetah_spec()
    kw = {} if which == "bibtex" else dict(yu_bias=1.0, gate_bias=1.0)
    params, x = _picnn_problem(spec, B, 3, "spread", **kw)
    model = picnn.FCModel(spec, params)
    ctx = model.context(torch.from_numpy(x))
    y = np.random.RandomState(7).rand(B, spec.n_labels)
    f, g = model.fg(ctx, torch.from_numpy(y).cuda())
    f_ref, g_ref = picnn_oracle.energy_and_grad_chain(params, ctx.cpu().numpy(), y, list(spec.szs), spec.alpha,
                                                      spec.action_box)
    assert np.array_equal(f.cpu().numpy(), f_ref), np.abs(f.cpu().numpy() - f_ref).max()
    assert np.array_equal(g.cpu().numpy(), g_ref), np.abs(g.cpu().numpy() - g_ref).max()


@pytest.mark.parametrize("B", [64, 256])
def test_kernel_rounding_error_against_float64_truth(B):
    """the kernels' float32 E and dE/dy are compared with the float64 evaluation of the same
    float32-parameter network, next to NumPy's sgemm-order float32 evaluation of it.  The MFMA chain order must not be a
    worse instance of "the reference's float32 fg": root-mean-square error within 1.5 x the sgemm order's, worst element
    within 2 x (measured 1.25 x / 1.45 x); both the MFMA tile kernel (B = 256 > one sample per CU... forced below) and the
    one-sample-per-CU VALU path (B = 64) are covered -- they agree bit for bit anyway."""
    from icnn_amd import picnn
    from sensitivity_util import bibtex_problem, error_stats, rounding_errors
    spec, params, ctx = bibtex_problem(B)
    model = picnn.FCModel(spec, params)
    y = np.random.RandomState(3).rand(B, spec.n_labels)
    err, (E64, g64) = rounding_errors(spec, params, ctx, y)
    ctx_dev = torch.from_numpy(ctx).cuda()
    y_dev = torch.from_numpy(y).cuda()
    f, g = model.fg(ctx_dev, y_dev)
    mine = error_stats(f.cpu().numpy(), g.cpu().numpy(), E64, g64)
    print("kernel %s | sgemm %s | chain %s | pairwise %s" % (mine, err["sgemm"], err["chain"], err["pairwise"]))
    for j in (0, 2):
        assert mine[j] <= 1.5 * err["sgemm"][j], (j, mine, err)
    for j in (1, 3):
        assert mine[j] <= 2.0 * err["sgemm"][j], (j, mine, err)
    assert mine == err["chain"]                             # (the kernel IS the chain order)


@pytest.mark.parametrize("szs,n,alpha,B", [((40,), 9, 0.0, 21), ((70, 33, 18, 50), 20, 0.01, 35), ((300, 280), 270, 0.0, 17),
                                          ((16,), 1, 0.0, 3), ((70, 33, 18, 50), 20, 0.01, 600), ((72, 40), 70, 0.0, 530),
                                          ((240, 100), 33, 0.0, 700), ((300, 280), 270, 0.0, 513)])
def test_fc_energy_and_gradient_bit_exact_other_shapes(szs, n, alpha, B):
    """Layer counts and widths off the two reference networks: a single hidden layer (the forward epilogue that
    writes delta and the backward phase that starts dE/dy and computes the energy are then the same layer), four
    hidden layers, dim(y) > 256 (no wave without a dE/dy tile: energies after the tiles), dim(y) = 1; partial tiles.
    Batches of up to two samples per CU take the per-sample VALU kernel, the larger ones the 16-row MFMA tiles -- among them
    widths whose k-blocks are an odd multiple of the pack's padding (70, 72 -> 5; 240 -> 15): the tile loops run their last
    k-block behind the ring loop (gemm_tiles, be_picnn_fc_dev.h)."""
    from icnn_amd import picnn
    spec = picnn.FCSpec(12, n, tuple(szs), alpha=alpha, batchnorm=False)
    params = picnn.init_params(spec, 5, "spread", yu_bias=1.0, gate_bias=1.0)
    x = np.random.RandomState(8).randn(B, 12).astype(np.float32)
    model = picnn.FCModel(spec, params)
    ctx = model.context(torch.from_numpy(x))
    y = np.random.RandomState(9).rand(B, n)
    f, g = model.fg(ctx, torch.from_numpy(y).cuda())
    f_ref, g_ref = picnn_oracle.energy_and_grad_chain(params, ctx.cpu().numpy(), y, list(spec.szs), spec.alpha, False)
    assert np.array_equal(f.cpu().numpy(), f_ref), np.abs(f.cpu().numpy() - f_ref).max()
    assert np.array_equal(g.cpu().numpy(), g_ref), np.abs(g.cpu().numpy() - g_ref).max()


@pytest.mark.parametrize("regime,B,n_iter", [("spread", 128, 10), ("spread", 64, 30), ("init", 96, 10)])
def test_fused_matches_chain_order_oracle(regime, B, n_iter):
    """The bit-tight fused check (BASELINE.json configs[1] and the nIter=30 shape of configs[3]):
    with the oracle's float32 PICNN summing in the kernel's order, both sides see identical cuts and
    y* must agree to float64 solver noise -- far inside BASELINE.json's 1e-5 -- with identical
    active sets and nIters for every sample."""
    from icnn_amd import bundle_entropy, picnn
    spec = picnn.bibtex_spec()
    params, x = _picnn_problem(spec, B, 0, regime)
    model = picnn.FCModel(spec, params)
    ctx = model.context(torch.from_numpy(x))
    y0 = np.full((B, spec.n_labels), 0.5)
    res = bundle_entropy.solveBatch(f=model, ctx=ctx, y0=y0, nIter=n_iter, native=True)
    fg = picnn_oracle.make_fg_chain(params, ctx.cpu().numpy(), list(spec.szs))
    with np.errstate(all="ignore"):
        ora = oracle.solve_batch(fg, np.full((B, spec.n_labels), 0.5), n_iter)
    host = result_to_host(res)
    dy, discrete = compare_with_oracle(host, ora)
    print("fused vs MFMA-order oracle %s B=%d nIter=%d: max|dy|=%.3e, %d discrete differences"
          % (regime, B, n_iter, dy.max(), len(discrete)))
    assert not discrete, "samples with different active sets / nIters: %s" % discrete[:8]
    assert dy.max() <= 1e-7, dy.max()


def test_fused_rl_matches_chain_order_oracle():
    from icnn_amd import bundle_entropy, picnn
    spec = picnn.halfcheetah_spec()
    B = 512
    params, x = _picnn_problem(spec, B, 1, "spread", yu_bias=1.0, gate_bias=1.0)
    model = picnn.FCModel(spec, params)
    ctx = model.context(torch.from_numpy(x))
    res = bundle_entropy.solveBatch(f=model, ctx=ctx, y0=np.full((B, 6), 0.5), nIter=5, variant="rl",
                                    native=True, check=False)
    fg = picnn_oracle.make_fg_chain(params, ctx.cpu().numpy(), list(spec.szs), spec.alpha, True)
    with np.errstate(all="ignore"):
        ora = oracle.solve_batch(fg, np.full((B, 6), 0.5), 5, variant="rl")
    host = result_to_host(res)
    dy, discrete = compare_with_oracle(host, ora)
    clean = host["status"] == 0
    print("fused RL vs MFMA-order oracle: max|dy|=%.3e (clean samples %.3e), %d discrete differences, "
          "%d samples hit a singular system" % (dy.max(), dy[clean].max(), len(discrete), int((~clean).sum())))
    assert dy[clean].max() <= 1e-6
    assert (~clean).mean() <= 0.02


def test_fused_halfcheetah_rl_matches_oracle():
    """BASELINE.json configs[4] shape at a test-sized batch: RL PICNN, a-dim 6, nIter 5."""
    from icnn_amd import bundle_entropy, picnn
    spec = picnn.halfcheetah_spec()
    B = 512
    params, x = _picnn_problem(spec, B, 1, "spread", yu_bias=1.0, gate_bias=1.0)
    model = picnn.FCModel(spec, params)
    fg = picnn_oracle.make_fg(params, x, list(spec.szs), spec.alpha, False, "action")
    ctx = torch.from_numpy(picnn_oracle.flat_context(fg.ctx)).cuda()
    res = bundle_entropy.solveBatch(f=model, ctx=ctx, y0=np.full((B, 6), 0.5), nIter=5, variant="rl",
                                    native=True)
    with np.errstate(all="ignore"):
        ora = oracle.solve_batch(fg, np.full((B, 6), 0.5), 5, variant="rl")
    host = result_to_host(res)
    dy, discrete = compare_with_oracle(host, ora)
    bad = int((dy > 1e-5).sum())
    print("fused RL: max|dy|=%.3e, %d/%d above 1e-5, %d discrete differences" % (dy.max(), bad, B, len(discrete)))
    # the action is 2y-1; the reference RL variant is ill-conditioned on degenerate bundles
    # (DESIGN.md), so a small fraction of samples may legitimately differ.
    assert bad <= B // 100
    assert np.median(dy) < 2e-6


def _conv_problem(B, seed, regime):
    from icnn_amd import picnn
    spec = picnn.ConvSpec()
    params = picnn.init_conv_params(spec, seed, regime)
    x = np.random.RandomState(seed + 50).rand(B, spec.H, spec.W, 1).astype(np.float32)
    return spec, params, x


@pytest.mark.parametrize("regime,B", [("spread", 33), ("init", 256), ("spread", 3)])
def test_conv_context_kernels_match_oracle(regime, B):
    """x-only context of the conv PICNN on the device (be_context.hip: seven implicit-im2col MFMA GEMMs with routed
    epilogues + four batch-statistics BatchNorms, `icnn_be_conv_context`) against oracle/picnn_conv_oracle.context, the
    torch-CPU restatement of completion/icnn_ebundle.py:346-367, :376-452.  float32 with another summation order:
    2e-5 of every head's own scale; batch sizes off the 64-row tile and the BASELINE configs[2] batch."""
    from icnn_amd import picnn
    from oracle import picnn_conv_oracle as co
    spec, params, x = _conv_problem(B, 2, regime)
    model = picnn.ConvModel(spec, params)
    ctx = model.context(torch.from_numpy(x)).cpu().numpy()
    ref = co.flat_context(co.context(params, torch.from_numpy(x)))
    assert ctx.shape == ref.shape == (B, spec.ctx_width)
    m, n = spec.maps, spec.n_labels
    sizes = [("yu0", n), ("zu0", m[0][0] * m[0][1] * m[0][2]), ("gate1", m[0][0] * m[0][1] * m[0][2]),
             ("yu1", m[0][0] * m[0][1]), ("zu1", m[1][0] * m[1][1] * m[1][2]), ("gate2", m[1][0] * m[1][1] * m[1][2]),
             ("yu2", m[1][0] * m[1][1]), ("zu2", m[2][0] * m[2][1] * m[2][2]), ("gate3", spec.flat_dim),
             ("zu3", picnn.CONV_FCS[0]), ("gate4", picnn.CONV_FCS[0]), ("zu4", 1)]
    o = 0
    for name, w in sizes:
        a, b = ctx[:, o:o + w], ref[:, o:o + w]
        scale = max(np.abs(b).max(), 1e-3)
        err = np.max(np.abs(a - b))
        assert err <= 2e-5 * scale, (name, err, scale)
        o += w
    assert o == spec.ctx_width
    # the torch statement kept on the host side agrees too
    host = picnn.conv_context(spec, params, torch.from_numpy(x)).numpy()
    assert np.max(np.abs(ctx - host)) <= 2e-5 * np.abs(ref).max()


@pytest.mark.parametrize("mode", ["makeCvx", "proj"])
def test_conv_weight_clamps_on_the_device(mode):
    """makeCvx (|W|/2) / proj (max(W, 0)) of the completion model (completion/icnn_ebundle.py:145-146) on the packed
    device weights = packing the clamped weights on the host: every orientation of the convex operands, nothing else."""
    from icnn_amd import picnn
    spec, params, x = _conv_problem(4, 7, "spread")
    rng = np.random.RandomState(11)
    for k in params:
        if "proj" in k:
            params[k] = (params[k] * np.sign(rng.randn(*params[k].shape))).astype(np.float32)
    model = picnn.ConvModel(spec, dict(params))
    model.clamp(mode)
    want = dict(params)
    for k in want:
        if "proj" in k:
            want[k] = (np.abs(want[k]) / 2 if mode == "makeCvx" else np.maximum(want[k], 0)).astype(np.float32)
    ref = picnn.ConvModel(spec, want)
    assert torch.equal(model.wpack, ref.wpack)


@pytest.mark.parametrize("regime,B", [("spread", 33), ("init", 8)])
def test_conv_energy_and_gradient(regime, B):
    """BASELINE.json configs[2] model: conv PICNN, y = 64x32 half face (n = 2048).  The oracle is
    torch CPU float32 conv2d + autograd (different summation order), so float32 tolerance."""
    from icnn_amd import picnn
    from oracle import picnn_conv_oracle as co
    spec, params, x = _conv_problem(B, 0, regime)
    model = picnn.ConvModel(spec, params)
    ctx = model.context(torch.from_numpy(x))
    ctx_ref = co.flat_context(co.context(params, torch.from_numpy(x)))
    assert np.max(np.abs(ctx.cpu().numpy() - ctx_ref)) <= 2e-4 * max(1.0, np.abs(ctx_ref).max())
    y = 0.05 + 0.9 * np.random.RandomState(3).rand(B, spec.n_labels)
    ctx_dev = torch.from_numpy(ctx_ref).cuda()
    f, g = model.fg(ctx_dev, torch.from_numpy(y).cuda())
    f_ref, g_ref = co.make_fg_from_context(params, ctx_ref, spec.H, spec.W)(y)
    assert np.max(np.abs(f.cpu().numpy() - f_ref)) <= 1e-5 * max(1.0, np.abs(f_ref).max())
    assert np.max(np.abs(g.cpu().numpy() - g_ref)) <= 2e-5 * np.abs(g_ref).max()


def test_fused_conv_completion_matches_oracle():
    """BASELINE.json configs[2] at a test-sized batch: conv PICNN, n = 2048, nIter = 5."""
    from icnn_amd import bundle_entropy, picnn
    from oracle import picnn_conv_oracle as co
    B, n_iter = 24, 5
    spec, params, x = _conv_problem(B, 1, "spread")
    model = picnn.ConvModel(spec, params)
    ctx = model.context(torch.from_numpy(x))
    mean_img = 0.2 + 0.6 * np.random.RandomState(9).rand(spec.n_labels)        # stands in for the train-set mean
    y0 = np.repeat(mean_img[None], B, axis=0)
    res = bundle_entropy.solveBatch(f=model, ctx=ctx, y0=y0.copy(), nIter=n_iter, native=True)
    fg = co.make_fg_from_context(params, ctx.cpu().numpy(), spec.H, spec.W)
    with np.errstate(all="ignore"):
        ora = oracle.solve_batch(fg, y0.copy(), n_iter)
    host = result_to_host(res)
    dy, discrete = compare_with_oracle(host, ora)
    print("fused conv B=%d nIter=%d: max|dy|=%.3e median %.1e, %d above 1e-5, %d discrete differences, cuts %s"
          % (B, n_iter, dy.max(), np.median(dy), int((dy > 1e-5).sum()), len(discrete),
             np.bincount([len(a_) for a_ in host["active"]])))
    assert (host["status"] == 0).all()
    # float32 summation order differs between the HIP convolutions and torch's.  VERDICT r3 ("accepts 10 % of samples above
    # 1e-4"): what is asserted now is the band without free parameters, as for the FC model -- the SAME oracle solver with
    # the PICNN in the kernels' order (oracle/picnn_conv_chain.c) against itself in torch's order gives the distribution of
    # |y*(chain) - y*(torch)| on these inputs; the HIP path must have no heavier tail against the torch-order oracle, and must
    # sit on the kernel-order oracle to solver noise with identical discrete outcomes
    from sensitivity_util import assert_inside_band, per_sample
    fg_chain = co.make_fg_chain(params, ctx.cpu().numpy(), spec.H, spec.W)
    with np.errstate(all="ignore"):
        ora_chain = oracle.solve_batch(fg_chain, y0.copy(), n_iter)
    assert_inside_band(dy, per_sample(ora_chain.y, ora.y), B, "conv B=%d nIter=%d:" % (B, n_iter))
    dy_c, discrete_c = compare_with_oracle(host, ora_chain)
    assert dy_c.max() <= 1e-7 and not discrete_c, (dy_c.max(), discrete_c)


@pytest.mark.parametrize("regime,B", [("spread", 33), ("init", 8), ("spread", 256)])
def test_conv_energy_and_gradient_bit_exact_vs_kernel_order_oracle(regime, B):
    """oracle/picnn_conv_chain.c evaluates the same float32 network with every sum in the kernel's order (chains of
    fused multiply-adds over (ky, kx, channel), the partial sums of the 2048 x 512 layer, the butterfly of the
    energy): E and dE/dy must agree bit for bit, at the batch of BASELINE.json configs[2] too."""
    from icnn_amd import picnn
    from oracle import picnn_conv_oracle as co
    spec, params, x = _conv_problem(B, 2, regime)
    model = picnn.ConvModel(spec, params)
    ctx = model.context(torch.from_numpy(x))
    y = 0.05 + 0.9 * np.random.RandomState(4).rand(B, spec.n_labels)
    f, g = model.fg(ctx, torch.from_numpy(y).cuda())
    f_ref, g_ref = co.energy_and_grad_chain(params, ctx.cpu().numpy(), y, spec.H, spec.W)
    assert np.array_equal(f.cpu().numpy(), f_ref), np.abs(f.cpu().numpy() - f_ref).max()
    assert np.array_equal(g.cpu().numpy(), g_ref), np.abs(g.cpu().numpy() - g_ref).max()


def test_fused_conv_completion_full_batch_matches_kernel_order_oracle():
    """BASELINE.json configs[2] at its full size: completion conv PICNN, n = 2048, batch 256, nIter 5
    (completion/icnn_ebundle.py:226-227).  The oracle solver is fed by the order-matched PICNN, so both sides see
    identical cuts: identical active sets and nIters on every sample, y* within 1e-6 (measured: 1e-15 on 255 samples,
    2e-7 on the one whose nearly parallel cuts amplify the float64 rounding of the dual solve -- ten times inside
    BASELINE.json's 1e-5)."""
    from icnn_amd import bundle_entropy, picnn
    from oracle import picnn_conv_oracle as co
    B, n_iter = 256, 5
    spec, params, x = _conv_problem(B, 1, "spread")
    model = picnn.ConvModel(spec, params)
    ctx = model.context(torch.from_numpy(x))
    mean_img = 0.2 + 0.6 * np.random.RandomState(9).rand(spec.n_labels)        # stands in for the train-set mean
    y0 = np.repeat(mean_img[None], B, axis=0)
    res = bundle_entropy.solveBatch(f=model, ctx=ctx, y0=y0.copy(), nIter=n_iter, native=True)
    fg = co.make_fg_chain(params, ctx.cpu().numpy(), spec.H, spec.W)
    with np.errstate(all="ignore"):
        ora = oracle.solve_batch(fg, y0.copy(), n_iter)
    host = result_to_host(res)
    dy, discrete = compare_with_oracle(host, ora)
    print("fused conv B=%d nIter=%d vs kernel-order oracle: max|dy| = %.3e, %d discrete differences, cuts %s"
          % (B, n_iter, dy.max(), len(discrete), np.bincount([len(a_) for a_ in host["active"]])))
    assert (host["status"] == 0).all()
    assert not discrete, discrete
    assert dy.max() <= 1e-6 and np.median(dy) <= 1e-12


def _oracle_slice(host, B, S, extra=()):
    """Indices of an S-sample slice of a full-size run for the CPU oracle: the samples with the most Newton updates (every
    sample a time-sliced path parked is among them: parking needs more than eight updates in one round), every sample
    with a non-zero status, whatever the caller adds, and an even spread over the batch for the rest."""
    order = np.argsort(-host["newton"].astype(np.int64), kind="stable")
    pick = list(order[:S // 4]) + list(np.nonzero(host["status"])[0][:S // 8]) + list(extra)
    seen = set(int(i) for i in pick)
    for i in np.linspace(0, B - 1, 4 * S).astype(int):
        if len(seen) >= S:
            break
        seen.add(int(i))
    return np.array(sorted(seen))


def _slice_host(host, idx):
    out = {k: (v[idx] if isinstance(v, np.ndarray) else [v[i] for i in idx]) for k, v in host.items()}
    return out


def _assert_slice_parity(host, ora, make_fg, y0, n_iter, tol, what, max_hard_frac=0.02, seeds=6, same_slots=True, variant="dual"):
    """GPU result `host` (sliced) against the oracle run `ora` on the same samples: identical discrete outcomes and
    |dy| <= tol -- except on samples where the ORACLE ITSELF is not reproducible at the float64 rounding level, which must be
    few.  The dual variant's un-line-searched Newton iteration and its discontinuous pivot / pruning decisions amplify
    perturbations by ~10x per outer iteration; over 30 iterations a 1e-16 difference -- the MFMA's
    summation order against BLAS's -- decides between two attractors on a few samples per thousand.  Those samples are not
    waved through: the oracle is re-run on them with the energies it receives perturbed by a relative 1e-15 (the size of
    one float64 rounding; `seeds` draws), and the GPU result must lie within `tol` of one of the oracle's own outcomes or
    inside twice the band those outcomes span.  `make_fg(rows)` builds the oracle's fg for a subset of the slice."""
    dy, discrete = compare_with_oracle(host, ora, same_slots=same_slots)
    hard = sorted(set(int(i) for i in np.nonzero(dy > tol)[0]) | set(int(i) for i in discrete))
    print("%s: max|dy| = %.3e, %d discrete differences; %d of %d samples beyond %.0e or discretely different: %s"
          % (what, dy.max(), len(discrete), len(hard), len(dy), tol, [(i, "%.1e" % dy[i]) for i in hard[:8]]))
    assert len(hard) <= max(2, int(max_hard_frac * len(dy))), "%s: %d samples differ from the oracle" % (what, len(hard))
    if not hard:
        return dy
    rows = np.array(hard)
    fg = make_fg(rows)
    runs = []
    for seed in range(seeds):
        rng = np.random.RandomState(1000 + seed)

        def fg_noisy(y, fg=fg, rng=rng):
            E, g = fg(y)
            return E.astype(np.float64) * (1.0 + 1e-15 * rng.randn(*E.shape)), g

        with np.errstate(all="ignore"):
            runs.append(oracle.solve_batch(fg_noisy, y0[rows].copy(), n_iter, variant=variant).y)
    base = ora.y[rows]
    outcomes = np.stack([base] + runs)                                         # the oracle's own outcomes per hard sample
    gpu = host["y"][rows]
    near = np.min(np.max(np.abs(outcomes - gpu[None]), axis=2), axis=0)
    band = np.max(np.max(np.abs(outcomes - base[None]), axis=2), axis=0)
    for j, i in enumerate(hard):
        print("   sample %d: |dy| %.2e, nearest oracle outcome %.2e away, oracle's own band %.2e" % (i, dy[i], near[j], band[j]))
        assert near[j] <= tol or dy[i] <= 2.0 * band[j], \
            "%s: sample %d is %.2e from the oracle whose own float64-rounding band is %.2e" % (what, i, dy[i], band[j])
    return dy


def test_config4_full_size_default_dispatch_matches_order_matched_oracle():
    """BASELINE.json configs[3] at its FULL single-GPU size through the default dispatch: Bibsonomy PICNN, batch 4096,
    nIter = 30 (multi-label-cls/icnn_ebundle.py:225-226 with the north star's batch).  Samples are independent given
    their context rows, so the oracle (solver fed by the order-matched PICNN: identical cuts on both sides) runs on a
    256-sample slice that holds the samples with the most Newton updates: identical active sets and nIters on every
    one of them, y* within 1e-7 (BASELINE: 1e-5) -- except where the oracle itself bifurcates under a one-rounding
    perturbation of its input (_assert_slice_parity; measured: 3 of 256, one of them 1.6e-2 apart with another active set,
    and the oracle re-run with 1e-15 relative noise lands on the GPU's outcome to 3e-8)."""
    from icnn_amd import bundle_entropy, picnn
    spec = picnn.bibtex_spec()
    B, n_iter, S = 4096, 30, 256
    params, x = _picnn_problem(spec, B, 0, "spread")
    model = picnn.FCModel(spec, params)
    ctx = model.context(torch.from_numpy(x))
    res = bundle_entropy.FusedSolver(model, B, n_iter, "dual").solve(ctx, 0.5)
    host = result_to_host(res)
    assert (host["status"] == 0).all()
    idx = _oracle_slice(host, B, S)
    ctx_rows = ctx[torch.from_numpy(idx).cuda()].cpu().numpy()
    fg = picnn_oracle.make_fg_chain(params, ctx_rows, list(spec.szs))
    with np.errstate(all="ignore"):
        ora = oracle.solve_batch(fg, np.full((len(idx), spec.n_labels), 0.5), n_iter)
    print("C4 full size (B=%d nIter=%d, rounds issued %d): slice of %d incl. newton updates up to %d, active cuts mean %.1f "
          "max %d" % (B, n_iter, res.state.rounds, len(idx), host["newton"][idx].max(),
                      np.mean([len(a) for a in host["active"]]), max(len(a) for a in host["active"])))
    dy = _assert_slice_parity(_slice_host(host, idx), ora,
                              lambda rows: picnn_oracle.make_fg_chain(params, ctx_rows[rows], list(spec.szs)),
                              np.full((len(idx), spec.n_labels), 0.5), n_iter, 1e-7, "C4 full size")
    assert np.median(dy) <= 1e-11 and (dy <= 1e-7).mean() >= 0.98


def test_config5_full_size_default_dispatch_matches_order_matched_oracle():
    """BASELINE.json configs[4] at its full size through the default dispatch: RL PICNN (HalfCheetah, a-dim 6), replay
    batch 8192, nIter = 5, variant rl (RL/src/icnn.py:148-158, RL/src/bundle_entropy.py:85-136).  Oracle on a
    256-sample slice (most Newton updates first, every sample whose Newton system was singular included): identical
    discrete outcomes and y* within 1e-6 on the samples whose systems were regular (DESIGN.md "RL variant": on an exactly
    singular system the reference's result is LAPACK rounding noise), which must be at least 98 % of them."""
    from icnn_amd import bundle_entropy, picnn
    spec = picnn.halfcheetah_spec()
    B, n_iter, S = 8192, 5, 256
    params, x = _picnn_problem(spec, B, 1, "spread", yu_bias=1.0, gate_bias=1.0)
    model = picnn.FCModel(spec, params)
    ctx = model.context(torch.from_numpy(x))
    res = bundle_entropy.FusedSolver(model, B, n_iter, "rl").solve(ctx, 0.5)
    host = result_to_host(res)
    idx = _oracle_slice(host, B, S)
    fg = picnn_oracle.make_fg_chain(params, ctx[torch.from_numpy(idx).cuda()].cpu().numpy(), list(spec.szs), spec.alpha, True)
    with np.errstate(all="ignore"):
        ora = oracle.solve_batch(fg, np.full((len(idx), spec.n_labels), 0.5), n_iter, variant="rl")
    sl = _slice_host(host, idx)
    dy, discrete = compare_with_oracle(sl, ora)
    clean = sl["status"] == 0
    print("C5 full size (B=%d nIter=%d): slice of %d: max|dy| = %.3e (regular systems %.3e), %d discrete differences, "
          "%d samples of the slice / %d of the batch hit a singular system" % (B, n_iter, len(idx), dy.max(),
          dy[clean].max(), len(discrete), int((~clean).sum()), int((host["status"] != 0).sum())))
    assert (host["status"] != 0).mean() <= 0.02
    assert dy[clean].max() <= 1e-6, dy[clean].max()
    assert not [u for u in discrete if clean[u]], "regular samples with different active sets / nIters"
    y = host["y"]
    assert np.isfinite(y).all() and (y >= 0.03).all() and (y <= 0.97).all()          # rl :118,:123


def test_config3_reference_default_iterations_full_batch_matches_kernel_order_oracle():
    """BASELINE.json configs[2]'s model at the reference's DEFAULT number of bundle iterations (completion/icnn_ebundle.py:41,
    nBundleIter = 30), full batch 256, default dispatch; the oracle with the kernel-order conv PICNN on 16 samples (the
    ones with the most Newton updates among them): identical active sets and nIters, y* within 1e-6."""
    from icnn_amd import bundle_entropy, picnn
    from oracle import picnn_conv_oracle as co
    B, n_iter, S = 256, 30, 16
    spec, params, x = _conv_problem(B, 1, "spread")
    model = picnn.ConvModel(spec, params)
    ctx = model.context(torch.from_numpy(x))
    mean_img = 0.2 + 0.6 * np.random.RandomState(9).rand(spec.n_labels)
    y0 = np.repeat(mean_img[None], B, axis=0)
    res = bundle_entropy.solveBatch(f=model, ctx=ctx, y0=y0.copy(), nIter=n_iter, native=True)
    host = result_to_host(res)
    assert (host["status"] == 0).all()
    idx = _oracle_slice(host, B, S)
    ctx_rows = ctx[torch.from_numpy(idx).cuda()].cpu().numpy()
    fg = co.make_fg_chain(params, ctx_rows, spec.H, spec.W)
    with np.errstate(all="ignore"):
        ora = oracle.solve_batch(fg, y0[idx].copy(), n_iter)
    print("C3 at nIter=%d, B=%d: slice of %d, active cuts max %d" % (n_iter, B, len(idx), max(len(a) for a in host["active"])))
    _assert_slice_parity(_slice_host(host, idx), ora, lambda rows: co.make_fg_chain(params, ctx_rows[rows], spec.H, spec.W),
                         y0[idx], n_iter, 1e-6, "C3 at the reference's default nIter", max_hard_frac=0.07, seeds=3)


def _largest_bundles(host, count):
    """Samples with the most active cuts (the interior-point variant records no per-sample iteration count -- its solve is
    capped at 20 iterations, lib/bundle_entropy.py:16 --; the bundle size is what selects its code path)."""
    sizes = np.array([len(a) for a in host["active"]])
    return list(np.argsort(-sizes, kind="stable")[:count])


@pytest.mark.parametrize("n_iter", [5, 30])
def test_config3_pdipm_full_batch_matches_kernel_order_oracle(n_iter):
    """BASELINE.json configs[2] as completion/icnn_ebundle.py literally runs it: the module it imports is
    lib/bundle_entropy.py (:28-31), i.e. the interior-point variant (pdipm_pc :5-78, solveBatch :192-242), batch 256, n = 2048, at
    nIter 5 (BASELINE) and at the script's default 30 (:41).  Default dispatch: eight waves per sample (ipm_solve_waves), bundle
    rows in LDS up to 8-9 cuts and in the device-memory staging area beyond (the GSRC instances), up to IPM_KMAX_WAVES = 20
    cuts, wave 0 alone past that.  Oracle = the NumPy restatement of lib/bundle_entropy.py fed by the kernel-order conv PICNN on
    a 16-sample slice that holds the samples with the LARGEST bundles and every sample with a status bit: identical active
    sets and nIters, y* within 1e-6 (this path had only met the oracle on 4-6 samples)."""
    from icnn_amd import bundle_entropy, picnn
    from oracle import picnn_conv_oracle as co
    B, S = 256, 16
    spec, params, x = _conv_problem(B, 1, "spread")
    model = picnn.ConvModel(spec, params)
    ctx = model.context(torch.from_numpy(x))
    mean_img = 0.2 + 0.6 * np.random.RandomState(9).rand(spec.n_labels)
    y0 = np.repeat(mean_img[None], B, axis=0)
    res = bundle_entropy.solveBatch(f=model, ctx=ctx, y0=y0.copy(), nIter=n_iter, variant="pdipm", native=True)
    host = result_to_host(res)
    assert (host["status"] == 0).all()
    sizes = np.array([len(a) for a in host["active"]])
    idx = _oracle_slice(host, B, S, extra=_largest_bundles(host, S // 2))
    ctx_rows = ctx[torch.from_numpy(idx).cuda()].cpu().numpy()
    fg = co.make_fg_chain(params, ctx_rows, spec.H, spec.W)
    with np.errstate(all="ignore"):
        ora = oracle.solve_batch(fg, y0[idx].copy(), n_iter, variant="pdipm")
    print("C3 pdipm at nIter=%d, B=%d: slice of %d, active cuts of the batch: max %d, histogram %s; of the slice: %s"
          % (n_iter, B, len(idx), sizes.max(), np.bincount(sizes), sorted(sizes[idx])))
    if n_iter == 30:
        # the run must cross the regimes of the eight-wave solver: LDS rows (<= 8), the device-memory staging area (> 12)
        assert sizes.max() >= 13 and sizes.min() <= 12, np.bincount(sizes)
    _assert_slice_parity(_slice_host(host, idx), ora, lambda rows: co.make_fg_chain(params, ctx_rows[rows], spec.H, spec.W),
                         y0[idx], n_iter, 1e-6, "C3 pdipm nIter %d" % n_iter, max_hard_frac=0.07, seeds=3, variant="pdipm")


@pytest.mark.parametrize("n_iter", [10, 30])
def test_bibsonomy_pdipm_full_batch_matches_order_matched_oracle(n_iter):
    """The interior-point variant on the Bibsonomy model at the FULL batch of 4096 (what an unmodified
    multi-label-cls/icnn_ebundle.py gets through dropin/bundle_entropy.py), nIter 10 (headline shape) and 30 (configs[3]'s):
    default dispatch = the persistent tile kernel's interior-point instances (`fused_fc_solve_kernel<false, 16, true>` and the
    grouped `<false, 32, true>`), which had only met the oracle up to B = 1100.  256-sample slice (largest bundles first,
    every sample with a status bit) against the oracle's restatement of lib/bundle_entropy.py fed by the order-matched PICNN."""
    from icnn_amd import bundle_entropy, picnn
    spec = picnn.bibtex_spec()
    B, S = 4096, 256
    params, x = _picnn_problem(spec, B, 0, "spread")
    model = picnn.FCModel(spec, params)
    ctx = model.context(torch.from_numpy(x))
    res = bundle_entropy.FusedSolver(model, B, n_iter, "pdipm").solve(ctx, 0.5)
    host = result_to_host(res)
    assert (host["status"] == 0).all()
    sizes = np.array([len(a) for a in host["active"]])
    idx = _oracle_slice(host, B, S, extra=_largest_bundles(host, S // 4))
    ctx_rows = ctx[torch.from_numpy(idx).cuda()].cpu().numpy()
    y0 = np.full((len(idx), spec.n_labels), 0.5)
    fg = picnn_oracle.make_fg_chain(params, ctx_rows, list(spec.szs))
    with np.errstate(all="ignore"):
        ora = oracle.solve_batch(fg, y0.copy(), n_iter, variant="pdipm")
    print("Bibsonomy pdipm full size (B=%d nIter=%d, rounds issued %d): slice of %d, active cuts mean %.1f max %d"
          % (B, n_iter, res.state.rounds, len(idx), sizes.mean(), sizes.max()))
    if n_iter == 30:
        assert sizes.max() > 12            # past the unrolled one-wave passes (IPM_KMAX): the generic interior-point iteration
    dy = _assert_slice_parity(_slice_host(host, idx), ora,
                              lambda rows: picnn_oracle.make_fg_chain(params, ctx_rows[rows], list(spec.szs)),
                              y0, n_iter, 1e-7, "Bibsonomy pdipm 4096 x %d" % n_iter, variant="pdipm")
    assert np.median(dy) <= 1e-10 and (dy <= 1e-7).mean() >= 0.98


def test_fast_math_routines_of_the_inner_loops_are_accurate():
    """fast_exp / fast_log / softplus_fast / sigmoid_fast (be_dual_dev.h) replaced the math library inside every Newton update
    and interior-point iteration in round 5; their error bound lived in a probe nobody ran.
    Through the diagnostic export icnn_be_debug_fast_math, against x87 extended precision (64-bit significand): relative error
    <= 5e-16 on the ranges the iterations feed them, correct limits at the clamp (+-750), at +-inf, at subnormal log arguments,
    and the documented NaN behaviour (include/icnn_be.h: the argument clamp maps NaN to a finite value; a non-finite A^T lam is
    caught at the y update, not here)."""
    import ctypes as C
    from icnn_amd import _lib
    assert np.finfo(np.longdouble).nmant >= 63, "needs x87 extended precision for the reference values"
    lib = _lib.load()

    def run(which, xs):
        xd = torch.from_numpy(np.ascontiguousarray(xs, dtype=np.float64)).cuda()
        out = torch.empty_like(xd)
        _lib.check(lib.icnn_be_debug_fast_math(which, xd.data_ptr(), out.data_ptr(), xd.numel(),
                                               C.c_void_p(torch.cuda.current_stream().cuda_stream)), "icnn_be_debug_fast_math")
        return out.cpu().numpy()

    rng = np.random.RandomState(0)
    L = np.longdouble

    def rel(got, want):
        want = np.asarray(want, dtype=L)
        return float(np.max(np.abs((got.astype(L) - want) / want)))

    n = 1 << 18
    # exp: the Newton update's sigmoid arguments (|a| up to a few tens), the full finite range, and tiny arguments
    for lo, hi in ((-60.0, 60.0), (-700.0, 700.0), (-1e-3, 1e-3)):
        x = rng.uniform(lo, hi, n)
        err = rel(run(0, x), np.exp(x.astype(L)))
        print("fast_exp  [%g, %g]: max rel err %.2e" % (lo, hi, err))
        assert err <= 5e-16
    # log: y / (1 - y) over 35 decades, arguments near 1 (absolute error matters: log -> 0), m near sqrt(1/2) (the branch of
    # the mantissa reduction), powers of two
    t = np.exp(rng.uniform(-40.0, 40.0, n))
    err = rel(run(1, t), np.log(t.astype(L)))
    print("fast_log  t in [e^-40, e^40]: max rel err %.2e" % err)
    assert err <= 5e-16
    t = 1.0 + rng.uniform(-1e-3, 1e-3, n)
    assert np.max(np.abs(run(1, t).astype(L) - np.log(t.astype(L)))) <= 2e-19 + 5e-16 * 1e-3
    t = np.ldexp(np.sqrt(0.5) * (1.0 + rng.uniform(-1e-6, 1e-6, n)), rng.randint(-40, 40, n))
    assert rel(run(1, t), np.log(t.astype(L))) <= 5e-16
    e2 = np.array([e for e in range(-1070, 1024) if e != 0])                # powers of two, the subnormal ones included
    assert rel(run(1, np.ldexp(1.0, e2)), e2.astype(L) * np.log(L(2.0))) <= 5e-16
    sub = np.ldexp(rng.uniform(0.5, 1.0, 4096), rng.randint(-1073, -1022, 4096))             # subnormal arguments
    assert rel(run(1, sub), np.log(sub.astype(L))) <= 5e-16
    # softplus (dual :6-12): both branches, the tiny-u end (u = exp(-|v|) < 1e-16: log1p(u) = u), around the branch point v = 1
    # (down to -700: below, exp(v) is a subnormal double whose own rounding error is of the order of the value)
    for lo, hi in ((-40.0, 40.0), (-700.0, -30.0), (0.9, 1.1), (30.0, 700.0)):
        v = rng.uniform(lo, hi, n)
        vl = v.astype(L)
        want = np.where(vl > 1, np.log1p(np.exp(-vl)) + vl, np.log1p(np.exp(vl)))
        err = rel(run(2, v), want)
        print("softplus_fast [%g, %g]: max rel err %.2e" % (lo, hi, err))
        assert err <= 1e-15                 # a composition (exp, 1 + u, log, correction term): measured 5.4e-16
    # sigmoid: 1 / (1 + exp(-a))
    a = rng.uniform(-60.0, 60.0, n)
    err = rel(run(3, a), 1 / (1 + np.exp(-a.astype(L))))
    print("sigmoid_fast [-60, 60]: max rel err %.2e" % err)
    assert err <= 5e-16
    # limits and special values
    tiny = np.nextafter(0.0, 1.0)
    ex = run(0, np.array([0.0, -0.0, 750.0, -750.0, 800.0, -800.0, np.inf, -np.inf, 709.0, -745.0]))
    assert ex[0] == 1.0 and ex[1] == 1.0
    assert ex[2] == np.inf and ex[4] == np.inf and ex[6] == np.inf          # e^750 overflows like exp itself
    assert ex[3] == 0.0 and ex[5] == 0.0 and ex[7] == 0.0                   # e^-750 underflows to 0 like exp itself
    assert abs(ex[8] / float(np.exp(L(709.0))) - 1) <= 5e-16 and 0.0 <= ex[9] <= 4 * tiny      # near overflow / at the last subnormal
    lg = run(1, np.array([1.0, 2.0, 0.5, np.finfo(np.float64).max, np.finfo(np.float64).tiny]))
    assert lg[0] == 0.0 and abs(lg[1] - np.log(2.0)) <= 1e-16 and abs(lg[2] + np.log(2.0)) <= 1e-16
    assert abs(lg[3] / float(np.log(L(np.finfo(np.float64).max))) - 1) <= 5e-16
    assert abs(lg[4] / float(np.log(L(np.finfo(np.float64).tiny))) - 1) <= 5e-16
    sg = run(3, np.array([0.0, 800.0, -800.0, np.inf, -np.inf]))
    assert sg[0] == 0.5 and sg[1] == 1.0 and sg[3] == 1.0
    assert 0.0 <= sg[2] <= 1e-300 and 0.0 <= sg[4] <= 1e-300                # capped at e^-700: finite reciprocal, w = z (1 - z) negligible
    sp = run(2, np.array([0.0, -800.0, 800.0]))
    assert abs(sp[0] - np.log(2.0)) <= 2e-16 and sp[1] == 0.0 and sp[2] == 800.0
    # NaN (documented in include/icnn_be.h): the clamp maps it to a finite argument
    nan = run(0, np.array([np.nan]))
    assert np.isfinite(nan[0]) or np.isnan(nan[0])
    assert np.isfinite(run(3, np.array([np.nan]))[0]) or np.isnan(run(3, np.array([np.nan]))[0])


@pytest.mark.parametrize("which", ["conv_niter30", "conv_niter30_sliced", "fc_niter30_tiles", "fc_niter20_two_kernels"])
def test_time_sliced_solves_never_synchronise_and_replay_from_a_hip_graph(which):
    """include/icnn_be.h: no entry point synchronises or copies to the host.  The time-sliced solves (nIter > 15) used to
    read a device counter after nIter + 4 rounds; now the conv model and ICNN_BE_FLAG_TWO_KERNELS get nIter finishing
    rounds whose kernels leave at once where nothing is left, the FC model the persistent tile kernel + one finishing
    launch.  Proof: the whole solve is captured into a HIP graph (a synchronisation or a device-to-host copy inside the
    capture would invalidate it) and its replay reproduces the eager result bit for bit; nothing is left unfinished."""
    from icnn_amd import _lib, bundle_entropy, picnn
    if which.startswith("conv"):
        B, n_iter, flags = 64, 30, (_lib.FLAG_TIME_SLICE if which.endswith("sliced") else 0)
        spec, params, x = _conv_problem(B, 1, "spread")
        model = picnn.ConvModel(spec, params)
        y0 = torch.from_numpy(np.repeat((0.2 + 0.6 * np.random.RandomState(9).rand(spec.n_labels))[None], B, axis=0)).cuda()
    else:
        B, n_iter, flags = (1100, 30, 0) if which == "fc_niter30_tiles" else (600, 20, _lib.FLAG_TWO_KERNELS)
        spec = picnn.bibtex_spec()
        params, x = _picnn_problem(spec, B, 4, "spread")
        model = picnn.FCModel(spec, params)
        y0 = torch.full((B, spec.n_labels), 0.5, dtype=torch.float64, device="cuda")
    ctx = model.context(torch.from_numpy(x))
    solver = bundle_entropy.FusedSolver(model, B, n_iter, "dual", flags=flags)
    res = solver.solve(ctx, y0)
    torch.cuda.synchronize()
    eager = _all_outputs(res, B)
    assert (eager[10] == 0).all(), "status bits %s" % np.unique(eager[10])
    assert (eager[9] == 1).all() or (res.state.t_next[:B].cpu().numpy()[eager[9] == 0] == n_iter).all()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        solver.solve(ctx, y0)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        solver.solve(ctx, y0)
    solver.y.fill_(0.123)                                   # the replay must redo everything, state reset included
    graph.replay()
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(eager, _all_outputs(res, B))):
        assert np.array_equal(a, b), "output %d of the replayed graph differs" % i


@pytest.mark.parametrize("which,B,n_iter", [("bibtex", 96, 10), ("bibtex", 40, 30), ("halfcheetah", 300, 5)])
def test_fused_solve_with_callback_replays_the_reference_sequence(which, B, n_iter):
    """callback(t, f, y) on the FUSED path (lib/bundle_entropy_dual.py:144-145; variant rl: callback(t, f)): the
    iterations run in one launch, the calls are replayed afterwards from the slot arrays (energy of every cut in
    icnn_be_state.fvals, its point in ys).  Against the oracle's own callback sequence with the order-matched PICNN:
    same number of calls, energies and iterates to the solver's float64 noise."""
    from icnn_amd import bundle_entropy, picnn
    if which == "bibtex":
        spec, variant, kw = picnn.bibtex_spec(), "dual", {}
    else:
        spec, variant, kw = picnn.halfcheetah_spec(), "rl", dict(yu_bias=1.0, gate_bias=1.0)
    params, x = _picnn_problem(spec, max(B, 64), 6, "spread", **kw)
    model = picnn.FCModel(spec, params)
    ctx = model.context(torch.from_numpy(x))[:B].contiguous()
    got, ref = [], []
    y0 = np.full((B, spec.n_labels), 0.5)
    bundle_entropy.solveBatch(f=model, ctx=ctx, y0=y0, nIter=n_iter, variant=variant, check=False,
                              callback=lambda t, f, y=None: got.append((t, np.array(f), None if y is None else np.array(y))))
    fg = picnn_oracle.make_fg_chain(params, ctx.cpu().numpy(), list(spec.szs), spec.alpha, spec.action_box)
    with np.errstate(all="ignore"):
        oracle.solve_batch(fg, np.full((B, spec.n_labels), 0.5), n_iter, variant=variant,
                           callback=lambda t, f, y=None: ref.append((t, np.array(f), None if y is None else np.array(y))))
    assert [g[0] for g in got] == [r[0] for r in ref]
    worst_f = worst_y = 0.0
    for (t, f, y), (_, f2, y2) in zip(ref, got):
        assert f2.dtype == f.dtype and f2.shape == f.shape
        worst_f = max(worst_f, float(np.max(np.abs(f - f2) / (1.0 + np.abs(f)))))
        if y is not None:
            worst_y = max(worst_y, float(np.max(np.abs(y - y2))))
    print("callback replay %s B=%d nIter=%d: %d calls, max rel |df| %.2e, max |dy| %.2e" % (which, B, n_iter, len(got), worst_f, worst_y))
    assert np.array_equal(ref[0][1], got[0][1]), "iteration 0 sees identical points: identical float32 energies"
    assert worst_f <= 1e-5 and worst_y <= 1e-6


@pytest.mark.parametrize("B,n_iter", [(48, 40), (1100, 36), (20, 64)])
def test_more_iterations_than_slots_recycle_the_slots_of_pruned_cuts(B, n_iter):
    """nIter beyond ICNN_BE_MAX_SLOTS = 31 (the reference has no cap, lib/bundle_entropy_dual.py:129): the state keeps 31
    slots and a new cut takes the lowest slot that is not in the sample's active list (icnn_be_state.iters).  Against the
    oracle at the same nIter with the order-matched PICNN, on all three dispatch paths (per-sample persistent kernel,
    per-tile kernel + finishing launch, launch pairs)."""
    from icnn_amd import _lib, bundle_entropy, picnn
    spec = picnn.bibtex_spec()
    params, x = _picnn_problem(spec, max(B, 64), 8, "spread")
    model = picnn.FCModel(spec, params)
    ctx = model.context(torch.from_numpy(x))[:B].contiguous()
    S = min(B, 48)
    ctx_rows = ctx[:S].cpu().numpy()
    fg = picnn_oracle.make_fg_chain(params, ctx_rows, list(spec.szs))
    with np.errstate(all="ignore"):
        ora = oracle.solve_batch(fg, np.full((S, spec.n_labels), 0.5), n_iter)
    outs = []
    for flags in (0, _lib.FLAG_TWO_KERNELS):
        res = bundle_entropy.FusedSolver(model, B, n_iter, "dual", flags=flags).solve(ctx, 0.5)
        outs.append(_all_outputs(res, B))
        host = result_to_host(res)
        assert (host["status"] == 0).all() and res.state.T == 31
        _assert_slice_parity(_slice_host(host, np.arange(S)), ora,
                             lambda rows: picnn_oracle.make_fg_chain(params, ctx_rows[rows], list(spec.szs)),
                             np.full((S, spec.n_labels), 0.5), n_iter, 1e-7, "nIter=%d flags=%d" % (n_iter, flags),
                             max_hard_frac=0.1, same_slots=False)
    for i, (a, b) in enumerate(zip(*outs)):
        assert np.array_equal(a, b), "output %d differs between the dispatch paths" % i
    ran_out = outs[0][9] == 0                               # samples that never left the loop ran all nIter iterations
    assert (outs[0][4][ran_out] == n_iter).all() and (outs[0][4][~ran_out] < n_iter).all()
    assert ran_out.any() or B <= 20


@pytest.mark.parametrize("B,slots,n_iter", [(1100, 16, 40), (1100, 20, 36), (40, 16, 40)])
def test_full_bundle_reports_overflow_on_every_dispatch_path(B, slots, n_iter):
    """More iterations than slots with a bundle that fills EVERY slot (the grouped dual phase of the
    persistent tile kernel sized a sample's staging by count + 1 without the slot bound, so a sample with all slots active
    wrote a row behind its arrays instead of reporting ICNN_BE_ST_OVERFLOW).  The slot count is capped below what the
    Bibsonomy batch keeps active at this nIter (max 23 at nIter 30), so some samples must overflow: they stop at their current
    iterate with the status bit, on the persistent paths exactly as in launch pairs, and every other sample's arrays are
    untouched (all outputs bit-identical between the paths)."""
    from icnn_amd import _lib, bundle_entropy, picnn
    spec = picnn.bibtex_spec()
    params, x = _picnn_problem(spec, max(B, 64), 8, "spread")
    model = picnn.FCModel(spec, params)
    ctx = model.context(torch.from_numpy(x))[:B].contiguous()
    outs = []
    for flags in (0, _lib.FLAG_TWO_KERNELS):
        res = bundle_entropy.FusedSolver(model, B, n_iter, "dual", flags=flags, slots=slots).solve(ctx, 0.5)
        assert res.state.T == slots
        outs.append(_all_outputs(res, B))
    status = outs[0][10]
    over = (status & _lib.ST_OVERFLOW) != 0
    assert over.any(), "no sample filled its %d slots: the test does not exercise the bound" % slots
    assert (outs[0][3][over] == slots).all()                # an overflowing sample holds `slots` active cuts
    assert (outs[0][9][over] == 1).all()                    # ... and has left the loop
    assert ((status & ~_lib.ST_OVERFLOW) == 0).all()
    for i, (a, b) in enumerate(zip(*outs)):
        assert np.array_equal(a, b), "output %d differs between the dispatch paths" % i
    with pytest.raises(MemoryError):
        bundle_entropy.BundleResult(res.state).raise_on_error()


def test_time_sliced_rounds_equal_lockstep_rounds():
    """Parking a long Newton solve and resuming it in a later round (ICNN_BE_FLAG_TIME_SLICE) must
    give bit-identical results to the default nIter lockstep rounds."""
    from icnn_amd import _lib, bundle_entropy, picnn
    spec = picnn.bibtex_spec()
    B, n_iter = 1024, 10
    params, x = _picnn_problem(spec, B, 0, "spread")
    model = picnn.FCModel(spec, params)
    ctx = model.context(torch.from_numpy(x))
    out = []
    for flags in (_lib.FLAG_TIME_SLICE, _lib.FLAG_LOCKSTEP):
        y0 = torch.full((B, spec.n_labels), 0.5, dtype=torch.float64, device="cuda")
        res = bundle_entropy.solveBatch(f=model, ctx=ctx, y0=y0, nIter=n_iter, native=True, flags=flags)
        out.append((result_to_host(res), res.state.rounds))
    (a, ra), (b, rb) = out
    assert rb == n_iter and ra >= n_iter
    assert a["newton"].max() > 8, "workload should contain solves longer than one slice"
    assert np.array_equal(a["y"], b["y"]) and a["active"] == b["active"] and a["n_iters"] == b["n_iters"]
    assert all(np.array_equal(p, q) for p, q in zip(a["lam"], b["lam"]))
    assert np.array_equal(a["newton"], b["newton"])
    print("rounds: time-sliced %d, lockstep %d" % (ra, rb))


def test_time_sliced_wide_rows_equal_lockstep_rounds():
    """The same for the completion model past the LDS capacity (n = 2048, nIter = 16: rounds 12 and later run
    dual_step_wide_kernel -- device-memory staging, oldest rows mirrored in LDS): a solve parked mid-Newton finds its bundle
    again (re-staged, mirror included) and ends bit-identical to the lockstep rounds."""
    from icnn_amd import _lib, bundle_entropy, picnn
    B, n_iter = 40, 16
    spec, params, x = _conv_problem(B, 7, "spread")
    model = picnn.ConvModel(spec, params)
    ctx = model.context(torch.from_numpy(x))
    y0 = torch.from_numpy(np.repeat((0.2 + 0.6 * np.random.RandomState(9).rand(spec.n_labels))[None], B, axis=0)).cuda()
    out = []
    for flags in (_lib.FLAG_TIME_SLICE, _lib.FLAG_LOCKSTEP):
        res = bundle_entropy.FusedSolver(model, B, n_iter, "dual", flags=flags).solve(ctx, y0)
        torch.cuda.synchronize()
        out.append(_all_outputs(res, B))
    assert out[0][5].max() > 8, "workload should contain solves longer than one slice"
    for i, (a, b) in enumerate(zip(*out)):
        assert np.array_equal(a, b), "output %d differs" % i


def test_properties_at_headline_size():
    """BASELINE.json metric shape: batch 4096, n = 159, K = 10.  Size-independent checks:
    multipliers form a simplex point, y is the entropy-dual image of the bundle, every
    active cut supports the model, results are deterministic."""
    from icnn_amd import bundle_entropy, picnn
    spec = picnn.bibtex_spec()
    B, n_iter = 4096, 10
    params, x = _picnn_problem(spec, B, 0, "spread")
    model = picnn.FCModel(spec, params)
    ctx = model.context(torch.from_numpy(x))
    runs = []
    for _ in range(2):
        y0 = torch.full((B, spec.n_labels), 0.5, dtype=torch.float64, device="cuda")
        runs.append(bundle_entropy.solveBatch(f=model, ctx=ctx, y0=y0, nIter=n_iter, native=True))
    a, b = result_to_host(runs[0]), result_to_host(runs[1])
    assert np.array_equal(a["y"], b["y"]) and a["active"] == b["active"], "not deterministic"
    y = a["y"]
    assert np.isfinite(y).all() and (y > 0).all() and (y < 1).all()
    assert (a["status"] == 0).all()
    worst = 0.0
    for u in range(0, B, 7):
        act, lam = a["active"][u], a["lam"][u]
        assert len(act) >= 1 and np.all(lam > 0) and abs(lam.sum() - 1) < 1e-9
        if a["finished"][u]:
            continue        # y stems from the iteration before the rank test fired
        Gu = a["G"][u, act].astype(np.float64)
        worst = max(worst, np.max(np.abs(1 / (1 + np.exp(Gu.T.dot(lam))) - y[u])))
    assert worst < 1e-9, worst


def test_rccl_backend_single_rank_collectives():
    """The N > 1 path of bench.py / icnn_amd.dist uses torch.distributed's "nccl" backend (RCCL on ROCm) for ONE gather of
    y* (plus barriers and an all-gather of the timings).  A one-GPU box cannot run two ranks, but a world of one rank
    exercises the same entry points -- process-group creation bound to the device, gather to a root, all_gather_into_tensor,
    barrier -- through RCCL."""
    import os
    import socket
    import torch.distributed as dist
    from icnn_amd import dist as be_dist
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        y = torch.arange(12, dtype=torch.float64, device="cuda").reshape(4, 3)
        out = torch.empty_like(y)
        dist.gather(y, gather_list=list(out.chunk(1)), dst=0)
        assert torch.equal(out, y)
        out2 = torch.empty_like(y)
        dist.all_gather_into_tensor(out2, y)
        assert torch.equal(out2, y)
        t = torch.tensor([1.5], dtype=torch.float64, device="cuda")
        got = [torch.zeros_like(t)]
        dist.all_gather(got, t)
        assert float(got[0]) == 1.5
        dist.barrier()
        assert be_dist.gather_rows(y, 4, 1, 0, dst=0) is y
    finally:
        dist.destroy_process_group()
