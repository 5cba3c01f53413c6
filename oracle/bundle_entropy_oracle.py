"""CPU oracle for the bundle-entropy inference path.  TEST INFRASTRUCTURE ONLY.

This module restates, in NumPy, the algorithm of the reference solver so that the
HIP path can be checked against it.  Nothing in the product (`icnn_amd/`) may
import it; only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s
`cpu_baseline` leg do.

Parity status: PINNED for the solver.  `oracle/gen_golden.py` imports the
reference modules by path in the build container, runs them on the seeded
problems of `tests/problems.py`, and stores their outputs under `tests/golden/`;
`tests/test_oracle_golden.py` checks this restatement against those vectors.

What is restated (reference paths relative to the reference checkout):

  variant "dual"  lib/bundle_entropy_dual.py   solveBatch :129-179,
                  proj_newton_logistic :15-85, logexp1p :6-12
  variant "rl"    RL/src/bundle_entropy.py     solveBatch :85-136,
                  proj_newton_logistic :14-83
  variant "pdipm" lib/bundle_entropy.py        solveBatch :192-242 with solver='pc',
                  pdipm_pc :5-78 (Mehrotra predictor-corrector on the primal-dual
                  form), get_step :158-163 -- the module the icnn_ebundle.py scripts
                  literally import ('../lib' on sys.path, `import bundle_entropy`)

Data model (differs from the reference on purpose): the reference keeps ragged
Python lists per sample; here a sample's bundle lives in fixed slots -- the cut
taken at outer iteration t is stored in slot t of `G[u]`, `h[u]`, `ys[u]` -- and
`active[u]` is the ordered list of slots still in the bundle.  This is the same
layout the HIP kernels use in HBM, so tests can compare slot for slot.
`BundleResult.as_reference_tuple()` rebuilds the reference's ragged 6-tuple.

The arithmetic deliberately goes through the same NumPy/LAPACK entry points on
the same shapes and dtypes as the reference (float32 row sums, dgemv on the
promoted bundle, gesv, gesdd-based matrix_rank), because the reference's results
depend on them at the 1e-7 level (see DESIGN.md "precision map").
"""
from dataclasses import dataclass, field

import numpy as np


# --------------------------------------------------------------------------- #
# variant switches
# --------------------------------------------------------------------------- #
@dataclass(frozen=True)
class VariantRules:
    name: str
    newton_cap: int          # dual :30 -> 100, rl :29 -> 20
    armijo: bool             # dual :15 line_search=False, rl :14 True
    scaled_first_step: bool  # rl :64  t0 = min(1/max|d|, 1)
    backoff_cap: int         # dual :67 -> 50, rl :65 -> 10
    tiny_step_on_td: bool    # dual :79 t<1e-10 ; rl :77 max(t|d|)<1e-10
    singular_raises: bool    # dual :56-63 re-raise ; rl :55-62 break
    rank_test: bool          # dual :155-161 ; rl none
    clip: tuple              # rl :118,:123 -> (0.03, 0.97)
    stall_tol: float         # rl :125 -> 1e-6
    callback_arity: int      # dual :145 callback(t, f, y) ; rl :104 callback(t, f)
    default_iters: int
    ipm: bool = False        # lib/bundle_entropy.py: multipliers AND y from pdipm_pc, prune lam > 1e-8


VARIANTS = {
    "dual": VariantRules("dual", 100, False, False, 50, False, True, True, None, None, 3, 10),
    "rl": VariantRules("rl", 20, True, True, 10, True, False, False, (0.03, 0.97), 1e-6, 2, 5),
    "pdipm": VariantRules("pdipm", 20, False, False, 0, False, True, True, None, None, 3, 10, True),
}

_BOUND_EPS = 1e-12      # dual :21 / rl :20
_ARMIJO_ALPHA = 1e-5    # dual :22
_SHRINK = 0.5           # dual :23
_GRAD_TOL = 1e-10       # dual :50
_TINY = 1e-10           # dual :79


def softplus_stable(v):
    """log(1+exp(v)) without overflow -- reference `logexp1p`, dual :6-12."""
    out = np.zeros_like(v)
    big = v > 1
    small = ~big
    out[big] = np.log1p(np.exp(-v[big])) + v[big]
    out[small] = np.log1p(np.exp(v[small]))
    return out


def _dual_objective(c, A, lam):
    return -c.dot(lam) + np.sum(softplus_stable(A.T.dot(lam)))


def simplex_newton(A, b, rules, stats=None):
    """min over the simplex of  -(A 1 + b)^T lam + sum softplus(A^T lam).

    Projected Newton with the largest multiplier eliminated through
    sum(lam) = 1 -- reference `proj_newton_logistic` (dual :15-85, rl :14-83).
    `A` is the k x n stack of cut gradients (dtype as produced by fg), `b` the
    k offsets (float64).  Returns lam (float64, exact zeros where clipped).
    """
    k = A.shape[0]
    c = np.sum(A, axis=1) + b            # float32 row sum when A is float32 (:18)
    lam = np.ones(k) / k                 # :26
    keep = np.ones(k)                    # the reference's `e`
    n_newton = 0

    for _ in range(rules.newton_cap):
        n_newton += 1
        a = A.T.dot(lam)                                     # :32
        z = 1 / (1 + np.exp(-a))                             # :33
        fval = -c.dot(lam) + np.sum(softplus_stable(a))      # :34
        grad = -c + A.dot(z)                                 # :35
        hess = (A * (z * (1 - z))).dot(A.T)                  # :36

        piv = np.argmax(lam)                                 # :39 first maximum
        red = lam.copy()
        red[piv] = 1
        keep[piv] = 0
        col = hess[:, piv]
        g0 = grad - keep * grad[piv]                         # :44
        h0 = (hess - keep[:, None] * col[None, :] - col[:, None] * keep[None, :]
              + hess[piv, piv] * (keep[:, None] * keep[None, :]))   # :45

        bound = (red <= _BOUND_EPS) & (g0 > 0)               # :48
        bound[piv] = True
        free = ~bound
        if np.linalg.norm(g0[free]) < _GRAD_TOL:             # :50
            if stats is not None:
                stats.append(n_newton)
            return lam
        step = np.zeros(k)
        try:
            step[free] = np.linalg.solve(h0[free, :][:, free], -g0[free])   # :55
        except np.linalg.LinAlgError:
            if rules.singular_raises:
                raise
            break                                            # rl :62

        if rules.scaled_first_step:
            t = min(1. / np.max(abs(step)), 1.)              # rl :64
        else:
            t = 1.
        for _ in range(rules.backoff_cap):
            trial = np.maximum(red + t * step, 0)            # :68
            trial[piv] = 1
            lam_new = trial.copy()
            lam_new[piv] = 1. - keep.dot(trial)              # :71
            if lam_new[piv] >= 0:
                if rules.armijo:
                    f_new = _dual_objective(c, A, lam_new)
                    if f_new < fval + t * _ARMIJO_ALPHA * step.dot(g0):   # rl :73
                        break
                else:
                    break
            if rules.tiny_step_on_td:
                if max(t * abs(step)) < _TINY:               # rl :77
                    if stats is not None:
                        stats.append(n_newton)
                    return lam_new
            elif t < _TINY:                                  # dual :79
                if stats is not None:
                    stats.append(n_newton)
                return lam_new
            t *= _SHRINK

        keep[piv] = 1.
        lam = lam_new.copy()

    if stats is not None:
        stats.append(n_newton)
    return lam


def _ratio_step(v, dv):
    """Largest step that keeps v + step dv >= 0, 1 if dv has no negative entry -- reference `get_step`,
    lib/bundle_entropy.py:158-163."""
    neg = dv < 0
    if np.any(neg):
        return np.min((-v / dv)[neg])
    return 1.


def interior_point(G, h, stats=None):
    """min_{y,t} t - H(y)  s.t.  G y + h <= t  by Mehrotra's predictor-corrector method on the primal-dual system --
    reference `pdipm_pc`, lib/bundle_entropy.py:5-78 (its per-iteration print is dropped).  Returns (y, z): the
    minimiser in the open unit box and the multipliers of the k cuts.  Same NumPy / LAPACK calls as the reference
    (dense products with the diagonal matrix included, cholesky + cho_solve)."""
    import scipy.linalg
    k, n = G.shape
    z = np.ones(k) / k                                        # :11
    y = np.full(n, 0.5)                                       # :12
    s = np.ones(k)                                            # :13
    t = 1.                                                    # :14
    ones = np.ones(k)
    for it in range(20):                                      # :16
        grad_negH = np.log(y) - np.log(1. - y)                # :17
        hess_negH_inv = np.diag(1. / (1. / y + 1. / (1. - y)))   # :19
        ry = grad_negH + G.T.dot(z)                           # :26
        rt = 1. - np.sum(z)                                   # :27
        rc = z                                                # :28
        rd = G.dot(y) + h - t * ones + s                      # :29
        pri_res = np.linalg.norm(np.concatenate([ry, [rt]]))  # :32
        dual_res = np.linalg.norm(rd)                         # :33
        if pri_res < 1e-8 and dual_res < 1e-8:                # :39
            if stats is not None:
                stats.append(it)
            return y, z
        M = G.dot(hess_negH_inv).dot(G.T) + np.diag(s / z)    # :41
        chol_M = np.linalg.cholesky(M)                        # :42
        Minv_1 = scipy.linalg.cho_solve((chol_M, True), ones)  # :43

        def kkt(ry, rt, rc, rd):                              # :45-51
            r = rd - G.dot(hess_negH_inv).dot(ry) - (s / z) * rc
            dt = (r.dot(Minv_1) - rt) / Minv_1.sum()
            dz = scipy.linalg.cho_solve((chol_M, True), r - dt)
            ds = -(s / z) * (rc + dz)
            dy = -hess_negH_inv.dot(ry + G.T.dot(dz))
            return dt, dz, ds, dy

        dt_aff, dz_aff, ds_aff, dy_aff = kkt(ry, rt, rc, rd)  # :53
        alpha = min(_ratio_step(z, dz_aff), _ratio_step(s, ds_aff), _ratio_step(y, dy_aff),
                    _ratio_step(-y + 1, -dy_aff), 1.0)        # :55-56
        sig = (np.dot(s + alpha * ds_aff, z + alpha * dz_aff) / (np.dot(s, z))) ** 3   # :57
        mu = np.dot(s, z) / k                                 # :59
        zero_n, zero_k = np.zeros(n), np.zeros(k)             # :61  ry[:] = rt = rd[:] = 0
        rc = -(mu * sig * ones - ds_aff * dz_aff) / s         # :62
        dt_cor, dz_cor, ds_cor, dy_cor = kkt(zero_n, 0, rc, zero_k)   # :63
        dy = dy_aff + dy_cor                                  # :65-68
        dt = dt_aff + dt_cor
        ds = ds_aff + ds_cor
        dz = dz_aff + dz_cor
        alpha = max(0.0, min(1.0, 0.99 * min(_ratio_step(s, ds), _ratio_step(z, dz), _ratio_step(y, dy),
                                             _ratio_step(-y + 1, -dy))))   # :70-71
        y = y + alpha * dy                                    # :73-76 (in place in the reference: same values)
        t = t + alpha * dt
        s = s + alpha * ds
        z = z + alpha * dz
    if stats is not None:
        stats.append(20)
    return y, z


_IPM_PRUNE = 1e-8      # lib/bundle_entropy.py:198,:234-237


@dataclass
class BundleResult:
    """Slot-addressed outcome of one solveBatch call (see module docstring)."""
    y: np.ndarray                 # [B, n] float64; the caller's array, updated in place
    G: np.ndarray                 # [B, T, n] cut gradients, slot t = outer iteration t
    h: np.ndarray                 # [B, T] float64 cut offsets
    ys: np.ndarray                # [B, T, n] float64 points the cuts were taken at
    active: list                  # per sample: ordered slots still in the bundle
    lam: list                     # per sample: float64 multipliers of the active slots, or None
    n_iters: list                 # per sample (reference `nIters`)
    finished: np.ndarray          # [B] bool
    newton_counts: list = field(default_factory=list)

    def as_reference_tuple(self):
        """(x, A, b, lam, xs, nIters) exactly as the reference returns it (dual :179)."""
        A = [[self.G[u, s] for s in act] for u, act in enumerate(self.active)]
        b = [[self.h[u, s] for s in act] for u, act in enumerate(self.active)]
        xs = [[self.ys[u, s] for s in act] for u, act in enumerate(self.active)]
        return self.y, A, b, self.lam, xs, self.n_iters


def solve_batch(fg, y0, n_iter=None, callback=None, variant="dual"):
    """Batched bundle-entropy minimisation of f(y) - H(y) over the unit box.

    Restates reference `solveBatch` (dual :129-179; rl :85-136).  `fg(y)` returns
    (f[B], g[B, n]) for the whole batch; `y0` is float64 [B, n] and is updated in
    place like the reference's `initXs`.
    """
    rules = VARIANTS[variant]
    if n_iter is None:
        n_iter = rules.default_iters
    y = y0
    B, n = y.shape
    G = None
    h = np.zeros((B, n_iter))
    ys = np.zeros((B, n_iter, n))
    active = [[] for _ in range(B)]
    lam = [None] * B
    n_iters = [n_iter] * B
    done = np.zeros(B, dtype=bool)
    newton_counts = []

    for t in range(n_iter):
        f_t, g_t = fg(y)
        if G is None:
            G = np.zeros((B, n_iter, n), dtype=np.asarray(g_t).dtype)
        off_t = f_t - np.sum(g_t * y, axis=1)                 # dual :143
        if callback is not None:
            if rules.callback_arity == 3:
                callback(t, f_t, y)
            else:
                callback(t, f_t)

        for u in range(B):
            if done[u]:
                continue
            G[u, t] = g_t[u]
            h[u, t] = off_t[u]
            ys[u, t] = y[u]
            slots = active[u] + [t]

            if rules.rank_test:
                # dual :155-161 -- the new cut must raise the rank of the bundle
                if np.linalg.matrix_rank(G[u, slots]) < len(slots):
                    done[u] = True
                    n_iters[u] = t - 1
                    continue

            before = y[u].copy()
            Au = G[u, slots]                   # fancy index -> fresh C-contiguous [k, n]
            if rules.ipm:                      # lib/bundle_entropy.py:224-237: y and lam from the interior-point solve
                y[u], lam_u = interior_point(Au, h[u, slots], newton_counts)
                pos = lam_u > _IPM_PRUNE
                active[u] = [s for s, p in zip(slots, pos) if p]
                lam[u] = lam_u[pos]
                continue
            if len(slots) > 1:
                lam_u = simplex_newton(Au, h[u, slots], rules, newton_counts)
                y[u] = 1 / (1 + np.exp(Au.T.dot(lam_u)))     # dual :165
            else:
                lam_u = np.array([1])
                y[u] = 1 / (1 + np.exp(Au[0]))               # dual :168 (cut dtype arithmetic)
            if rules.clip is not None:
                y[u] = np.clip(y[u], rules.clip[0], rules.clip[1])   # rl :118
            if rules.stall_tol is not None and max(abs(before - y[u])) < rules.stall_tol:
                done[u] = True                                # rl :125-126

            pos = lam_u > 0                                   # dual :171-174
            active[u] = [s for s, p in zip(slots, pos) if p]
            lam[u] = lam_u[pos]

        if done.all():
            break

    if G is None:
        G = np.zeros((B, n_iter, n), dtype=np.float32)
    return BundleResult(y, G, h, ys, active, lam, n_iters, done, newton_counts)


def solveBatch(fg, initXs, nIter=None, callback=None, variant="dual"):
    """Reference-shaped entry point: returns the ragged 6-tuple."""
    return solve_batch(fg, initXs, nIter, callback, variant).as_reference_tuple()
