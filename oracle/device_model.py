"""Scalar NumPy model of the algorithm AS THE HIP KERNELS RUN IT.
TEST INFRASTRUCTURE ONLY.

The HIP dual-step kernel cannot call LAPACK, so three steps of the reference
solver are realised differently on the device.  This file spells those device
formulations out in plain NumPy so that CPU-only tests can prove, against the
reference-generated golden vectors, that they lead to the same results:

  1. rank test (reference: np.linalg.matrix_rank = SVD, dual :155).  Device:
     Gram matrix S = A A^T in float64, then Sylvester-inertia counts of
     S - mu I by an unpivoted LDL^T to decide "sigma_min <= tol" with
     tol = sigma_max * max(k, n) * eps(cut dtype); cyclic Jacobi on S only when
     the cheap brackets on sigma_max leave the decision open.  Float64 cuts use
     one-sided (Hestenes) Jacobi on the rows instead, because the Gram matrix
     cannot resolve a 1e-14 relative threshold.
  2. Newton system (reference: np.linalg.solve = gesv, dual :55).  Device:
     Gaussian elimination in natural order (the reduced Hessian is symmetric
     positive semi-definite, no pivoting needed); "singular" = an exactly zero
     pivot, as LAPACK reports it.
  3. Newton cap (reference: always runs its 100 / 20 iterations when the
     un-line-searched iteration falls into a limit cycle; measured: 0.7 % of the
     solves of the Bibsonomy-shaped workload, almost always an attracting
     2-cycle that is reached to ~1e-15 within ~20 iterations and then only
     jitters in the last bits; about one solve in 40 000 is a 3-cycle).  Device:
     once max|lam_t - lam_{t-p}| <= CYCLE_TOL for p = 1, 2 or 3 the remaining
     iterations are skipped and the iterate whose phase matches the reference's
     final count is returned.  The result differs from running
     all iterations by the size of that jitter (<= ~1e-13), far below the
     float64 noise between two BLAS builds.  CYCLE_TOL = 0 disables it.  A 2-cycle that
     is approached slowly (contraction 0.6-0.7 per update, 35-85 updates to settle) is
     extrapolated after 24 updates: even and odd subsequence are moved to their estimated
     limits and the iteration continues from there under the same stopping rule.  The common ratio
     may be negative (each subsequence oscillates around its limit) where the reference's own
     remaining updates would close the gap anyway (ACCEL_NEG_RESID below).

  plus 4. every reduction whose order NumPy fixes (float32 row sums of the
     bundle, the cut offset f - sum(g*y)) is evaluated in NumPy's pairwise order.

Reference line numbers refer to lib/bundle_entropy_dual.py unless prefixed.
"""
import numpy as np

from . import bundle_entropy_oracle as ref


# --------------------------------------------------------------------------- #
# 4. NumPy's pairwise summation order (numpy/_core/src/umath/loops_utils.h.src)
# --------------------------------------------------------------------------- #
def pairwise_sum(a, T):
    n = len(a)
    if n < 8:
        r = T(0)
        for v in a:
            r = T(r + v)
        return r
    if n <= 128:
        r = [T(a[j]) for j in range(8)]
        i = 8
        while i < n - (n % 8):
            for j in range(8):
                r[j] = T(r[j] + a[i + j])
            i += 8
        res = T(T(T(r[0] + r[1]) + T(r[2] + r[3])) + T(T(r[4] + r[5]) + T(r[6] + r[7])))
        while i < n:
            res = T(res + a[i])
            i += 1
        return res
    n2 = n // 2
    n2 -= n2 % 8
    return T(pairwise_sum(a[:n2], T) + pairwise_sum(a[n2:], T))


# --------------------------------------------------------------------------- #
# 1. rank decision
# --------------------------------------------------------------------------- #
def ldl_negative_pivots(S, mu):
    """Number of eigenvalues of S below mu (Sylvester), unpivoted LDL^T; a zero
    pivot counts as 'not above'."""
    M = S - mu * np.eye(len(S))
    k = len(M)
    neg = 0
    for p in range(k):
        d = M[p, p]
        if not d > 0:
            neg += 1
            if d == 0:
                return neg + (k - p - 1)      # cannot continue; treat the rest as suspect
        for r in range(p + 1, k):
            f = M[r, p] / d
            M[r, p + 1:] -= f * M[p, p + 1:]
    return neg


def jacobi_eigenvalues(S, sweeps=30):
    M = np.array(S, dtype=np.float64)
    k = len(M)
    for _ in range(sweeps):
        off = 0.0
        for p in range(k - 1):
            for q in range(p + 1, k):
                apq = M[p, q]
                off += apq * apq
                if apq == 0.0:
                    continue
                theta = (M[q, q] - M[p, p]) / (2 * apq)
                t = np.sign(theta) / (abs(theta) + np.sqrt(theta * theta + 1)) if theta != 0 else 1.0
                c = 1 / np.sqrt(t * t + 1)
                s = t * c
                rp, rq = M[p].copy(), M[q].copy()
                M[p], M[q] = c * rp - s * rq, s * rp + c * rq
                cp, cq = M[:, p].copy(), M[:, q].copy()
                M[:, p], M[:, q] = c * cp - s * cq, s * cp + c * cq
        if off <= 1e-60 or off <= (1e-22 * np.trace(M)) ** 2:
            break
    return np.sort(np.diag(M))


def hestenes_singular_values(A, sweeps=30):
    W = np.array(A, dtype=np.float64)
    k = len(W)
    for _ in range(sweeps):
        rotated = False
        for p in range(k - 1):
            for q in range(p + 1, k):
                al, be, ga = W[p].dot(W[p]), W[q].dot(W[q]), W[p].dot(W[q])
                if ga == 0 or abs(ga) <= 2.3e-16 * np.sqrt(al * be):
                    continue
                rotated = True
                zeta = (be - al) / (2 * ga)
                t = np.sign(zeta) / (abs(zeta) + np.sqrt(1 + zeta * zeta)) if zeta != 0 else 1.0
                c = 1 / np.sqrt(1 + t * t)
                s = c * t
                wp, wq = W[p].copy(), W[q].copy()
                W[p], W[q] = c * wp - s * wq, s * wp + c * wq
        if not rotated:
            break
    return np.sqrt(np.sum(W * W, axis=1))


def bundle_is_rank_deficient(A, counters=None):
    """Device formulation of `matrix_rank(A) < len(A)` (dual :155)."""
    k, n = A.shape
    eps = np.finfo(A.dtype).eps
    c = max(k, n) * eps
    if A.dtype == np.float64:
        sv = hestenes_singular_values(A)
        return bool(np.sum(sv > sv.max() * c) < k)
    A64 = A.astype(np.float64)
    S = A64.dot(A64.T)
    if k == 1:
        return bool(S[0, 0] == 0.0)
    lo = max(np.max(np.diag(S)), S.sum() / k)              # Rayleigh quotients: <= lambda_max
    hi = min(np.trace(S), np.max(np.sum(np.abs(S), axis=1)))   # trace, Gershgorin: >= lambda_max
    c2 = c * c
    if ldl_negative_pivots(S, c2 * hi) == 0:
        if counters is not None:
            counters["fast_full"] = counters.get("fast_full", 0) + 1
        return False                                        # lambda_min > (c sigma_max)^2
    if ldl_negative_pivots(S, c2 * lo) > 0:
        if counters is not None:
            counters["fast_deficient"] = counters.get("fast_deficient", 0) + 1
        return True                                         # lambda_min < (c sigma_max)^2
    if counters is not None:
        counters["jacobi"] = counters.get("jacobi", 0) + 1
    ev = np.maximum(jacobi_eigenvalues(S), 0)
    sv = np.sqrt(ev)
    return bool(np.sum(sv > sv.max() * c) < k)


# --------------------------------------------------------------------------- #
# 2. small dense solve
# --------------------------------------------------------------------------- #
class ExactlySingular(Exception):
    pass


def gepp_solve(M, rhs):
    """Gaussian elimination without pivoting + back substitution (name kept for the call sites)."""
    M = np.array(M, dtype=np.float64)
    x = np.array(rhs, dtype=np.float64)
    m = len(x)
    for p in range(m):
        if M[p, p] == 0.0 or M[p, p] != M[p, p]:
            raise ExactlySingular()
        inv = 1.0 / M[p, p]
        for r in range(p + 1, m):
            f = M[r, p] * inv
            M[r, p + 1:] -= f * M[p, p + 1:]
            x[r] -= f * x[p]
    for r in range(m - 1, -1, -1):
        x[r] = (x[r] - M[r, r + 1:].dot(x[r + 1:])) / M[r, r]
    return x


# --------------------------------------------------------------------------- #
# Newton on the simplex, device formulation
# --------------------------------------------------------------------------- #
CYCLE_TOL = 1e-13
ACCEL_T0, ACCEL_GAP, ACCEL_D2MAX, ACCEL_RMAX = 24, 6, 1e-3, 0.98     # extrapolation of slow 2-cycles (be_dual_dev.h)
# A limit cycle whose own rounding jitter is above CYCLE_TOL (ill-conditioned Newton systems: the iterates repeat to 1e-12, not
# 1e-13) never passes the test above and used to run the full cap -- the same sample, every outer iteration.  Noise floor rule:
# from NOISE_T0 updates on a period p is also accepted when lam_t - lam_{t-p} is below NOISE_TOL and has STOPPED SHRINKING
# (not below its value p updates earlier: a converging sequence shrinks geometrically, a cycle at its rounding floor
# fluctuates).  Variant dual only (the RL variant's cap is 20 updates and its Armijo search does not cycle).
NOISE_TOL, NOISE_T0 = 1e-10, 12
ACCEL_NEG_RESID = 1e-10                     # extrapolation with a negative ratio (below)


def simplex_newton_device(A, b, rules, stats=None):
    k, n = A.shape
    T = A.dtype.type
    c = np.array([np.float64(pairwise_sum(A[i], T)) for i in range(k)]) + b
    A64 = A.astype(np.float64)
    lam = np.ones(k) / k
    prev1 = prev2 = prev3 = prev4 = None
    last_jump = -1000
    done = 0
    result = None
    d4_prev = -1.0          # lam_{t-1} - lam_{t-5} (max norm) of the previous update, -1: not available
    while done < rules.newton_cap:
        a = A64.T.dot(lam)
        z = 1 / (1 + np.exp(-a))
        grad = -c + A64.dot(z)
        hess = (A64 * (z * (1 - z))).dot(A64.T)
        piv = int(np.argmax(lam))
        red = lam.copy()
        red[piv] = 1
        keep = np.ones(k)
        keep[piv] = 0
        col = hess[:, piv]
        g0 = grad - keep * grad[piv]
        h0 = (hess - keep[:, None] * col[None, :] - col[:, None] * keep[None, :]
              + hess[piv, piv] * (keep[:, None] * keep[None, :]))
        bound = (red <= 1e-12) & (g0 > 0)
        bound[piv] = True
        free = ~bound
        if np.sqrt(np.sum(g0[free] ** 2)) < 1e-10:
            result = lam
            break
        step = np.zeros(k)
        try:
            step[free] = gepp_solve(h0[free, :][:, free], -g0[free])
        except ExactlySingular:
            if rules.singular_raises:
                raise np.linalg.LinAlgError("Singular matrix")
            result = lam
            break
        t = min(1. / np.max(abs(step)), 1.) if rules.scaled_first_step else 1.
        fval = None
        if rules.armijo:
            fval = -c.dot(lam) + pairwise_sum(ref.softplus_stable(a), np.float64)
        returned = False
        for _ in range(rules.backoff_cap):
            trial = np.maximum(red + t * step, 0)
            trial[piv] = 1
            lam_new = trial.copy()
            lam_new[piv] = 1. - keep.dot(trial)
            if lam_new[piv] >= 0:
                if rules.armijo:
                    f_new = -c.dot(lam_new) + pairwise_sum(
                        ref.softplus_stable(A64.T.dot(lam_new)), np.float64)
                    if f_new < fval + t * 1e-5 * step.dot(g0):
                        break
                else:
                    break
            if rules.tiny_step_on_td:
                if max(t * abs(step)) < 1e-10:
                    returned = True
                    break
            elif t < 1e-10:
                returned = True
                break
            t *= 0.5
        done += 1
        if returned:
            result = lam_new
            break
        # 3. limit-cycle shortcut
        if CYCLE_TOL > 0 and prev1 is not None and np.max(np.abs(lam_new - prev1)) <= CYCLE_TOL:
            result = lam_new                                              # fixed point
            break
        if CYCLE_TOL > 0 and prev2 is not None and np.max(np.abs(lam_new - prev2)) <= CYCLE_TOL:
            remaining = rules.newton_cap - done                           # period 2
            result = lam_new if remaining % 2 == 0 else prev1
            break
        if CYCLE_TOL > 0 and prev3 is not None and np.max(np.abs(lam_new - prev3)) <= CYCLE_TOL:
            r = (rules.newton_cap - done) % 3                             # period 3: lam_{t+1} = lam_{t-2}
            result = (lam_new, prev2, prev1)[r]
            break
        if CYCLE_TOL > 0 and prev4 is not None and np.max(np.abs(lam_new - prev4)) <= CYCLE_TOL:
            r = (rules.newton_cap - done) % 4                             # period 4: lam_{t+1} = lam_{t-3}
            result = (lam_new, prev3, prev2, prev1)[r]
            break
        # the same periods at their rounding floor (NOISE_TOL, see above); stateless: "p updates earlier" is formed from
        # the four stored iterates (period 3 compares with ONE update earlier, lam_{t-1} - lam_{t-4}: lam_{t-6} is not kept)
        if CYCLE_TOL > 0 and not rules.armijo and done >= NOISE_T0 and prev2 is not None:
            mx = lambda a, b: float(np.max(np.abs(a - b)))                      # noqa: E731
            remaining = rules.newton_cap - done
            d1 = mx(lam_new, prev1)
            if d1 <= NOISE_TOL and d1 >= mx(prev1, prev2):
                result = lam_new
                break
            if prev4 is not None:
                d2 = mx(lam_new, prev2)
                if d2 <= NOISE_TOL and d2 >= mx(prev2, prev4):
                    result = lam_new if remaining % 2 == 0 else prev1
                    break
                d3 = mx(lam_new, prev3)
                if d3 <= NOISE_TOL and d3 >= mx(prev1, prev4):
                    result = (lam_new, prev2, prev1)[remaining % 3]
                    break
                d4 = mx(lam_new, prev4)                                   # period 4 against its own value one update earlier
                if d4 <= NOISE_TOL and d4_prev >= 0 and d4 >= d4_prev:
                    result = (lam_new, prev3, prev2, prev1)[remaining % 4]
                    break
                d4_prev = d4
            else:
                d4_prev = -1.0
        # slow 2-cycles: extrapolate the even and the odd subsequence to their limits (common ratio), go on from there
        if (CYCLE_TOL > 0 and prev4 is not None and done >= ACCEL_T0 and done - last_jump >= ACCEL_GAP):
            d_t, d_p = lam_new - prev2, prev2 - prev4
            n1, n0 = np.max(np.abs(d_t)), np.max(np.abs(d_p))
            if 0 < n1 < ACCEL_D2MAX and n1 < n0:
                ratio = float(d_t.dot(d_p) / d_p.dot(d_p))
                take = 0 < ratio < ACCEL_RMAX
                if -ACCEL_RMAX < ratio < 0:                 # each subsequence oscillates around its limit: only where the
                    left = n1                               # reference's remaining updates would close the gap anyway
                    for _ in range((rules.newton_cap - done) // 2):
                        left = left * -ratio
                    take = left <= ACCEL_NEG_RESID
                if take:
                    gain = ratio / (1 - ratio)
                    xe, xo = lam_new + d_t * gain, prev1 + (prev1 - prev3) * gain
                    if (xe >= 0).all() and (xo >= 0).all():
                        prev4 = prev3 = None
                        d4_prev = -1.0
                        prev2, prev1, lam = xo, xe.copy(), xe.copy()
                        last_jump = done
                        result = lam
                        continue
        prev4, prev3, prev2, prev1 = prev3, prev2, prev1, lam_new.copy()
        lam = lam_new.copy()
        result = lam
    if stats is not None:
        stats.append(done)
    return result


def solve_batch_device(fg, y0, n_iter=None, callback=None, variant="dual", counters=None):
    """Outer loop exactly as oracle.solve_batch, with the device formulations."""
    rules = ref.VARIANTS[variant]
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
        g_t = np.asarray(g_t)
        if G is None:
            G = np.zeros((B, n_iter, n), dtype=g_t.dtype)
        if callback is not None:
            callback(t, f_t, y) if rules.callback_arity == 3 else callback(t, f_t)
        for u in range(B):
            if done[u]:
                continue
            G[u, t] = g_t[u]
            h[u, t] = np.float64(f_t[u]) - pairwise_sum(g_t[u].astype(np.float64) * y[u], np.float64)
            ys[u, t] = y[u]
            slots = active[u] + [t]
            Au = G[u, slots]
            if rules.rank_test and bundle_is_rank_deficient(Au, counters):
                done[u] = True
                n_iters[u] = t - 1
                continue
            before = y[u].copy()
            if len(slots) > 1:
                lam_u = simplex_newton_device(Au, h[u, slots], rules, newton_counts)
                y[u] = 1 / (1 + np.exp(Au.astype(np.float64).T.dot(lam_u)))
            else:
                lam_u = np.array([1.0])
                y[u] = 1 / (1 + np.exp(Au[0]))          # cut-dtype arithmetic, as dual :168
            if rules.clip is not None:
                y[u] = np.clip(y[u], rules.clip[0], rules.clip[1])
            if rules.stall_tol is not None and max(abs(before - y[u])) < rules.stall_tol:
                done[u] = True
            pos = lam_u > 0
            active[u] = [s for s, p in zip(slots, pos) if p]
            lam[u] = lam_u[pos]
        if done.all():
            break
    if G is None:
        G = np.zeros((B, n_iter, n), dtype=np.float32)
    return ref.BundleResult(y, G, h, ys, active, lam, n_iters, done, newton_counts)
