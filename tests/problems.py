"""Seeded convex test problems f_u(y) on the unit box (test infrastructure).

Every problem is built from `numpy.random.RandomState(seed)` only, so the same
inputs are regenerated bit-for-bit on any machine (the GPU box has no access to
the reference checkout; golden vectors were produced from these very inputs by
`oracle/gen_golden.py`).

Each factory returns an object with
    .B, .n, .cut_dtype
    .fg(y)            NumPy: y float64 [B, n] -> (f [B], g [B, n]) in cut_dtype
    .y0()             fresh float64 start point
"""
import numpy as np


class Problem:
    def __init__(self, name, B, n, cut_dtype, fg, y0_value=0.5):
        self.name, self.B, self.n = name, B, n
        self.cut_dtype = np.dtype(cut_dtype)
        self.fg = fg
        self._y0 = y0_value

    def y0(self):
        if np.isscalar(self._y0):
            return np.full((self.B, self.n), float(self._y0))
        return np.array(self._y0, dtype=np.float64, copy=True)


def quadratic(seed=0, B=32, n=4):
    """BASELINE.json configs[0] / SURVEY.md 8(d) "C1 inputs": a fixed random
    convex quadratic, float64 cuts."""
    rng = np.random.RandomState(seed)
    M = rng.randn(n, n)
    Q = M.T.dot(M) + 0.1 * np.eye(n)
    P = rng.randn(B, n)

    def fg(y):
        Qy = y.dot(Q)                       # Q symmetric
        return 0.5 * np.sum(y * Qy, axis=1) + np.sum(P * y, axis=1), Qy + P

    return Problem("quadratic_s%d_B%d_n%d" % (seed, B, n), B, n, np.float64, fg)


def max_affine(seed=1, B=32, n=159, pieces=24, scale=1.0, cut_dtype=np.float32):
    """f_u(y) = max_p a_up . y + b_up  -- piecewise linear like a ReLU ICNN in y.
    Gradients are rows of a stored table, so repeated pieces give bit-identical
    cuts (this is what drives the reference's rank-test termination)."""
    rng = np.random.RandomState(seed)
    a = (rng.randn(B, pieces, n) * scale).astype(cut_dtype)
    centre = rng.rand(B, 1, n)
    # offsets chosen so that the pieces cross inside the box
    b = (-(a.astype(np.float64) * centre).sum(-1) + 0.05 * rng.randn(B, pieces)).astype(cut_dtype)
    a64, b64 = a.astype(np.float64), b.astype(np.float64)
    rows = np.arange(B)

    def fg(y):
        vals = np.einsum("bpn,bn->bp", a64, y) + b64
        best = np.argmax(vals, axis=1)
        return vals[rows, best].astype(cut_dtype), a[rows, best].copy()

    return Problem("maxaffine_s%d_B%d_n%d_p%d" % (seed, B, n, pieces), B, n, cut_dtype, fg)


def log_sum_exp(seed=2, B=32, n=159, pieces=12, scale=0.6, cut_dtype=np.float32):
    """Smooth convex f_u(y) = log sum_p exp(a_up . y + b_up): cuts never repeat,
    so bundles grow until pruning removes them (exercises k up to nIter)."""
    rng = np.random.RandomState(seed)
    a = (rng.randn(B, pieces, n) * scale)
    centre = rng.rand(B, 1, n)
    b = -(a * centre).sum(-1) + 0.3 * rng.randn(B, pieces)

    def fg(y):
        v = np.einsum("bpn,bn->bp", a, y) + b
        m = v.max(axis=1, keepdims=True)
        e = np.exp(v - m)
        s = e.sum(axis=1, keepdims=True)
        f = (m + np.log(s))[:, 0]
        g = np.einsum("bp,bpn->bn", e / s, a)
        return f.astype(cut_dtype), g.astype(cut_dtype)

    return Problem("lse_s%d_B%d_n%d_p%d" % (seed, B, n, pieces), B, n, cut_dtype, fg)


def zero_gradient_rows(seed=3, B=8, n=5, cut_dtype=np.float32):
    """Edge case: some samples have an identically-zero first gradient (rank 0 < 1
    in the reference: the sample finishes at t=0 with an empty bundle, lam None)."""
    base = max_affine(seed, B, n, pieces=6, cut_dtype=cut_dtype)
    dead = np.zeros(B, dtype=bool)
    dead[::3] = True

    def fg(y):
        f, g = base.fg(y)
        f = f.copy()
        g = g.copy()
        g[dead] = 0
        f[dead] = 1.25
        return f, g

    p = Problem("zerograd_s%d_B%d_n%d" % (seed, B, n), B, n, cut_dtype, fg)
    p.dead = dead
    return p


def action_box(seed=4, B=64, n=6, pieces=10, cut_dtype=np.float32):
    """RL-shaped problem (n = dimA = 6): a leaky max-affine
    energy over the action box, seen through the reference's [-1,1] <-> [0,1]
    wrapper (RL/src/icnn.py:148-158: act = 2y-1, gradient doubled)."""
    rng = np.random.RandomState(seed)
    a = rng.randn(B, pieces, n).astype(cut_dtype)
    b = (0.2 * rng.randn(B, pieces)).astype(cut_dtype)
    a64, b64 = a.astype(np.float64), b.astype(np.float64)
    q = (0.3 * np.abs(rng.randn(B, n))).astype(np.float64)
    rows = np.arange(B)

    def fg(y):
        act = 2 * y - 1
        vals = np.einsum("bpn,bn->bp", a64, act) + b64
        best = np.argmax(vals, axis=1)
        f = vals[rows, best] + 0.5 * np.sum(q * act * act, axis=1)
        g = a64[rows, best] + q * act
        return f.astype(cut_dtype), (2 * g).astype(cut_dtype)

    return Problem("actionbox_s%d_B%d_n%d" % (seed, B, n), B, n, cut_dtype, fg)


# The fixtures under tests/golden/ were generated for exactly these instances.
GOLDEN_CASES = {
    "c1_quadratic": (lambda: quadratic(0, 32, 4), 10),
    "maxaffine_n159": (lambda: max_affine(1, 32, 159, 24, 1.0), 10),
    "maxaffine_n159_long": (lambda: max_affine(11, 16, 159, 48, 0.5), 30),
    "maxaffine_f64": (lambda: max_affine(5, 16, 40, 16, 1.0, np.float64), 10),
    "lse_n159": (lambda: log_sum_exp(2, 32, 159, 12, 0.6), 10),
    "lse_n33": (lambda: log_sum_exp(7, 24, 33, 8, 1.0), 12),
    "zero_gradient": (lambda: zero_gradient_rows(3, 8, 5), 6),
    "action_box": (lambda: action_box(4, 64, 6, 10), 5),
    "single_sample": (lambda: max_affine(9, 1, 7, 5, 1.5), 8),
    "n_equals_1": (lambda: log_sum_exp(10, 16, 1, 4, 2.0), 6),
}


# ---- problems for the RL agent's Adam inner optimiser: func(obs, act) -> (f float32 [B], g float32 [B, n]) ------
def adam_quadratic(seed, B, n, shift=0.3, curv=1.0):
    """Convex quadratic in the action, coefficients depend on the observation; float32 like a TensorFlow fetch.
    `shift` moves the unconstrained minimisers (|shift| > 1: outside the box, the clip at +-(1-1e-8) binds)."""
    rng = np.random.RandomState(seed)
    M = rng.randn(n, n)
    Q = (curv * (M.T.dot(M) / n + 0.2 * np.eye(n))).astype(np.float32)
    obs = rng.randn(B, 3).astype(np.float32)
    centre = (shift * np.tanh(obs.dot(rng.randn(3, n)))).astype(np.float32)

    def neg_q(o, act):
        d = act.astype(np.float32) - centre
        Qd = d.dot(Q)
        return (np.float32(0.5) * np.sum(d * Qd, axis=1)).astype(np.float32), Qd.astype(np.float32)

    return obs, n, neg_q


ADAM_CASES = {
    "quad_b1_n6": lambda: adam_quadratic(21, 1, 6),
    "quad_b32_n6": lambda: adam_quadratic(22, 32, 6),
    "quad_b5_n17_clipped": lambda: adam_quadratic(23, 5, 17, shift=1.6),
    "quad_b3_n2_flat": lambda: adam_quadratic(24, 3, 2, shift=0.0, curv=1e-3),
    "quad_b4_n3_stiff": lambda: adam_quadratic(25, 4, 3, shift=0.9, curv=40.0),
    "quad_b2_n4_walls": lambda: adam_quadratic(26, 2, 4, shift=3.0),
}
