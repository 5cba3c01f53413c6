"""CPU oracle for the RL agent's default inner optimiser `adam()`.
TEST INFRASTRUCTURE ONLY -- nothing under icnn_amd/ imports this module.

Restates RL/src/icnn.py:160-215: projected Adam on `func(obs, act) -> (f[B] float32, g[B,n] float32)` over
act in [-1+1e-8, 1-1e-8]^n from act = 0, keeping the best iterate per sample and stopping when the
exponentially smoothed mean displacement of the best iterates falls under 1e-3 (after more than 5 iterations).
Quirks that parity depends on, all kept:
  * the step divides by sqrt(v), NOT by sqrt(vhat) (:201 -- `vhat` is computed and never used);
  * (1-b1)*g and (1-b2)*(g*g) are float32 products (g is a float32 array, the Python scalars are weak), only
    then promoted to the float64 of m and v (:194-195);
  * the stopping rule looks at the whole batch (:186-189), and the best iterates are returned, not the last.
`entropy_fg` restates `negQ_entr = negQ - entropy(act)` and its gradient (RL/src/icnn.py:59-63, 455-458) around a
negQ closure -- that is `_fg_entr`, the function the agent hands to adam() (:270-272).

Pinned: oracle/gen_golden_adam.py lifts the reference's own `adam` out of RL/src/icnn.py with `ast` (the module
imports TensorFlow, the method itself is pure NumPy) and runs it on seeded convex problems;
tests/golden/adam__*.npz hold its results and tests/test_adam.py compares this restatement bit for bit.
The entropy term is TensorFlow arithmetic in the reference (float32 log, autodiff); its restatement here fixes
one evaluation order (float64 log rounded to float32, sequential float32 sum) that the HIP kernel reproduces
exactly -- unpinned at the TensorFlow boundary like the PICNN itself.
"""
import numpy as np

F32 = np.float32
B1, B2, SMOOTH, EPS, ALPHA = 0.9, 0.999, 0.5, 1e-8, 0.01
BOX = 1. - 1e-8


def adam(func, obs, n_act, max_iter=1000):
    """-> (act_best[B,n] float64, iterations run (== max_iter if the rule never fired), f_best[B])."""
    B = obs.shape[0]
    x = np.zeros((B, n_act))
    mom1 = np.zeros_like(x)
    mom2 = np.zeros_like(x)
    pow1 = pow2 = 1.
    best_x = best_f = None
    drift = None
    for it in range(max_iter):
        f, g = func(obs, x)
        if it == 0:
            best_x, best_f = x.copy(), f.copy()                          # :176-178
        else:
            better = f < best_f                                          # :181
            moved = np.zeros(B)
            moved[better] = np.linalg.norm(x[better] - best_x[better], axis=1)
            best_x[better] = x[better]
            best_f[better] = f[better]
            step_mean = np.mean(moved)                                   # :184
            drift = step_mean if drift is None else SMOOTH * drift + (1. - SMOOTH) * step_mean   # :185-186
            if drift < 1e-3 and it > 5:                                  # :188
                return best_x, it, best_f
        mom1 = B1 * mom1 + (1. - B1) * g                                 # :194  (float32 product, float64 sum)
        mom2 = B2 * mom2 + (1. - B2) * (g * g)                           # :195
        pow1 *= B1
        pow2 *= B2
        mhat = mom1 / (1. - pow1)                                        # :198 (vhat :199 is dead code)
        x = x - ALPHA * mhat / (np.sqrt(mom2) + EPS)                     # :201
        x = np.clip(x, -BOX, BOX)                                        # :203
    return best_x, max_iter, best_f


def entropy_terms(act):
    """Per element, from float32 act: (pen, dpen/dact) of  pen = p log p + (1-p) log(1-p),
    p = clip((act+1)/2, 1e-4, 0.9999)  (RL/src/icnn.py:455-458; clip_by_value passes no gradient outside)."""
    a = np.asarray(act, dtype=F32)
    half = (a + F32(1)) * F32(0.5)
    p = np.minimum(np.maximum(half, F32(1e-4)), F32(0.9999))
    q = F32(1) - p
    lp = np.log(p.astype(np.float64)).astype(F32)
    lq = np.log(q.astype(np.float64)).astype(F32)
    pen = p * lp + q * lq
    inside = (half >= F32(1e-4)) & (half <= F32(0.9999))
    grad = np.where(inside, F32(0.5) * (lp - lq), F32(0))
    return pen.astype(F32), grad.astype(F32)


def entropy_fg(neg_q):
    """`_fg_entr` (RL/src/icnn.py:59-63,131): neg_q(obs, act) -> (f, g) float32 becomes negQ - entropy(act)."""
    def fg(obs, act):
        f, g = neg_q(obs, act)
        pen, dpen = entropy_terms(act)
        tot = np.zeros(pen.shape[0], dtype=F32)
        for j in range(pen.shape[1]):                                    # sequential float32 sum, like the kernel
            tot = tot + pen[:, j]
        return (np.asarray(f, dtype=F32) + tot).astype(F32), (np.asarray(g, dtype=F32) + dpen).astype(F32)
    return fg
