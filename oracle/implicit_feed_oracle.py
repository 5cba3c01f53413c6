"""CPU oracle for the implicit-differentiation feed that consumes the solver's output in a
training step.  TEST INFRASTRUCTURE ONLY.

Restates
  multi-label-cls/icnn_ebundle.py:296-314  train_step_fd   and :390-417 crossEntrGrad
  completion/icnn_ebundle.py:315-335       train_step_fd   and :493-522 mseGrad
For every sample j with a non-empty bundle G_j (k x n), multipliers lam_j and minimiser y_j:
    Z^-1 = 1 / (1/y + 1/(1-y)),   dl = d loss / d y,
    [[G Z^-1 G^T, 1], [1^T, 0]] [c_lam; c_t] = [G Z^-1 dl; 0],
    c_y = Z^-1 dl - (G Z^-1)^T c_lam,   c_y = 0 where y is exactly 0 or 1,
and one output row per active cut i:  (sample j, y = ys_{j,i}, v = lam_i c_y + c_lam,i (y_j - ys_{j,i}), c = c_lam,i).
Pinned: oracle/gen_golden_feed.py executes the reference's own crossEntrGrad / mseGrad (extracted
from the scripts' source with `ast`, since the scripts import TensorFlow at module level) and stores
their results in tests/golden/feed__*.npz.
"""
import numpy as np


def kkt_grad(y, true_y, G, loss):
    """(c_y, c_lam, c_t) -- crossEntrGrad (loss='xent', :390-417) / mseGrad (loss='mse', :493-522)."""
    k, n = G.shape
    if loss == "xent":
        yc = np.clip(np.copy(y), 1e-8, 1. - 1e-8)          # :393-395
        z = 1. / yc + 1. / (1. - yc)                       # :406
        dl = true_y / yc - (1 - true_y) / (1 - yc)         # :411
    elif loss == "mse":
        with np.errstate(divide="ignore"):
            z = 1. / y + 1. / (1. - y)                     # completion :508 (the masked version above it is overwritten)
        dl = -(y - true_y)                                 # :515
    else:
        raise ValueError(loss)
    zinv = 1. / z
    Gz = G * zinv
    H = np.block([[Gz.dot(G.T), np.ones((k, 1))], [np.ones((1, k)), np.zeros((1, 1))]])
    rhs = np.concatenate([Gz.dot(dl), np.zeros(1)])
    sol = np.linalg.solve(H, rhs)
    c_lam, c_t = sol[:k], sol[k:]
    c_y = zinv * dl - Gz.T.dot(c_lam)
    c_y[(y == 0) | (y == 1)] = 0
    return c_y, c_lam, c_t


def feed_rows(y_n, true_y, G, ys, lam, loss):
    """Rows of the training feed: (sample index, y rows, v rows, c) -- train_step_fd :296-314 / :315-335."""
    idx, rows_y, rows_v, rows_c = [], [], [], []
    for j in range(len(G)):
        if len(G[j]) == 0:                                  # completion :319-320
            continue
        c_y, c_lam, _ = kkt_grad(y_n[j], true_y[j], np.array(G[j]), loss)
        for i in range(len(G[j])):
            idx.append(j)
            rows_y.append(ys[j][i])
            rows_v.append(lam[j][i] * c_y + c_lam[i] * (y_n[j] - ys[j][i]))
            rows_c.append(c_lam[i])
    n = y_n.shape[1]
    return (np.array(idx, dtype=np.int64), np.array(rows_y).reshape(-1, n), np.array(rows_v).reshape(-1, n),
            np.array(rows_c))
