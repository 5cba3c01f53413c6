"""CPU oracle for the fully-connected PICNN energy E(x, y) and dE/dy.
TEST INFRASTRUCTURE ONLY -- never imported by `icnn_amd/`.

Parity status: UNPINNED at the TensorFlow/tflearn boundary.  The reference builds
this function out of tflearn layers on TensorFlow r0.10; neither package is in
the reference checkout or installable here, and the reference has no tests or
stored activations for it.  What is restated below is the layer algebra spelled
out at the reference's call sites:

  multi-label-cls/icnn_ebundle.py:316-388   Model.f   (ReLU, BatchNorm on u-path)
  RL/src/icnn.py:325-404                    negQ      (leaky ReLU, BatchNorm optional)

with the third-party semantics assumed as: `fully_connected` = x @ W[in,out] + b,
`batch_normalization` in training mode = (x-mean)/sqrt(biased_var+1e-5)*gamma+beta
over the batch axis, `tf.gradients` = exact reverse-mode derivative.  The oracle
is checked for self-consistency (autograd, convexity in y) in tests/, and the HIP
kernels are checked against it.

Parameters are a plain dict of float32 NumPy arrays keyed by the reference's
variable-scope names: 'u{i}/W', 'u{i}/b', 'u{i}/bn/gamma', 'u{i}/bn/beta',
'z{i}_zu_u/W', 'z{i}_zu_u/b', 'z{i}_zu_proj/W', 'z{i}_yu_u/W', 'z{i}_yu_u/b',
'z{i}_yu/W', 'z{i}_u/W', 'z{i}_u/b'.  Layer widths: sizes = szs + [1].
"""
import numpy as np

F32 = np.float32


def _act(v, alpha):
    return np.where(v > 0, v, F32(alpha) * v).astype(F32)


def _dact(v, alpha):
    return np.where(v > 0, F32(1), F32(alpha)).astype(F32)


def u_path(params, x, n_hidden, batchnorm=True, eps=1e-5):
    """x-only trunk: u_i = BN(relu(fc(u_{i-1}))) for i < L-1, u_{L-1} = fc(u_{L-2})
    (icnn_ebundle.py:339-347; RL/src/icnn.py:345-354)."""
    us, prev = [], x.astype(F32)
    for i in range(n_hidden):
        u = prev.dot(params["u%d/W" % i]) + params["u%d/b" % i]
        if i < n_hidden - 1:
            u = np.maximum(u, F32(0))
            if batchnorm:
                mean = u.mean(axis=0, dtype=F32)
                var = ((u - mean) ** 2).mean(axis=0, dtype=F32)   # biased, as tf.nn.moments
                u = (u - mean) / np.sqrt(var + F32(eps)) * params["u%d/bn/gamma" % i] \
                    + params["u%d/bn/beta" % i]
        us.append(u.astype(F32))
        prev = us[-1]
    return us


def context(params, x, szs, batchnorm=True):
    """Everything in E(x, y) that does not depend on y, per layer i = 0..L:
         yu_i   = fc(prevU -> n)            multiplies y elementwise   (:363-365)
         zu_i   = fc(prevU -> s_i)          additive term              (:372-373)
         gate_i = relu(fc(prevU -> s_{i-1}))  multiplies z_{i-1}, i>0  (:354-356)
    with prevU = x for i = 0 and u_{i-1} afterwards (:349, :384)."""
    L = len(szs)
    us = u_path(params, x, L, batchnorm)
    layers = []
    for i in range(L + 1):
        prev = x.astype(F32) if i == 0 else us[i - 1]
        yu = prev.dot(params["z%d_yu_u/W" % i]) + params["z%d_yu_u/b" % i]
        zu = prev.dot(params["z%d_u/W" % i]) + params["z%d_u/b" % i]
        gate = None
        if i > 0:
            gate = np.maximum(prev.dot(params["z%d_zu_u/W" % i]) + params["z%d_zu_u/b" % i], F32(0))
        layers.append(dict(yu=yu.astype(F32), zu=zu.astype(F32),
                           gate=None if gate is None else gate.astype(F32)))
    return layers


def energy_and_grad(params, ctx, y, szs, alpha=0.0):
    """E[B] and dE/dy[B, n] in float32 for y float32 [B, n] (:349-388 + :146).

    alpha = 0 is the multi-label model's ReLU; alpha = FLAGS.lrelu for negQ."""
    L = len(szs)
    y = y.astype(F32)
    pre, zs = [], []
    z_prev = None
    for i in range(L + 1):
        c = ctx[i]
        p = (y * c["yu"]).dot(params["z%d_yu/W" % i]) + c["zu"]
        if i > 0:
            p = p + (z_prev * c["gate"]).dot(params["z%d_zu_proj/W" % i])
        p = p.astype(F32)
        pre.append(p)
        z_prev = _act(p, alpha) if i < L else p
        zs.append(z_prev)
    E = zs[-1].reshape(-1)

    delta = np.ones_like(pre[L])                       # dE/d pre_L
    gy = np.zeros_like(y)
    for i in range(L, -1, -1):
        c = ctx[i]
        gy += c["yu"] * delta.dot(params["z%d_yu/W" % i].T)
        if i > 0:
            dz = c["gate"] * delta.dot(params["z%d_zu_proj/W" % i].T)
            delta = (dz * _dact(pre[i - 1], alpha)).astype(F32)
    return E.astype(F32), gy.astype(F32)


def make_fg(params, x, szs, alpha=0.0, batchnorm=True, box=None):
    """The closure the reference training loop hands to solveBatch
    (icnn_ebundle.py:218-221).  TensorFlow recomputes the x-only part on every
    call; it is deterministic, so computing it once gives the same numbers.

    box="action": the RL wrapper (RL/src/icnn.py:148-158) -- the network sees
    2y-1 and the gradient is doubled."""
    ctx = context(params, x, szs, batchnorm)

    def fg(y):
        if box == "action":
            # 2y-1 is formed in float64 by the caller and rounded once on the feed
            act = (2 * np.asarray(y, dtype=np.float64) - 1).astype(F32)
            E, g = energy_and_grad(params, ctx, act, szs, alpha)
            return E, (F32(2) * g).astype(F32)
        return energy_and_grad(params, ctx, np.asarray(y).astype(F32), szs, alpha)

    fg.ctx = ctx
    return fg


def flat_context(ctx):
    """[B, C] float32 in the order the HIP kernels read it (include/icnn_be.h):
    for each layer i: yu_i (n) | zu_i (s_i) | gate_i (s_{i-1}, i > 0)."""
    parts = []
    for c in ctx:
        parts += [c["yu"], c["zu"]]
        if c["gate"] is not None:
            parts.append(c["gate"])
    return np.ascontiguousarray(np.concatenate(parts, axis=1), dtype=F32)


def unflatten_context(flat, n, widths):
    """Inverse of flat_context: [B, C] -> per-layer dicts (widths = szs + [1])."""
    layers, o = [], 0
    for i, w in enumerate(widths):
        yu = flat[:, o:o + n]; o += n
        zu = flat[:, o:o + w]; o += w
        gate = None
        if i > 0:
            gate = flat[:, o:o + widths[i - 1]]; o += widths[i - 1]
        layers.append(dict(yu=np.ascontiguousarray(yu), zu=np.ascontiguousarray(zu),
                           gate=None if gate is None else np.ascontiguousarray(gate)))
    assert o == flat.shape[1]
    return layers


def make_fg_from_context(params, flat_ctx, szs, alpha=0.0, box=None):
    """fg closure over a given flat context (used to drive the oracle with exactly the
    context rows the device kernels read)."""
    n = params["z0_yu/W"].shape[0]
    ctx = unflatten_context(np.asarray(flat_ctx, dtype=F32), n, list(szs) + [1])

    def fg(y):
        if box == "action":
            act = (2 * np.asarray(y, dtype=np.float64) - 1).astype(F32)
            E, g = energy_and_grad(params, ctx, act, szs, alpha)
            return E, (F32(2) * g).astype(F32)
        return energy_and_grad(params, ctx, np.asarray(y).astype(F32), szs, alpha)

    fg.ctx = ctx
    return fg


# --------------------------------------------------------------------------------------- #
# MFMA-order evaluation (oracle/picnn_chain.c): the same network, float32 dot products
# accumulated as the k-ordered fused-multiply-add chain of v_mfma_f32_16x16x4_f32.
# --------------------------------------------------------------------------------------- #
_chain_lib = None


def chain_lib():
    """Load (building if necessary) oracle/_build/libpicnn_chain.so."""
    global _chain_lib
    if _chain_lib is None:
        import ctypes
        import os
        import subprocess
        here = os.path.dirname(os.path.abspath(__file__))
        so = os.path.join(here, "_build", "libpicnn_chain.so")
        src = os.path.join(here, "picnn_chain.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.run(["make", "-C", here], check=True, stdout=subprocess.DEVNULL)
        _chain_lib = ctypes.CDLL(so)
    return _chain_lib


def energy_and_grad_chain(params, flat_ctx, y, szs, alpha=0.0, action_box=False):
    """E[B], dE/dy[B, n] in float32, MFMA accumulation order; `y` is float64 (rounded like a feed)."""
    import ctypes as C
    lib = chain_lib()
    widths = list(szs) + [1]
    L1 = len(widths)
    flat_ctx = np.ascontiguousarray(flat_ctx, dtype=F32)
    y = np.ascontiguousarray(y, dtype=np.float64)
    B, n = y.shape
    keep = []

    def ptr(a):
        a = np.ascontiguousarray(a, dtype=F32)
        keep.append(a)
        return a.ctypes.data

    w_yu = (C.c_void_p * L1)(*[ptr(params["z%d_yu/W" % i]) for i in range(L1)])
    w_zu = (C.c_void_p * L1)(*([None] + [ptr(params["z%d_zu_proj/W" % i]) for i in range(1, L1)]))
    wid = (C.c_int * L1)(*widths)
    E = np.empty(B, dtype=F32)
    g = np.empty((B, n), dtype=F32)
    lib.picnn_chain_fg(C.c_int(B), C.c_int(n), C.c_int(L1), wid, C.c_float(alpha), C.c_int(int(action_box)),
                       C.c_void_p(flat_ctx.ctypes.data), C.c_int(flat_ctx.shape[1]), w_yu, w_zu,
                       C.c_void_p(y.ctypes.data), C.c_void_p(E.ctypes.data), C.c_void_p(g.ctypes.data))
    return E, g


def make_fg_chain(params, flat_ctx, szs, alpha=0.0, action_box=False):
    """fg closure evaluating the PICNN in MFMA order on a given flat context."""
    flat_ctx = np.ascontiguousarray(flat_ctx, dtype=F32)

    def fg(y):
        return energy_and_grad_chain(params, flat_ctx, y, szs, alpha, action_box)

    return fg


def context_rows_chain(params, x, szs, stage_weights):
    """Flat context rows [B, C] of a model WITHOUT BatchNorm with every dot product as the k-ascending fma chain the
    one-launch act() path applies (oracle/picnn_chain.c: picnn_context_rows_chain).  `stage_weights` is the list of
    (W_stage, b_stage) the device consumes (icnn_amd.picnn.stage_weights: host packing, passed in by the test)."""
    import ctypes as C
    lib = chain_lib()
    widths = list(szs) + [1]
    L1 = len(widths)
    x = np.ascontiguousarray(x, dtype=F32)
    B, n_features = x.shape
    n = params["z0_yu/W"].shape[0]
    keep = [np.ascontiguousarray(a, dtype=F32) for pair in stage_weights for a in pair]
    Ws = (C.c_void_p * L1)(*[keep[2 * i].ctypes.data for i in range(L1)])
    bs = (C.c_void_p * L1)(*[keep[2 * i + 1].ctypes.data for i in range(L1)])
    width = (C.c_int * L1)(*widths)
    Cw = sum(n + widths[i] + (widths[i - 1] if i > 0 else 0) for i in range(L1))
    ctx = np.zeros((B, Cw), dtype=F32)
    lib.picnn_context_rows_chain(C.c_int(B), C.c_int(n_features), C.c_int(n), C.c_int(L1), width, Ws, bs,
                                 C.c_void_p(x.ctypes.data), C.c_void_p(ctx.ctypes.data), C.c_int(Cw))
    return ctx


# --------------------------------------------------------------------------------------- #
# Float64 evaluation of the SAME float32-parameter network ("truth" for rounding-error comparisons) and a third,
# unrelated float32 summation order (products rounded to float32, then NumPy's pairwise tree: no fused multiply-add,
# no k-ordered chain) -- tests/test_sensitivity.py, tests/test_gpu_parity.py (round 4).
def energy_and_grad_f64(params, ctx, y, szs, alpha=0.0):
    """E[B], dE/dy[B, n] of energy_and_grad with every operation in float64 (parameters, context and y are the float32
    values, promoted): what the float32 evaluations approximate."""
    D = np.float64
    L = len(szs)
    y = np.asarray(y, dtype=F32).astype(D)
    pre, z_prev = [], None
    for i in range(L + 1):
        c = ctx[i]
        p = (y * c["yu"].astype(D)).dot(params["z%d_yu/W" % i].astype(D)) + c["zu"].astype(D)
        if i > 0:
            p = p + (z_prev * c["gate"].astype(D)).dot(params["z%d_zu_proj/W" % i].astype(D))
        pre.append(p)
        z_prev = np.where(p > 0, p, alpha * p) if i < L else p
    E = z_prev.reshape(-1)
    delta = np.ones_like(pre[L])
    gy = np.zeros_like(y)
    for i in range(L, -1, -1):
        c = ctx[i]
        gy += c["yu"].astype(D) * delta.dot(params["z%d_yu/W" % i].astype(D).T)
        if i > 0:
            dz = c["gate"].astype(D) * delta.dot(params["z%d_zu_proj/W" % i].astype(D).T)
            delta = dz * np.where(pre[i - 1] > 0, 1.0, alpha)
    return E, gy


def _dot_pairwise(a, W):
    """a[B, K] @ W[K, N] in float32 with every product rounded to float32 and the K products of an output summed by
    NumPy's pairwise tree (np.add.reduce over a contiguous float32 axis)."""
    prod = (a[:, None, :] * np.ascontiguousarray(W.T)[None, :, :]).astype(F32)      # [B, N, K], K contiguous
    return np.add.reduce(prod, axis=2, dtype=F32)


def energy_and_grad_pairwise(params, ctx, y, szs, alpha=0.0):
    """energy_and_grad with the dot products in the pairwise order of _dot_pairwise."""
    L = len(szs)
    y = np.asarray(y).astype(F32)
    pre, z_prev = [], None
    for i in range(L + 1):
        c = ctx[i]
        p = _dot_pairwise((y * c["yu"]).astype(F32), params["z%d_yu/W" % i]) + c["zu"]
        if i > 0:
            p = p + _dot_pairwise((z_prev * c["gate"]).astype(F32), params["z%d_zu_proj/W" % i])
        p = p.astype(F32)
        pre.append(p)
        z_prev = _act(p, alpha) if i < L else p
    E = z_prev.reshape(-1)
    delta = np.ones_like(pre[L])
    gy = np.zeros_like(y)
    for i in range(L, -1, -1):
        c = ctx[i]
        gy += c["yu"] * _dot_pairwise(delta, np.ascontiguousarray(params["z%d_yu/W" % i].T))
        if i > 0:
            dz = c["gate"] * _dot_pairwise(delta, np.ascontiguousarray(params["z%d_zu_proj/W" % i].T))
            delta = (dz * _dact(pre[i - 1], alpha)).astype(F32)
    return E.astype(F32), gy.astype(F32)


def make_fg_pairwise(params, flat_ctx, szs, alpha=0.0):
    """fg closure over a flat context with the pairwise-order PICNN (a third instance of "the reference's float32 fg")."""
    n = params["z0_yu/W"].shape[0]
    ctx = unflatten_context(np.asarray(flat_ctx, dtype=F32), n, list(szs) + [1])

    def fg(y):
        return energy_and_grad_pairwise(params, ctx, np.asarray(y).astype(F32), szs, alpha)

    fg.ctx = ctx
    return fg
