"""CPU oracle for the convolutional PICNN of the image-completion experiment.
TEST INFRASTRUCTURE ONLY -- never imported by `icnn_amd/`.

Restates completion/icnn_ebundle.py:337-452 (`Model.f`) and :118-121 (`dE_dyFlat_` =
flatten(tf.gradients(E_, y_))) with torch CPU float32 ops; the y-gradient comes from torch
autograd, i.e. the exact reverse-mode derivative tf.gradients computes.

Parity status: UNPINNED at the TensorFlow/tflearn boundary (third-party, absent, no stored
activations in the reference).  Assumed semantics: `conv_2d` = NHWC cross-correlation with
'SAME' padding (for the kernel/stride pairs used -- 8/4, 4/2, 3/1 -- SAME is symmetric:
2, 1, 1 pixels), W stored [k, k, in, out]; `fully_connected` = x @ W[in,out] + b;
`batch_normalization` (training mode, as the reference runs it) = batch statistics over
N,H,W (conv) or N (fc), biased variance, eps 1e-5; tf `flatten` = NHWC row-major.

Architecture (reference lines): u-path :349-372 -- three conv+ReLU+BN (32 k8 s4, 64 k4 s2,
64 k3 s1), fc 512 + ReLU + BN, fc 1.  z-path conv layers :376-409, fc layers :411-445
(the y passthrough of the fc layers is commented out in the reference, :425-434).
Parameters: dict of float32 arrays keyed by the reference's variable-scope names.
"""
import numpy as np
import torch
import torch.nn.functional as F

CONVS = [(32, 8, 4), (64, 4, 2), (64, 3, 1)]     # (filters, kernel, stride)  :344
FCS = [512, 1]                                   # :345


def _t(a):
    return torch.as_tensor(np.asarray(a), dtype=torch.float32)


def _conv(x_nhwc, W, b, stride):
    """NHWC cross-correlation, TF 'SAME' padding; W is [k, k, in, out]."""
    k = W.shape[0]
    pad = {8: 2, 4: 1, 3: 1}[k] if stride > 1 or k == 3 else k // 2
    out = F.conv2d(x_nhwc.permute(0, 3, 1, 2), W.permute(3, 2, 0, 1), None if b is None else b,
                   stride=stride, padding=pad)
    return out.permute(0, 2, 3, 1)


def _bn(x, gamma, beta, dims, eps=1e-5):
    mean = x.mean(dim=dims, keepdim=True)
    var = ((x - mean) ** 2).mean(dim=dims, keepdim=True)
    return (x - mean) / torch.sqrt(var + eps) * gamma + beta


def u_path(p, x):
    """x: [B, H, W, 1] float32 tensor -> list us (five entries, :349-372)."""
    us, prev = [], x
    for l, (nf, k, s) in enumerate(CONVS):
        u = torch.relu(_conv(prev, _t(p["u%d/W" % l]), _t(p["u%d/b" % l]), s))
        u = _bn(u, _t(p["u%d/bn/gamma" % l]), _t(p["u%d/bn/beta" % l]), (0, 1, 2))
        us.append(u)
        prev = u
    prev = prev.reshape(prev.shape[0], -1)
    u3 = torch.relu(prev @ _t(p["u3/W"]) + _t(p["u3/b"]))
    u3 = _bn(u3, _t(p["u3/bn/gamma"]), _t(p["u3/bn/beta"]), (0,))
    us.append(u3)
    us.append((u3 @ _t(p["u4/W"]) + _t(p["u4/b"])).reshape(-1))
    return us


def context(p, x):
    """Everything in E(x, y) that does not depend on y, as a dict of NHWC tensors."""
    us = u_path(p, x)
    c = {}
    prevU = x
    for l, (nf, k, s) in enumerate(CONVS):
        if l > 0:
            c["gate%d" % l] = torch.relu(_conv(prevU, _t(p["z%d_zu_u/W" % l]), _t(p["z%d_zu_u/b" % l]), 1))
        c["yu%d" % l] = _conv(prevU, _t(p["z%d_yu_u/W" % l]), _t(p["z%d_yu_u/b" % l]), 1)
        c["zu%d" % l] = _conv(prevU, _t(p["z%d_u/W" % l]), _t(p["z%d_u/b" % l]), s)
        prevU = us[l]
    prevU = prevU.reshape(prevU.shape[0], -1)
    for l, sz in zip((3, 4), FCS):
        c["gate%d" % l] = torch.relu(prevU @ _t(p["z%d_zu_u/W" % l]) + _t(p["z%d_zu_u/b" % l]))
        c["zu%d" % l] = prevU @ _t(p["z%d_u/W" % l]) + _t(p["z%d_u/b" % l])
        prevU = us[l]
    return c


def energy(p, c, y):
    """E[B] for y [B, H, W, 1] (torch tensor, may require grad)."""
    y_red, z = y, None
    for l, (nf, k, s) in enumerate(CONVS):
        acc = _conv(y_red * c["yu%d" % l], _t(p["z%d_yu/W" % l]), None, s) + c["zu%d" % l]
        if l > 0:
            acc = acc + _conv(z * c["gate%d" % l], _t(p["z%d_zu_proj/W" % l]), None, s)
        y_red = _conv(y_red, _t(p["z%d_y_red/W" % l]), _t(p["z%d_y_red/b" % l]), s)
        z = torch.relu(acc)
    z = z.reshape(z.shape[0], -1)
    z = torch.relu((z * c["gate3"]) @ _t(p["z3_zu_proj/W"]) + c["zu3"])
    z = (z * c["gate4"]) @ _t(p["z4_zu_proj/W"]) + c["zu4"]
    return z.reshape(-1)


def flat_context(c):
    """[B, C] float32 in the order the HIP kernel reads it (include/icnn_be.h, icnn_be_conv_model)."""
    order = ["yu0", "zu0", "gate1", "yu1", "zu1", "gate2", "yu2", "zu2", "gate3", "zu3", "gate4", "zu4"]
    B = c["yu0"].shape[0]
    return np.ascontiguousarray(torch.cat([c[k].reshape(B, -1) for k in order], dim=1).numpy())


def unflatten_context(flat, H, W):
    """Inverse of flat_context for the reference architecture on an H x W image."""
    flat = torch.as_tensor(flat, dtype=torch.float32)
    B = flat.shape[0]
    shapes, h, w, cin = {}, H, W, 1
    dims = []
    for l, (nf, k, s) in enumerate(CONVS):
        if l > 0:
            dims.append(("gate%d" % l, (h, w, cin)))
        dims.append(("yu%d" % l, (h, w, 1)))
        h, w = (h + s - 1) // s, (w + s - 1) // s
        dims.append(("zu%d" % l, (h, w, nf)))
        cin = nf
    flatdim = h * w * cin
    dims += [("gate3", (flatdim,)), ("zu3", (FCS[0],)), ("gate4", (FCS[0],)), ("zu4", (1,))]
    order = ["yu0", "zu0", "gate1", "yu1", "zu1", "gate2", "yu2", "zu2", "gate3", "zu3", "gate4", "zu4"]
    shapes = dict(dims)
    c, o = {}, 0
    for k_ in order:
        sz = int(np.prod(shapes[k_]))
        c[k_] = flat[:, o:o + sz].reshape((B,) + tuple(shapes[k_]))
        o += sz
    assert o == flat.shape[1]
    return c


def make_fg_from_context(p, flat_ctx, H, W):
    """fg closure of completion/icnn_ebundle.py:217-221: y arrives flat [B, H*W] (float64), is
    reshaped to [B, H, W, 1] and fed as float32; returns E[B] and the flattened gradient."""
    c = unflatten_context(flat_ctx, H, W)

    def fg(y):
        yt = torch.as_tensor(np.asarray(y, dtype=np.float64).astype(np.float32)).reshape(-1, H, W, 1)
        yt.requires_grad_(True)
        E = energy(p, c, yt)
        g, = torch.autograd.grad(E.sum(), yt)
        return E.detach().numpy().astype(np.float32), g.reshape(g.shape[0], -1).numpy().astype(np.float32)

    return fg


# --------------------------------------------------------------------------------------- #
# Kernel-order evaluation (oracle/picnn_conv_chain.c): the same network with every float32 sum in
# the order conv_fg_kernel applies, so that the HIP path can be compared bit for bit.
# --------------------------------------------------------------------------------------- #
_conv_chain_lib = None


def conv_chain_lib():
    """Load (building if necessary) oracle/_build/libpicnn_conv_chain.so."""
    global _conv_chain_lib
    if _conv_chain_lib is None:
        import ctypes
        import os
        import subprocess
        here = os.path.dirname(os.path.abspath(__file__))
        so = os.path.join(here, "_build", "libpicnn_conv_chain.so")
        src = os.path.join(here, "picnn_conv_chain.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.run(["make", "-C", here], check=True, stdout=subprocess.DEVNULL)
        _conv_chain_lib = ctypes.CDLL(so)
    return _conv_chain_lib


def energy_and_grad_chain(p, flat_ctx, y, H, W):
    """E[B], dE/dy[B, H*W] in float32, kernel accumulation order; `y` flat float64 (rounded like a feed)."""
    import ctypes as C
    lib = conv_chain_lib()
    F32 = np.float32
    flat_ctx = np.ascontiguousarray(flat_ctx, dtype=F32)
    y = np.ascontiguousarray(y, dtype=np.float64)
    B, n = y.shape
    assert n == H * W
    keep = []

    def ptr(a):
        a = np.ascontiguousarray(a, dtype=F32)
        keep.append(a)
        return a.ctypes.data

    w_yu = (C.c_void_p * 3)(*[ptr(p["z%d_yu/W" % l]) for l in range(3)])
    w_yr = (C.c_void_p * 3)(*([ptr(p["z%d_y_red/W" % l]) for l in range(2)] + [None]))
    b_yr = (C.c_void_p * 3)(*([ptr(p["z%d_y_red/b" % l]) for l in range(2)] + [None]))
    w_zu = (C.c_void_p * 3)(*([None] + [ptr(p["z%d_zu_proj/W" % l]) for l in (1, 2)]))
    Fs = (C.c_int * 3)(*[c[0] for c in CONVS])
    Ks = (C.c_int * 3)(*[c[1] for c in CONVS])
    Ss = (C.c_int * 3)(*[c[2] for c in CONVS])
    E = np.empty(B, dtype=F32)
    g = np.empty((B, n), dtype=F32)
    lib.picnn_conv_chain_fg(C.c_int(B), C.c_int(H), C.c_int(W), Fs, Ks, Ss, C.c_int(FCS[0]),
                            C.c_void_p(flat_ctx.ctypes.data), C.c_int(flat_ctx.shape[1]), w_yu, w_yr, b_yr, w_zu,
                            C.c_void_p(ptr(p["z3_zu_proj/W"])), C.c_void_p(ptr(p["z4_zu_proj/W"])),
                            C.c_void_p(y.ctypes.data), C.c_void_p(E.ctypes.data), C.c_void_p(g.ctypes.data))
    return E, g


def make_fg_chain(p, flat_ctx, H, W):
    """fg closure evaluating the conv PICNN in the kernel's accumulation order on a given flat context."""
    flat_ctx = np.ascontiguousarray(flat_ctx, dtype=np.float32)

    def fg(y):
        return energy_and_grad_chain(p, flat_ctx, y, H, W)

    return fg
