"""Host side of the fully-connected PICNN energy E(x, y).

The bundle-entropy hot path only ever needs the y-dependent part of the network;
everything that depends on x alone is constant across bundle iterations.  This
module therefore splits the reference model

    multi-label-cls/icnn_ebundle.py:316-388  Model.f      (szs=[600, 159], ReLU, BN)
    RL/src/icnn.py:325-404                   negQ         (szs=[200, 200], leaky ReLU)

into (a) the *context* -- yu_i, zu_i, gate_i per layer, a [B, C] float32 tensor
computed once per minibatch with ordinary torch GEMMs (plumbing, not the hot
path) -- and (b) the *y-path weights* 'z{i}_yu/W' and
'z{i}_zu_proj/W', which the HIP kernels stream every bundle iteration.

Parameters are a dict of float32 arrays keyed by the reference's variable-scope
names ('u0/W', 'z1_zu_proj/W', ...), W stored [in, out] as tflearn does.
"""
from dataclasses import dataclass
from typing import Dict, List

import numpy as np
import torch


@dataclass(frozen=True)
class FCSpec:
    """Shape of one FC-PICNN.  `szs` are the hidden widths exactly as the
    reference passes them (icnn_ebundle.py:321-323 appends nLabels itself)."""
    n_features: int
    n_labels: int
    szs: tuple
    alpha: float = 0.0        # 0 -> ReLU (multi-label); FLAGS.lrelu for negQ
    batchnorm: bool = True    # u-path BN in batch-statistics mode (icnn_ebundle.py:209,259)
    action_box: bool = False  # RL wrapper: network sees 2y-1, gradient doubled (icnn.py:148-158)

    @property
    def widths(self) -> List[int]:
        """s_0 .. s_L with the final scalar layer appended (:351)."""
        return list(self.szs) + [1]

    @property
    def n_layers(self) -> int:
        return len(self.szs) + 1

    @property
    def ctx_offsets(self):
        """Per layer (yu_off, zu_off, gate_off) into a context row; gate_off=-1 for layer 0."""
        offs, o = [], 0
        w = self.widths
        for i in range(self.n_layers):
            yu = o
            o += self.n_labels
            zu = o
            o += w[i]
            gate = -1
            if i > 0:
                gate = o
                o += w[i - 1]
            offs.append((yu, zu, gate))
        return offs

    @property
    def ctx_width(self) -> int:
        w = self.widths
        return sum(self.n_labels + w[i] + (w[i - 1] if i > 0 else 0) for i in range(self.n_layers))

    @property
    def y_path_params(self) -> int:
        w, n = self.widths, self.n_labels
        return sum(n * w[i] + (w[i - 1] * w[i] if i > 0 else 0) for i in range(self.n_layers))


def bibtex_spec():
    """multi-label-cls defaults: 1836 features, 159 labels, --layerSizes 600
    (icnn_ebundle.py:40; bibsonomy.py:19)."""
    return FCSpec(1836, 159, (600, 159))


def halfcheetah_spec():
    """RL defaults: dimO=17, dimA=6, l1size=l2size=200, lrelu=0.01, icnn_bn=False
    (RL/src/agent.py:9-10,23)."""
    return FCSpec(17, 6, (200, 200), alpha=0.01, batchnorm=False, action_box=True)


def _trunc_normal(rng, shape, std):
    v = rng.randn(*shape)
    bad = np.abs(v) > 2
    while bad.any():
        v[bad] = rng.randn(int(bad.sum()))
        bad = np.abs(v) > 2
    return (v * std).astype(np.float32)


def init_params(spec: FCSpec, seed=0, regime="init", yu_bias=0.0, gate_bias=0.0) -> Dict[str, np.ndarray]:
    """Random-init weights of the reference architecture (there is no checkpoint to load).

    regime "init":   tflearn defaults -- truncated normal std 0.02, zero biases, BN
                     gamma ~ N(1, 0.002), then |W| on the 'proj' weights (makeCvx,
                     icnn_ebundle.py:143,204).
    regime "spread": the same draw with the y-path rescaled so that pre-activations
                     are O(1) and the minimiser spreads over (0, 1) -- the regime of a
                     trained model, where bundles hold several cuts.
    RL: yu_bias = gate_bias = 1 (icnn.py:364,375 bias_init)."""
    rng = np.random.RandomState(seed)
    w, n, L = spec.widths, spec.n_labels, len(spec.szs)
    p = {}
    prev = spec.n_features
    for i in range(L):
        p["u%d/W" % i] = _trunc_normal(rng, (prev, spec.szs[i]), 0.02)
        p["u%d/b" % i] = np.zeros(spec.szs[i], np.float32)
        if i < L - 1 and spec.batchnorm:
            p["u%d/bn/gamma" % i] = (1 + 0.002 * rng.randn(spec.szs[i])).astype(np.float32)
            p["u%d/bn/beta" % i] = np.zeros(spec.szs[i], np.float32)
        prev = spec.szs[i]
    for i in range(L + 1):
        in_u = spec.n_features if i == 0 else spec.szs[i - 1]
        if i > 0:
            p["z%d_zu_u/W" % i] = _trunc_normal(rng, (in_u, w[i - 1]), 0.02)
            p["z%d_zu_u/b" % i] = np.full(w[i - 1], gate_bias, np.float32)
            p["z%d_zu_proj/W" % i] = np.abs(_trunc_normal(rng, (w[i - 1], w[i]), 0.02))
        p["z%d_yu_u/W" % i] = _trunc_normal(rng, (in_u, n), 0.02)
        p["z%d_yu_u/b" % i] = np.full(n, yu_bias, np.float32)
        p["z%d_yu/W" % i] = _trunc_normal(rng, (n, w[i]), 0.02)
        p["z%d_u/W" % i] = _trunc_normal(rng, (in_u, w[i]), 0.02)
        p["z%d_u/b" % i] = np.zeros(w[i], np.float32)
    if regime == "spread":
        for i in range(L + 1):
            in_u = spec.n_features if i == 0 else spec.szs[i - 1]
            # x-side heads: make yu / gate / zu O(1) instead of O(0.02 * sqrt(nnz))
            p["z%d_yu_u/W" % i] *= np.float32(8.0)
            p["z%d_yu_u/b" % i] += np.float32(0.5)
            p["z%d_u/W" % i] *= np.float32(4.0)
            if i > 0:
                p["z%d_zu_u/W" % i] *= np.float32(8.0)
                p["z%d_zu_u/b" % i] += np.float32(0.5)
                p["z%d_zu_proj/W" % i] *= np.float32(50.0 / np.sqrt(w[i - 1]))
            p["z%d_yu/W" % i] *= np.float32(50.0 / np.sqrt(n))
    elif regime != "init":
        raise ValueError("unknown regime %r" % regime)
    return p


def make_convex(params):
    """reference `makeCvx` (icnn_ebundle.py:143): |W| on every 'proj' weight."""
    for k in params:
        if "proj" in k and k.endswith("/W"):
            params[k] = np.abs(params[k])
    return params


def project(params):
    """reference `proj` (icnn_ebundle.py:144): clamp the 'proj' weights at 0."""
    for k in params:
        if "proj" in k and k.endswith("/W"):
            params[k] = np.maximum(params[k], 0)
    return params


def stage_weights(spec: FCSpec, params):
    """Per stage i = 0..L the column-wise concatenation icnn_be_fc_context multiplies prev_i with (include/icnn_be.h):
    [ u{i}/W (i < L) | z{i}_yu_u/W | z{i}_u/W | z{i}_zu_u/W (i > 0) ], padded with zero columns to a multiple of four,
    and the biases in the same order.  Host-side packing, once per weight update."""
    L = len(spec.szs)
    out = []
    for i in range(L + 1):
        names = (["u%d" % i] if i < L else []) + ["z%d_yu_u" % i, "z%d_u" % i] + (["z%d_zu_u" % i] if i > 0 else [])
        W = np.concatenate([params[k + "/W"] for k in names], axis=1).astype(np.float32)
        b = np.concatenate([params[k + "/b"] for k in names]).astype(np.float32)
        pad = (-W.shape[1]) % 4
        if pad:
            W = np.concatenate([W, np.zeros((W.shape[0], pad), np.float32)], axis=1)
        out.append((np.ascontiguousarray(W), b))
    return out


def context(spec: FCSpec, params, x: torch.Tensor, all_reduce=None, batch_total=None) -> torch.Tensor:
    """Host-side (torch) statement of the x-only context [B, C] float32, laid out per layer as
    yu_i | zu_i | gate_i (include/icnn_be.h); what the CPU tests and the gloo sharding tests use.  On the GPU
    `FCModel.context` runs the hand-written kernels of be_context.hip instead.  BatchNorm uses the statistics of
    the batch it is given (the reference runs with tflearn.is_training(True)), so
    when a minibatch is sharded across GPUs call this on the whole batch first."""
    dev = x.device
    t = {k: torch.as_tensor(v, device=dev) for k, v in params.items()}
    L = len(spec.szs)
    x = x.to(torch.float32)
    if all_reduce is not None and batch_total is None:
        # the global row count by the same collective (what FCModel.context_sharded does on the device path)
        cnt = torch.tensor([float(x.shape[0])], dtype=torch.float64, device=dev)
        all_reduce(cnt)
        batch_total = float(cnt.item())
    us, prev = [], x
    for i in range(L):
        u = torch.addmm(t["u%d/b" % i], prev, t["u%d/W" % i])
        if i < L - 1:
            u = torch.relu(u)
            if spec.batchnorm and all_reduce is not None:
                # data-parallel ranks: x is this rank's shard, the statistics are the global batch's -- one all-reduce of
                # (sum u, sum u^2) in float64, as icnn_be_fc_context_stage / _norm do on the device
                st = torch.stack([u.double().sum(dim=0), (u.double() ** 2).sum(dim=0)])
                all_reduce(st)
                mean = st[0] / batch_total
                var = torch.clamp(st[1] / batch_total - mean * mean, min=0.0)
                u = (u - mean.float()) * (1.0 / torch.sqrt(var.float() + 1e-5)) * t["u%d/bn/gamma" % i] + t["u%d/bn/beta" % i]
            elif spec.batchnorm:
                mean = u.mean(dim=0)
                var = ((u - mean) ** 2).mean(dim=0)
                u = (u - mean) / torch.sqrt(var + 1e-5) * t["u%d/bn/gamma" % i] + t["u%d/bn/beta" % i]
        us.append(u)
        prev = u
    parts = []
    for i in range(L + 1):
        prev = x if i == 0 else us[i - 1]
        parts.append(torch.addmm(t["z%d_yu_u/b" % i], prev, t["z%d_yu_u/W" % i]))
        parts.append(torch.addmm(t["z%d_u/b" % i], prev, t["z%d_u/W" % i]))
        if i > 0:
            parts.append(torch.relu(torch.addmm(t["z%d_zu_u/b" % i], prev, t["z%d_zu_u/W" % i])))
    ctx = torch.cat(parts, dim=1).contiguous()
    assert ctx.shape[1] == spec.ctx_width
    return ctx


class FCModel:
    solve_entry = "icnn_be_solve_fc"
    """Device-resident y-path of one FC-PICNN: the packed 'z{i}_yu/W' / 'z{i}_zu_proj/W'
    weights (MFMA B-fragment order, both orientations) plus the C descriptor the
    kernels take.  Re-create (or call `repack`) after every weight update."""

    def __init__(self, spec: FCSpec, params, device="cuda"):
        import ctypes as C

        from . import _lib
        self.spec = spec
        self.params = params
        self.device = torch.device(device)
        self._lib = _lib.load()
        m = _lib.FcModel()
        m.n = spec.n_labels
        m.n_layers = spec.n_layers
        for i, w in enumerate(spec.widths):
            m.width[i] = w
        m.alpha = float(spec.alpha)
        m.action_box = int(spec.action_box)
        m.ctx_width = spec.ctx_width
        m.wpack = None
        self.c_model = m
        n_floats = self._lib.icnn_be_fc_pack_floats(C.byref(m))
        if n_floats == 0:
            raise ValueError("model shape rejected by libicnn_be (layer count / widths / LDS budget)")
        self.n_pack_floats = int(n_floats)
        self.wpack = None
        self._ctx_keep = None
        self.c_ctx = None
        self.repack(params)                     # y-path pack + x-only stage weights

    def repack_context(self, params):
        """Upload the x-only weights (stage concatenations, BN parameters) for icnn_be_fc_context."""
        from . import _lib
        spec, dev = self.spec, self.device
        c = _lib.FcCtx()
        c.n_features, c.n, c.n_layers = spec.n_features, spec.n_labels, spec.n_layers
        for i, w in enumerate(spec.widths):
            c.width[i] = w
        c.batchnorm = int(spec.batchnorm)
        c.bn_eps = 1e-5
        keep = []
        for i, (W, b) in enumerate(stage_weights(spec, params)):
            Wd, bd = torch.from_numpy(W).to(dev), torch.from_numpy(b).to(dev)
            keep += [Wd, bd]
            c.w_stage[i], c.b_stage[i] = Wd.data_ptr(), bd.data_ptr()
            if spec.batchnorm and i < len(spec.szs) - 1:
                ga = torch.from_numpy(np.ascontiguousarray(params["u%d/bn/gamma" % i], dtype=np.float32)).to(dev)
                be = torch.from_numpy(np.ascontiguousarray(params["u%d/bn/beta" % i], dtype=np.float32)).to(dev)
                keep += [ga, be]
                c.bn_gamma[i], c.bn_beta[i] = ga.data_ptr(), be.data_ptr()
        self._ctx_keep, self.c_ctx = keep, c

    def repack(self, params):
        import ctypes as C
        L1 = self.spec.n_layers
        keep = []

        def ptr(name):
            a = np.ascontiguousarray(params[name], dtype=np.float32)
            keep.append(a)
            return a.ctypes.data

        yu = (C.c_void_p * L1)(*[ptr("z%d_yu/W" % i) for i in range(L1)])
        zu = (C.c_void_p * L1)(*([None] + [ptr("z%d_zu_proj/W" % i) for i in range(1, L1)]))
        host = np.empty(self.n_pack_floats, dtype=np.float32)
        from . import _lib
        _lib.check(self._lib.icnn_be_fc_pack(C.byref(self.c_model), yu, zu, host.ctypes.data),
                   "icnn_be_fc_pack")
        self.wpack = torch.from_numpy(host).to(self.device)
        self.c_model.wpack = self.wpack.data_ptr()
        self.params = params
        # the per-update flow of INTEGRATION.md is model.repack(params) then model.context(x) / rl_adam.adam(model, obs):
        # the x-only stage weights follow the same parameter set
        self.repack_context(params)

    def context(self, x: torch.Tensor) -> torch.Tensor:
        """x-only context rows [B, ctx_width] of the minibatch x [B, n_features] by the HIP kernels of be_context.hip
        (one MFMA GEMM per stage with routed epilogue, batch-statistics BatchNorm in place); current stream."""
        import ctypes as C

        from . import _lib
        x = x.to(self.device, torch.float32).contiguous()
        B = x.shape[0]
        assert x.shape[1] == self.spec.n_features
        ctx = torch.empty(B, self.spec.ctx_width, dtype=torch.float32, device=self.device)
        work = torch.empty(max(int(self._lib.icnn_be_fc_context_work_floats(C.byref(self.c_ctx), B)), 1),
                           dtype=torch.float32, device=self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(self._lib.icnn_be_fc_context(C.byref(self.c_ctx), x.data_ptr(), B, ctx.data_ptr(), self.spec.ctx_width,
                                                work.data_ptr(), C.c_void_p(stream)), "icnn_be_fc_context")
        return ctx

    def context_sharded(self, x_local: torch.Tensor, batch_total=None, all_reduce=None) -> torch.Tensor:
        """The same for ONE RANK'S SHARD of a data-parallel minibatch: context rows [B_local, ctx_width] of x_local with the
        u-path BatchNorm statistics of the GLOBAL batch -- the stages are issued one by one (icnn_be_fc_context_stage) and
        behind every normalised stage the ranks all-reduce 2 x width doubles (sum u, sum u^2; RCCL through
        torch.distributed by default), then icnn_be_fc_context_norm.  No rank touches rows it does not own
        (multi-label-cls/icnn_ebundle.py:339-347)."""
        import ctypes as C

        import torch.distributed as dist

        from . import _lib
        x = x_local.to(self.device, torch.float32).contiguous()
        B = x.shape[0]
        assert x.shape[1] == self.spec.n_features
        if all_reduce is None:
            if dist.is_initialized() and dist.get_world_size() > 1:
                all_reduce = dist.all_reduce
            else:
                all_reduce = lambda t: t                                       # noqa: E731 -- a world of one rank
        if batch_total is None:
            cnt = torch.tensor([float(B)], dtype=torch.float64, device=self.device)
            all_reduce(cnt)
            batch_total = float(cnt.item())
        ctx = torch.empty(B, self.spec.ctx_width, dtype=torch.float32, device=self.device)
        work = torch.empty(max(int(self._lib.icnn_be_fc_context_work_floats(C.byref(self.c_ctx), max(B, 1))), 1),
                           dtype=torch.float32, device=self.device)
        stats = torch.zeros(2 * max(self.spec.widths), dtype=torch.float64, device=self.device)
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        for stage in range(self.spec.n_layers):
            rc = self._lib.icnn_be_fc_context_stage(C.byref(self.c_ctx), stage, x.data_ptr(), B, ctx.data_ptr(),
                                                    self.spec.ctx_width, work.data_ptr(), stats.data_ptr(), stream)
            if rc < 0:
                _lib.check(rc, "icnn_be_fc_context_stage")
            if rc == 1:
                w = self.spec.widths[stage]
                if B == 0:
                    stats.zero_()
                part = stats[:2 * w]
                all_reduce(part)
                _lib.check(self._lib.icnn_be_fc_context_norm(C.byref(self.c_ctx), stage, B, C.c_double(batch_total),
                                                             part.data_ptr(), work.data_ptr(), stream),
                           "icnn_be_fc_context_norm")
        return ctx

    def clamp(self, mode="proj"):
        """The reference's clamp ops on the device-resident packed 'zu_proj' weights: mode 'makeCvx' = |W|
        (icnn_ebundle.py:143,:204), 'proj' = max(W, 0) (:144,:244-245)."""
        import ctypes as C

        from . import _lib
        code = {"makeCvx": _lib.CLAMP_ABS, "proj": _lib.CLAMP_RELU}[mode]
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(self._lib.icnn_be_fc_clamp(C.byref(self.c_model), code, C.c_void_p(stream)), "icnn_be_fc_clamp")

    def fg(self, ctx: torch.Tensor, y: torch.Tensor, finished=None):
        """E[B] float32 and dE/dy[B, n] float32 at y (float64 [B, n]) on the current stream."""
        import ctypes as C

        from . import _lib
        B = y.shape[0]
        assert y.dtype == torch.float64 and y.is_contiguous() and ctx.is_contiguous()
        assert ctx.shape == (B, self.spec.ctx_width) and ctx.dtype == torch.float32
        f = torch.empty(B, dtype=torch.float32, device=self.device)
        g = torch.empty(B, self.spec.n_labels, dtype=torch.float32, device=self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(self._lib.icnn_be_fc_fg(C.byref(self.c_model), ctx.data_ptr(), y.data_ptr(), B,
                                           f.data_ptr(), g.data_ptr(),
                                           None if finished is None else finished.data_ptr(),
                                           C.c_void_p(stream)), "icnn_be_fc_fg")
        return f, g


# --------------------------------------------------------------------------------------------- #
# Convolutional PICNN of the image-completion experiment (completion/icnn_ebundle.py:337-452)
# --------------------------------------------------------------------------------------------- #
CONV_LAYERS = ((32, 8, 4), (64, 4, 2), (64, 3, 1))    # (filters, kernel, stride), reference :344
CONV_FCS = (512, 1)                                   # reference :345


@dataclass(frozen=True)
class ConvSpec:
    """Olivetti half-face completion: x and y are H x W x 1 images (64 x 32), n = H*W = 2048."""
    H: int = 64
    W: int = 32

    @property
    def n_labels(self):
        return self.H * self.W

    @property
    def maps(self):
        """Spatial size and channels after each conv layer: [(16, 8, 32), (8, 4, 64), (8, 4, 64)]."""
        out, h, w = [], self.H, self.W
        for nf, k, s in CONV_LAYERS:
            h, w = (h + s - 1) // s, (w + s - 1) // s
            out.append((h, w, nf))
        return out

    @property
    def flat_dim(self):
        h, w, c = self.maps[-1]
        return h * w * c

    @property
    def ctx_width(self):
        m, n = self.maps, self.n_labels
        sizes = [n, m[0][0] * m[0][1] * m[0][2]]                                   # yu0, zu0
        sizes += [m[0][0] * m[0][1] * m[0][2], m[0][0] * m[0][1], m[1][0] * m[1][1] * m[1][2]]   # gate1 yu1 zu1
        sizes += [m[1][0] * m[1][1] * m[1][2], m[1][0] * m[1][1], m[2][0] * m[2][1] * m[2][2]]   # gate2 yu2 zu2
        sizes += [self.flat_dim, CONV_FCS[0], CONV_FCS[0], 1]                      # gate3 zu3 gate4 zu4
        return sum(sizes)


def _uniform_scaling(rng, shape, fan_in):
    """tflearn 'uniform_scaling' (the conv_2d / fully_connected default there): U(-a, a), a = sqrt(3/fan_in)."""
    a = np.sqrt(3.0 / fan_in)
    return rng.uniform(-a, a, size=shape).astype(np.float32)


def init_conv_params(spec: ConvSpec, seed=0, regime="init"):
    """Random-init weights of the conv PICNN; 'proj' weights |W|/2 as the reference's makeCvx does
    (completion/icnn_ebundle.py:145).  regime "spread" rescales the y-path so that the minimiser
    leaves the neighbourhood of the start point and bundles hold several cuts."""
    rng = np.random.RandomState(seed)
    p, cin = {}, 1
    for l, (nf, k, s) in enumerate(CONV_LAYERS):
        p["u%d/W" % l] = _uniform_scaling(rng, (k, k, cin, nf), k * k * cin)
        p["u%d/b" % l] = np.zeros(nf, np.float32)
        p["u%d/bn/gamma" % l] = (1 + 0.002 * rng.randn(nf)).astype(np.float32)
        p["u%d/bn/beta" % l] = np.zeros(nf, np.float32)
        if l > 0:
            p["z%d_zu_u/W" % l] = _uniform_scaling(rng, (3, 3, cin, cin), 9 * cin)
            p["z%d_zu_u/b" % l] = np.zeros(cin, np.float32)
            p["z%d_zu_proj/W" % l] = np.abs(_uniform_scaling(rng, (k, k, cin, nf), k * k * cin)) / 2
        p["z%d_yu_u/W" % l] = _uniform_scaling(rng, (3, 3, cin, 1), 9 * cin)
        p["z%d_yu_u/b" % l] = np.zeros(1, np.float32)
        p["z%d_yu/W" % l] = _uniform_scaling(rng, (k, k, 1, nf), k * k)
        p["z%d_y_red/W" % l] = _uniform_scaling(rng, (k, k, 1, 1), k * k)
        p["z%d_y_red/b" % l] = np.zeros(1, np.float32)
        p["z%d_u/W" % l] = _uniform_scaling(rng, (k, k, cin, nf), k * k * cin)
        p["z%d_u/b" % l] = np.zeros(nf, np.float32)
        cin = nf
    flat = spec.flat_dim
    p["u3/W"] = _uniform_scaling(rng, (flat, CONV_FCS[0]), flat)
    p["u3/b"] = np.zeros(CONV_FCS[0], np.float32)
    p["u3/bn/gamma"] = (1 + 0.002 * rng.randn(CONV_FCS[0])).astype(np.float32)
    p["u3/bn/beta"] = np.zeros(CONV_FCS[0], np.float32)
    p["u4/W"] = _uniform_scaling(rng, (CONV_FCS[0], 1), CONV_FCS[0])
    p["u4/b"] = np.zeros(1, np.float32)
    prev = flat
    for l, sz in zip((3, 4), CONV_FCS):
        p["z%d_zu_u/W" % l] = _uniform_scaling(rng, (prev, prev), prev)
        p["z%d_zu_u/b" % l] = np.zeros(prev, np.float32)
        p["z%d_zu_proj/W" % l] = np.abs(_uniform_scaling(rng, (prev, sz), prev)) / 2
        p["z%d_u/W" % l] = _uniform_scaling(rng, (prev, sz), prev)
        p["z%d_u/b" % l] = np.zeros(sz, np.float32)
        prev = sz
    if regime == "spread":
        for l in range(3):
            p["z%d_yu/W" % l] *= np.float32(2.0)
            p["z%d_yu_u/b" % l] += np.float32(0.5)
        for l in (1, 2, 3, 4):
            p["z%d_zu_u/b" % l] += np.float32(0.2)
            p["z%d_zu_proj/W" % l] *= np.float32(1.3)
    elif regime != "init":
        raise ValueError(regime)
    return p


def conv_context(spec: ConvSpec, params, x: torch.Tensor) -> torch.Tensor:
    """x-only context [B, ctx_width] float32 on x's device (x: [B, H, W, 1], already h-flipped by the
    caller as completion/icnn_ebundle.py:215 does).  Plain torch ops: plumbing, not the hot path."""
    import torch.nn.functional as F
    dev = x.device
    t = {k: torch.as_tensor(v, device=dev) for k, v in params.items()}
    pad_of = {8: 2, 4: 1, 3: 1}

    def conv(inp, W, b, stride):
        out = F.conv2d(inp.permute(0, 3, 1, 2), W.permute(3, 2, 0, 1), b, stride=stride, padding=pad_of[W.shape[0]])
        return out.permute(0, 2, 3, 1)

    def bn(v, g, b, dims):
        mean = v.mean(dim=dims, keepdim=True)
        var = ((v - mean) ** 2).mean(dim=dims, keepdim=True)
        return (v - mean) / torch.sqrt(var + 1e-5) * g + b

    x = x.to(torch.float32)
    us, prev = [], x
    for l, (nf, k, s) in enumerate(CONV_LAYERS):
        u = bn(torch.relu(conv(prev, t["u%d/W" % l], t["u%d/b" % l], s)), t["u%d/bn/gamma" % l],
               t["u%d/bn/beta" % l], (0, 1, 2))
        us.append(u)
        prev = u
    flat = prev.reshape(prev.shape[0], -1)
    u3 = bn(torch.relu(flat @ t["u3/W"] + t["u3/b"]), t["u3/bn/gamma"], t["u3/bn/beta"], (0,))
    us.append(u3)
    B = x.shape[0]
    parts, prevU = [], x
    for l, (nf, k, s) in enumerate(CONV_LAYERS):
        if l > 0:
            parts.append(torch.relu(conv(prevU, t["z%d_zu_u/W" % l], t["z%d_zu_u/b" % l], 1)).reshape(B, -1))
        parts.append(conv(prevU, t["z%d_yu_u/W" % l], t["z%d_yu_u/b" % l], 1).reshape(B, -1))
        parts.append(conv(prevU, t["z%d_u/W" % l], t["z%d_u/b" % l], s).reshape(B, -1))
        prevU = us[l]
    prevU = prevU.reshape(B, -1)
    for l in (3, 4):
        parts.append(torch.relu(prevU @ t["z%d_zu_u/W" % l] + t["z%d_zu_u/b" % l]))
        parts.append(prevU @ t["z%d_u/W" % l] + t["z%d_u/b" % l])
        prevU = us[3]
    # order: yu0 zu0 | gate1 yu1 zu1 | gate2 yu2 zu2 | gate3 zu3 | gate4 zu4
    ctx = torch.cat(parts, dim=1).contiguous()
    assert ctx.shape[1] == spec.ctx_width, (ctx.shape, spec.ctx_width)
    return ctx


# Operands of the conv model's context producer that read the same input through the same window, in the column order
# include/icnn_be.h (icnn_be_conv_ctx) lists: stage -> (weight names, bias names)
CONV_CTX_STAGES = (
    (("u0/W", "z0_u/W"), ("u0/b", "z0_u/b")),                                                   # x, 8x8 / 4
    (("z0_yu_u/W",), ("z0_yu_u/b",)),                                                           # x, 3x3 / 1
    (("u1/W", "z1_u/W"), ("u1/b", "z1_u/b")),                                                   # u0, 4x4 / 2
    (("z1_zu_u/W", "z1_yu_u/W"), ("z1_zu_u/b", "z1_yu_u/b")),                                   # u0, 3x3 / 1
    (("u2/W", "z2_zu_u/W", "z2_yu_u/W", "z2_u/W"), ("u2/b", "z2_zu_u/b", "z2_yu_u/b", "z2_u/b")),   # u1, 3x3 / 1
    (("u3/W", "z3_zu_u/W", "z3_u/W"), ("u3/b", "z3_zu_u/b", "z3_u/b")),                         # flat u2
    (("z4_zu_u/W", "z4_u/W"), ("z4_zu_u/b", "z4_u/b")),                                         # u3
)


def stage_conv_weights(params):
    """[(W [K][N], b [N])] per stage of CONV_CTX_STAGES: tflearn's [k][k][Cin][F] read as [K][F] (dense [in][out] as
    is), concatenated column-wise."""
    def mat(name):
        w = np.asarray(params[name], dtype=np.float32)
        return w.reshape(-1, w.shape[-1])
    return [(np.concatenate([mat(k) for k in ws], axis=1),
             np.concatenate([np.asarray(params[k], np.float32).reshape(-1) for k in bs]))
            for ws, bs in CONV_CTX_STAGES]


class ConvModel:
    """Device-resident y-path of the conv PICNN (struct icnn_be_conv_model + packed weights)."""
    solve_entry = "icnn_be_solve_conv"

    def __init__(self, spec: ConvSpec, params, device="cuda"):
        import ctypes as C

        from . import _lib
        self.spec, self.params, self.device = spec, params, torch.device(device)
        self._lib = _lib.load()
        m = _lib.ConvModel()
        m.H, m.W = spec.H, spec.W
        for l, (nf, k, s) in enumerate(CONV_LAYERS):
            m.filters[l], m.ksize[l], m.stride[l] = nf, k, s
        m.fc_hidden = CONV_FCS[0]
        m.ctx_width = spec.ctx_width
        m.wpack = None
        m.work, m.work_batch = None, 0
        self.c_model = m
        self.n_pack_floats = int(self._lib.icnn_be_conv_pack_floats(C.byref(m)))
        if self.n_pack_floats == 0:
            raise ValueError("conv model shape rejected by libicnn_be")
        self.work = None
        self.repack(params)

    def reserve(self, batch):
        """Device scratch for evaluations of up to `batch` samples (struct icnn_be_conv_model.work).  One ConvModel is used
        from ONE stream at a time (the scratch is written by every evaluation, include/icnn_be.h); growing it waits for the
        work already enqueued on the device, so an evaluation in flight on another stream never loses its buffer, and the
        old buffer stays referenced until then."""
        import ctypes as C
        if self.c_model.work_batch >= batch:
            return
        n = int(self._lib.icnn_be_conv_work_floats(C.byref(self.c_model), batch))
        if self.c_model.work_batch > 0 and self.device.type == "cuda":
            torch.cuda.synchronize(self.device)            # rare (first call per batch size): outside any timed loop
        self.work = torch.empty(max(n, 1), dtype=torch.float32, device=self.device)
        self.c_model.work, self.c_model.work_batch = self.work.data_ptr(), batch

    def repack(self, params):
        import ctypes as C

        from . import _lib
        keep = []

        def ptr(name):
            a = np.ascontiguousarray(params[name], dtype=np.float32)
            keep.append(a)
            return a.ctypes.data

        w_yu = (C.c_void_p * 3)(*[ptr("z%d_yu/W" % l) for l in range(3)])
        w_yr = (C.c_void_p * 3)(*([ptr("z%d_y_red/W" % l) for l in range(2)] + [None]))
        b_yr = (C.c_void_p * 3)(*([ptr("z%d_y_red/b" % l) for l in range(2)] + [None]))
        w_zu = (C.c_void_p * 3)(*([None] + [ptr("z%d_zu_proj/W" % l) for l in (1, 2)]))
        host = np.empty(self.n_pack_floats, dtype=np.float32)
        _lib.check(self._lib.icnn_be_conv_pack(C.byref(self.c_model), w_yu, w_yr, b_yr, w_zu, ptr("z3_zu_proj/W"),
                                               ptr("z4_zu_proj/W"), host.ctypes.data), "icnn_be_conv_pack")
        self.wpack = torch.from_numpy(host).to(self.device)
        self.c_model.wpack = self.wpack.data_ptr()
        self.params = params
        # the per-update flow of INTEGRATION.md is model.repack(params) then model.context(x) / rl_adam.adam(model, obs):
        # the x-only stage weights follow the same parameter set
        self.repack_context(params)

    def repack_context(self, params):
        """Upload the stage operands of the x-only context producer (struct icnn_be_conv_ctx, include/icnn_be.h)."""
        from . import _lib
        stages = stage_conv_weights(params)
        c = _lib.ConvCtx()
        c.bn_eps = 1e-5
        self._ctx_keep = []
        for s, (w, b) in enumerate(stages):
            ld = (w.shape[1] + 3) & ~3
            wp = np.zeros((w.shape[0], ld), np.float32)
            wp[:, :w.shape[1]] = w
            wd, bd = torch.from_numpy(wp).to(self.device), torch.from_numpy(b).to(self.device)
            self._ctx_keep += [wd, bd]
            c.w_stage[s], c.b_stage[s] = wd.data_ptr(), bd.data_ptr()
        for i in range(4):
            gd = torch.from_numpy(np.asarray(params["u%d/bn/gamma" % i], np.float32)).to(self.device)
            bd = torch.from_numpy(np.asarray(params["u%d/bn/beta" % i], np.float32)).to(self.device)
            self._ctx_keep += [gd, bd]
            c.bn_gamma[i], c.bn_beta[i] = gd.data_ptr(), bd.data_ptr()
        self.c_ctx = c

    def context(self, x: torch.Tensor) -> torch.Tensor:
        """x-only context [B, ctx_width] of x [B, H, W, 1] (already h-flipped by the caller as
        completion/icnn_ebundle.py:215 does), on the device: be_context.hip through `icnn_be_conv_context`
        (`conv_context` above is the torch restatement the tests compare it with)."""
        import ctypes as C

        from . import _lib
        x = x.to(self.device, torch.float32).contiguous()
        B = x.shape[0]
        assert tuple(x.shape[1:]) == (self.spec.H, self.spec.W, 1)
        if getattr(self, "c_ctx", None) is None:
            self.repack_context(self.params)
        ctx = torch.empty(B, self.spec.ctx_width, dtype=torch.float32, device=self.device)
        n = int(self._lib.icnn_be_conv_context_work_floats(C.byref(self.c_model), B))
        work = torch.empty(max(n, 1), dtype=torch.float32, device=self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(self._lib.icnn_be_conv_context(C.byref(self.c_model), C.byref(self.c_ctx), x.data_ptr(), B,
                                                  ctx.data_ptr(), work.data_ptr(), C.c_void_p(stream)), "icnn_be_conv_context")
        return ctx

    def clamp(self, mode="proj"):
        """makeCvx ("makeCvx": |W|/2, completion/icnn_ebundle.py:145,:190) / proj (max(W, 0), :146,:248-249) on the
        packed convex weights, in place on the device."""
        import ctypes as C

        from . import _lib
        code = {"makeCvx": _lib.CLAMP_ABS_HALF, "proj": _lib.CLAMP_RELU}[mode]
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(self._lib.icnn_be_conv_clamp(C.byref(self.c_model), code, C.c_void_p(stream)), "icnn_be_conv_clamp")

    def fg(self, ctx: torch.Tensor, y: torch.Tensor, finished=None):
        """E[B] and dE/dy[B, H*W] (float32) at y (float64, flat [B, H*W]) on the current stream."""
        import ctypes as C

        from . import _lib
        B = y.shape[0]
        assert y.dtype == torch.float64 and y.is_contiguous() and ctx.is_contiguous()
        assert y.shape[1] == self.spec.n_labels and ctx.shape == (B, self.spec.ctx_width)
        self.reserve(B)
        f = torch.empty(B, dtype=torch.float32, device=self.device)
        g = torch.empty(B, self.spec.n_labels, dtype=torch.float32, device=self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(self._lib.icnn_be_conv_fg(C.byref(self.c_model), ctx.data_ptr(), y.data_ptr(), B, f.data_ptr(),
                                             g.data_ptr(), None if finished is None else finished.data_ptr(),
                                             C.c_void_p(stream)), "icnn_be_conv_fg")
        return f, g
