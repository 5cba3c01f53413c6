"""Host mirror of the reference's `bundle_entropy` module on top of libicnn_be.so.

    solveBatch(fg, initXs, nIter=10, callback=None)          the reference signature
        lib/bundle_entropy_dual.py:129, lib/bundle_entropy.py:192, RL/src/bundle_entropy.py:85
    solveBatch(f=model, x=features, y0=..., nIter=...)        fused PICNN form (north star)

Both forms run the per-sample work (cut bookkeeping, rank test, projected-Newton
dual solve, pruning) on the GPU.  In the first form `fg` is the caller's opaque
Python callable, evaluated once per bundle iteration exactly like the reference
does; in the second form the PICNN energy and its y-gradient are evaluated by
the HIP kernels too and the whole loop is enqueued without a host round trip.

Return value: the reference's 6-tuple `(x, A, b, lam, xs, nIters)` -- `x` IS the
`initXs` array that was passed in, updated in place -- unless `native=True`,
in which case a `BundleResult` holding the device tensors is returned.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib

__all__ = ["solveBatch", "solve", "BundleResult", "BundleState", "FusedSolver", "implicit_feed"]


# False: never allocate the device-memory staging area of wide-row solves (struct icnn_be_state.scratch); samples whose
# bundle outgrows the LDS then stop with ICNN_BE_ST_OVERFLOW, as for a C caller that passes NULL (tests)
ALLOW_SCRATCH = True


class BundleState:
    """Device buffers of one solve (struct icnn_be_state)."""

    def __init__(self, y: torch.Tensor, n_iter: int, variant: str, cut_dtype=torch.float32, flags=0, slots=None):
        if variant not in _lib.VARIANT:
            raise ValueError("variant must be 'dual', 'rl' or 'pdipm', got %r" % (variant,))
        if not (1 <= n_iter <= _lib.MAX_ITERS):
            raise ValueError("nIter must be in 1..%d, got %d" % (_lib.MAX_ITERS, n_iter))
        # up to MAX_SLOTS iterations the cut of iteration t lives in slot t; beyond, the slots of pruned cuts are recycled
        # (struct icnn_be_state.iters): the reference has no cap on nIter (lib/bundle_entropy_dual.py:129), only the ACTIVE
        # bundle is limited to MAX_SLOTS cuts here
        # `slots` (optional) caps the active bundle lower still (less memory: the slot arrays are [B, slots, n]); a sample that
        # needs more gets ICNN_BE_ST_OVERFLOW and stops at its current iterate
        if slots is not None and not (1 <= slots <= min(n_iter, _lib.MAX_SLOTS)):
            raise ValueError("slots must be in 1..min(nIter, %d), got %d" % (_lib.MAX_SLOTS, slots))
        slots = min(n_iter, _lib.MAX_SLOTS) if slots is None else slots
        assert y.dtype == torch.float64 and y.dim() == 2 and y.is_contiguous() and y.is_cuda
        dev = y.device
        B, n = y.shape
        self.B, self.n, self.T, self.variant, self.n_iter = B, n, slots, variant, n_iter
        self.y = y
        self.G = torch.zeros(B, slots, n, dtype=cut_dtype, device=dev)
        self.h = torch.zeros(B, slots, dtype=torch.float64, device=dev)
        self.ys = torch.zeros(B, slots, n, dtype=torch.float64, device=dev)
        self.lam = torch.zeros(B, slots, dtype=torch.float64, device=dev)
        self.active = torch.zeros(B, slots, dtype=torch.int32, device=dev)
        ints = torch.zeros(8, max(B, 1), dtype=torch.int32, device=dev)
        (self.count, self.n_iters, self.finished, self.status, self.newton_iters,
         self.t_next, self.phase, self.skip_fg) = ints
        self.pending = torch.zeros(_lib.MAX_ROUNDS, dtype=torch.int32, device=dev)
        self.park = torch.zeros(max(B, 1), 5 * slots + 4, dtype=torch.float64, device=dev)
        self.fvals = torch.zeros(max(B, 1), slots, dtype=torch.float64, device=dev)
        s = _lib.State()
        s.batch, s.n, s.slots = B, n, slots
        s.iters = n_iter if n_iter > slots else 0
        s.fvals = self.fvals.data_ptr()
        s.cut_dtype = _lib.CUT_F64 if cut_dtype == torch.float64 else _lib.CUT_F32
        s.variant = _lib.VARIANT[variant]
        s.flags = flags
        for name in ("y", "G", "h", "ys", "lam", "active", "count", "n_iters", "finished", "status",
                     "newton_iters", "t_next", "phase", "skip_fg", "pending", "park"):
            setattr(s, name, getattr(self, name).data_ptr())
        self.c_state = s
        self.lib = _lib.load()
        # wide rows (n = 2048): staging area in device memory for the rounds whose bundle exceeds the LDS capacity
        need = int(self.lib.icnn_be_scratch_bytes(C.byref(s))) if ALLOW_SCRATCH else 0
        self.scratch = torch.empty(need, dtype=torch.uint8, device=dev) if need else None
        s.scratch = self.scratch.data_ptr() if need else None

    def stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.y.device).cuda_stream)

    def init(self):
        _lib.check(self.lib.icnn_be_state_init(C.byref(self.c_state), self.stream()), "icnn_be_state_init")

    def step(self, t, f: torch.Tensor, g: torch.Tensor):
        f_dtype = torch.float64 if (self.c_state.flags & _lib.FLAG_F64_ENERGY) else self.G.dtype
        assert f.is_contiguous() and g.is_contiguous() and g.dtype == self.G.dtype and f.dtype == f_dtype
        assert f.shape == (self.B,) and g.shape == (self.B, self.n)
        _lib.check(self.lib.icnn_be_dual_step(C.byref(self.c_state), t, f.data_ptr(), g.data_ptr(),
                                              self.stream()), "icnn_be_dual_step")


class BundleResult:
    """Slot-addressed result on the device (see include/icnn_be.h): cut taken at outer
    iteration t lives in slot t; `active[u, :count[u]]` are the slots still in sample u's
    bundle, `lam[u, :count[u]]` their multipliers."""

    def __init__(self, state: BundleState, host_y=None):
        self.state = state
        self.y, self.G, self.h, self.ys = state.y, state.G, state.h, state.ys
        self.lam, self.active, self.count = state.lam, state.active, state.count
        self.n_iters, self.finished, self.status = state.n_iters, state.finished, state.status
        self.newton_iters = state.newton_iters
        self._host_y = host_y

    def raise_on_error(self):
        """Map per-sample status to the reference's exceptions (SURVEY.md 8(b) 'Errors')."""
        status = self.status[:self.state.B].cpu().numpy()
        if self.state.variant in ("dual", "pdipm") and (status & _lib.ST_SINGULAR).any():
            # lib/bundle_entropy_dual.py:54-63 re-raises numpy's LinAlgError; lib/bundle_entropy.py:42 lets
            # numpy.linalg.cholesky's propagate (completion/icnn_ebundle.py:229-237 catches it and skips the batch)
            raise np.linalg.LinAlgError("Singular matrix (sample %d)" % int(np.nonzero(status & 1)[0][0]))
        if (status & _lib.ST_NONFINITE).any():
            raise FloatingPointError("non-finite value in the bundle of sample %d"
                                     % int(np.nonzero(status & _lib.ST_NONFINITE)[0][0]))
        if (status & _lib.ST_UNFINISHED).any():
            raise RuntimeError("sample %d is still behind after the finishing rounds of a time-sliced solve "
                               "(ICNN_BE_ST_UNFINISHED)" % int(np.nonzero(status & _lib.ST_UNFINISHED)[0][0]))
        if (status & _lib.ST_OVERFLOW).any():
            # no counterpart in the reference (its bundle is a Python list): wide rows (n = 2048) leave LDS room for
            # 13 active cuts, include/icnn_be.h icnn_be_bundle_capacity
            cs = self.state.c_state
            raise MemoryError("the active bundle of sample %d outgrew the %d cuts one workgroup can stage (n = %d)"
                              % (int(np.nonzero(status & _lib.ST_OVERFLOW)[0][0]),
                                 _lib.load().icnn_be_bundle_capacity(cs.n, cs.slots, cs.cut_dtype, cs.variant), cs.n))

    def as_reference_tuple(self):
        """(x, A, b, lam, xs, nIters) with the reference's Python types (dual :179): NumPy y, per-sample lists of the active
        cuts' gradients / offsets / points, an array of multipliers (or None) per sample.

        What an unmodified icnn_ebundle.py call site pays on top of the solve (hard part 6), so it is
        built for ONE device round trip: the counts come over first (they size everything), icnn_be_export_active packs the
        active rows (sum of count rows, not the [B, T, n] slot arrays) behind y in one device buffer, one copy brings that
        buffer into pinned host memory (torch's caching host allocator: a block that the previous call's result no longer
        references is reused, not pinned again).  The four ragged containers are `_RaggedList`s -- `list` subclasses over
        views of that block whose per-sample entries (real lists of row views / NumPy scalars, arrays of multipliers) are
        built in one bulk pass when the container is first used: O(B K) Python objects cost more than the copy (~1.3 ms
        per container against 0.8 ms for the copy at batch 4096)."""
        st = self.state
        B, T, n = st.B, st.T, st.n
        dev = self.y.device
        cnt_dev = self.count[:B]
        offs_dev = (torch.cumsum(cnt_dev, 0, dtype=torch.int32) - cnt_dev).contiguous()
        head = torch.stack([cnt_dev, self.n_iters[:B]]).cpu().numpy()       # the one wait before the copy is sized
        cnt, n_iters = head[0], head[1].tolist()
        R = int(cnt.sum())
        g_item = self.G.element_size()
        sizes = [B * n * 8, R * n * 8, R * 8, R * 8, R * n * g_item]          # y | ys | h | lam | G  (float64 first: alignment)
        starts = np.concatenate([[0], np.cumsum([(v + 15) & ~15 for v in sizes])]).tolist()
        dbuf = torch.empty(starts[-1], dtype=torch.uint8, device=dev)

        def sec(buf, i, dtype, shape):
            return buf[starts[i]:starts[i] + sizes[i]].view(dtype).view(shape)

        sec(dbuf, 0, torch.float64, (B, n)).copy_(self.y)
        if R:
            _lib.check(st.lib.icnn_be_export_active(C.byref(st.c_state), offs_dev.data_ptr(), dbuf[starts[4]:].data_ptr(),
                                                    dbuf[starts[1]:].data_ptr(), dbuf[starts[2]:].data_ptr(),
                                                    dbuf[starts[3]:].data_ptr(), st.stream()), "icnn_be_export_active")
        hbuf = torch.empty(starts[-1], dtype=torch.uint8, pin_memory=True)
        hbuf.copy_(dbuf, non_blocking=True)
        torch.cuda.current_stream(dev).synchronize()
        y = sec(hbuf, 0, torch.float64, (B, n)).numpy()
        if self._host_y is not None:
            self._host_y[...] = y
            y = self._host_y
        ys_act = sec(hbuf, 1, torch.float64, (R, n)).numpy()
        h_act = sec(hbuf, 2, torch.float64, (R,)).numpy()
        lam_act = sec(hbuf, 3, torch.float64, (R,)).numpy()
        G_act = sec(hbuf, 4, self.G.dtype, (R, n)).numpy()
        offs = np.concatenate([[0], np.cumsum(cnt)]).tolist()

        def rows_of(arr):           # list(arr): one C-level pass that makes the row views (scalars for h); then B list slices
            return _RaggedList(B, lambda: (lambda rows: [rows[offs[u]:offs[u + 1]] for u in range(B)])(list(arr)))

        # dual :134/:155-161: a sample whose very first cut is the zero vector never gets multipliers
        lams = _RaggedList(B, lambda: [None if (offs[u + 1] == offs[u] and n_iters[u] < 0) else lam_act[offs[u]:offs[u + 1]]
                                       for u in range(B)])
        return y, rows_of(G_act), rows_of(h_act), lams, rows_of(ys_act), n_iters


class _BuiltList(list):
    """What a `_RaggedList` turns into once its entries exist: a `list` subclass with no overrides, i.e. indexing and
    iteration at the C speed of `list` (the reference's inner loops index `ys[j][i]`, `lam[j][i]` per cut)."""
    __slots__ = ("_make_all",)


class _RaggedList(list):
    """One of the per-sample lists `A, b, xs, lam` of the reference's return value (lib/bundle_entropy_dual.py:131-134,
    :179) over one packed host array.  A real `list` of B entries whose entries -- O(B K) Python objects, more expensive than
    the device-to-host copy they describe -- are built in ONE bulk pass the first time the list is used in any way (`len`
    excepted: the length is known): indexing, iteration, comparison, slicing, pickling, `numpy.array`, mutation.  From then
    on the object is a plain list subclass without overrides (`_BuiltList`).  A caller that never looks at a container
    (RL/src/icnn.py:155 uses `[0]` of the tuple only) never pays for it.  Until the first use the storage holds `None`s
    that only code reaching into the list's storage from C without going through the sequence protocol could see:
    `materialize()` first there."""
    __slots__ = ("_make_all",)

    def __init__(self, size, make_all):
        list.__init__(self, [None] * size)
        self._make_all = make_all

    def materialize(self):
        if type(self) is _RaggedList:
            built = self._make_all()
            self._make_all = None
            self.__class__ = _BuiltList
            self[:] = built
        return self

    def __array__(self, *args, **kwargs):                  # numpy looks here before it walks the storage
        return np.array(list(self.materialize()), *args, **kwargs)

    def __reduce_ex__(self, protocol):                     # pickles, copy.copy and copy.deepcopy give a plain list
        return list, (list(self.materialize()),)

    def __reduce__(self):
        return self.__reduce_ex__(2)


def _materialized(name):
    def method(self, *args, **kwargs):
        return getattr(self.materialize(), name)(*args, **kwargs)       # (the built object's own, C-level, method)

    method.__name__ = name
    return method


for _name in ("__getitem__", "__iter__", "__reversed__", "__contains__", "__eq__", "__ne__", "__lt__", "__le__", "__gt__", "__ge__",
              "__repr__", "__add__", "__iadd__", "__mul__", "__rmul__", "__imul__", "__setitem__", "__delitem__", "append", "extend",
              "insert", "pop", "remove", "clear", "index", "count", "copy", "sort", "reverse"):
    setattr(_RaggedList, _name, _materialized(_name))
_RaggedList.__hash__ = None


def _pick_device(device):
    if device is not None:
        return torch.device(device)
    if not torch.cuda.is_available():
        raise RuntimeError("icnn_amd.bundle_entropy needs a GPU: there is no CPU fallback "
                           "(the CPU restatement under oracle/ is test infrastructure)")
    return torch.device("cuda", torch.cuda.current_device())


_SOLVERS = ("pc", "boyd")     # lib/bundle_entropy.py:192, :224-232


def solveBatch(fg=None, initXs=None, nIter=None, callback=None, solver=None, *, f=None, x=None, y0=None, ctx=None,
               variant="dual", native=False, device=None, fg_on_device=False, flags=0, check=True):
    """Batched argmin_y f(y) - H(y) over [0,1]^n by the bundle entropy method.

    Reference form      solveBatch(fg, initXs, nIter=10, callback=None)
        fg(y[B,n] float64 ndarray) -> (f[B], g[B,n]) ndarrays (float32 or float64);
        callback(t, f, y) (variant 'dual') / callback(t, f) (variant 'rl') is invoked
        before the update of iteration t.  `initXs` is updated in place and returned.
        With fg_on_device=True, fg and callback receive/return torch tensors on the GPU.
    Fused form          solveBatch(f=FCModel, x=features | ctx=context, y0=..., nIter=...)
        the PICNN energy is evaluated by the HIP kernels; `fg` must be None.

    variant: 'dual' = lib/bundle_entropy_dual.py (default nIter 10), 'rl' =
    RL/src/bundle_entropy.py (default nIter 5).

    variant 'pdipm' = lib/bundle_entropy.py with solver='pc' (default nIter 10): the per-sample subproblem solved by
    Mehrotra's predictor-corrector interior-point method -- the module multi-label-cls/icnn_ebundle.py and
    completion/icnn_ebundle.py import; dropin/bundle_entropy.py selects it.

    solver: the fifth positional argument of lib/bundle_entropy.py:192.  'pc' selects variant 'pdipm'; 'boyd'
    (pdipm_boyd, :80-156, never used by the reference's scripts) is not built and raises NotImplementedError; any
    other name raises the reference's RuntimeError (:232).
    """
    if solver is not None:
        if solver not in _SOLVERS:
            raise RuntimeError("Solver unknown: %s." % solver)          # lib/bundle_entropy.py:232
        if solver == "boyd":
            raise NotImplementedError("solver='boyd' (pdipm_boyd, lib/bundle_entropy.py:80-156) is not built: the reference's "
                                      "scripts use the default solver='pc', and pdipm_boyd's step-size loop (:147-148) is "
                                      "decided by rounding noise (DESIGN.md section 7)")
        variant = "pdipm"
    if nIter is None:
        nIter = 5 if variant == "rl" else 10
    dev = _pick_device(device)
    fused = f is not None
    if fused:
        if fg is not None:
            raise TypeError("pass either fg (generic mode) or f= (fused PICNN mode), not both")
        if initXs is None:
            initXs = y0
    if initXs is None:
        raise TypeError("initXs / y0 is required")

    host_y = None
    if isinstance(initXs, np.ndarray):
        host_y = initXs
        y = torch.from_numpy(np.ascontiguousarray(initXs, dtype=np.float64)).to(dev)
    else:
        y = initXs if (initXs.is_cuda and initXs.dtype == torch.float64 and initXs.is_contiguous()) \
            else initXs.to(dev, torch.float64).contiguous()
    B, n = y.shape
    if B == 0:                                              # nothing to solve: the reference's loops are empty
        if native:
            state = BundleState(y, nIter, variant, torch.float32, flags)
            return BundleResult(state, host_y)
        return initXs, [], [], [], [], []

    if fused:
        if callback is not None and nIter > _lib.MAX_SLOTS:
            raise TypeError("callback in fused mode needs nIter <= %d (beyond it the slots are recycled and hold no "
                            "per-iteration history); use the generic fg form" % _lib.MAX_SLOTS)
        if ctx is None:
            ctx = f.context(torch.as_tensor(x))
        state = BundleState(y, nIter, variant, torch.float32, flags)
        state.init()
        if hasattr(f, "reserve"):
            f.reserve(B)                                   # scratch that crosses the launches of one evaluation
        f_work = torch.empty(B, dtype=torch.float32, device=dev)
        g_work = torch.empty(B, n, dtype=torch.float32, device=dev)
        rounds = getattr(state.lib, f.solve_entry)(C.byref(f.c_model), ctx.data_ptr(), C.byref(state.c_state),
                                                   f_work.data_ptr(), g_work.data_ptr(), state.stream())
        if rounds < 0:
            _lib.check(rounds, f.solve_entry)
        state.rounds = rounds
        state._keep = (ctx, f_work, g_work)
        if callback is not None:
            _replay_callbacks(state, callback, variant, lambda yy: f.fg(ctx, yy)[0])
    else:
        state = None
        for t in range(nIter):
            if fg_on_device:
                f_t, g_t = fg(y)
            else:
                if host_y is None:                         # initXs was a tensor: fg still gets a NumPy view of y
                    host_y = y.cpu().numpy().copy()
                elif t > 0:
                    host_y[...] = y.cpu().numpy()          # the reference mutates initXs every iteration
                f_t, g_t = fg(host_y)
            g_t = torch.as_tensor(g_t)
            f_t = torch.as_tensor(f_t)
            cut_dtype = torch.float64 if g_t.dtype == torch.float64 else torch.float32
            if state is None:
                # float64 energies with float32 gradients keep their precision in b = f - <g, y> (dual :143)
                if cut_dtype == torch.float32 and f_t.dtype == torch.float64:
                    flags |= _lib.FLAG_F64_ENERGY
                state = BundleState(y, nIter, variant, cut_dtype, flags)
                state.init()
            g_t = g_t.to(dev, cut_dtype).contiguous()
            f_dtype = torch.float64 if (flags & _lib.FLAG_F64_ENERGY) else cut_dtype
            f_t = f_t.to(dev, f_dtype).contiguous().reshape(B)
            if callback is not None:
                f_cb = f_t if fg_on_device else f_t.cpu().numpy()
                y_cb = y if fg_on_device else host_y
                if variant == "rl":
                    callback(t, f_cb)                      # RL/src/bundle_entropy.py:104
                else:
                    callback(t, f_cb, y_cb)                # lib/bundle_entropy_dual.py:145
            state.step(t, f_t, g_t)
            if t + 1 < nIter and int(state.finished[:B].min().item()) == 1:
                break                                      # dual :176-177
        if state is None:                                   # nIter == 0 cannot happen (checked above)
            raise ValueError("nIter must be >= 1")

    res = BundleResult(state, host_y)
    if check:
        res.raise_on_error()
    if native:
        if host_y is not None:
            host_y[...] = y.cpu().numpy()
        return res
    return res.as_reference_tuple()


def solve(fg, initX, nIter=10, callback=None, variant="dual", device=None):
    """Single-sample form of the reference, `solve(fg, initX, nIter=10, callback=None)` (lib/bundle_entropy_dual.py:87-127):
    `fg(x[n]) -> (f scalar, g[n])`, `callback(t, f, x)`; returns the minimiser as a NEW array (the reference rebinds `x`, it
    does not write into `initX`).  Runs as a batch of one through `solveBatch`, i.e. on the device like everything else.
    One deliberate difference: the reference's `solve` has no rank test (:155-161 exist in `solveBatch` only) -- on a
    rank-deficient bundle it hands `proj_newton_logistic` a singular system and raises or returns noise; here the sample
    stops at its current iterate as in `solveBatch`.  With float32 gradients the reference's `solve` also keeps the first
    update as a float32 array for one iteration (:119), which moves its result by ~1e-7 from the batch algorithm's; this
    function is the batch algorithm (within BASELINE.json's 1e-5 of the reference's `solve`, tests/golden/solve__dual.npz).
    No script of the reference calls `solve`; the one in
    lib/bundle_entropy.py (:168-190) refers to undefined names (`pdipm`, `G`, `h`) and raises NameError there."""
    x0 = np.array(initX, dtype=np.float64, copy=True).reshape(1, -1)

    def fg_batch(Y):
        f, g = fg(np.array(Y[0], copy=True))
        g = np.asarray(g)
        return np.asarray(f).reshape(1).astype(np.float64 if g.dtype == np.float64 else g.dtype, copy=False), g.reshape(1, -1)

    cb = None if callback is None else (lambda t, f, Y: callback(t, f[0], Y[0]))
    y = solveBatch(fg_batch, x0, nIter, cb, variant=variant, device=device)[0]
    return np.array(y[0], copy=True)


def _replay_callbacks(state, callback, variant, f_at):
    """callback(t, f, y) (variant 'rl': callback(t, f)) of a FUSED solve, replayed after the launch from the state: the
    reference invokes it at the start of every outer iteration with the whole batch's energies and iterates
    (lib/bundle_entropy_dual.py:144-145, RL/src/bundle_entropy.py:103-104), but here the iterations run on the device without
    a host round trip.  Slot t holds what iteration t saw -- the point (ys) and its energy (fvals) --, and a sample that left
    the loop keeps its iterate, so every argument is reconstructed exactly: same values, same number of calls (the loop ends
    after the iteration in which the last sample finished, dual :176-177), only later.  `f_at(y)` evaluates the energy at
    the final iterates for samples that stopped by the RL stall rule (rl :125-126: their iterate moved once more after the
    last evaluation).  ebundle-vs-gd.py-style per-iteration objectives therefore work on the fused path."""
    B, T = state.B, state.T
    fv = state.fvals[:B].cpu().numpy()
    ys = state.ys.cpu().numpy()
    y_fin = state.y.cpu().numpy()
    t_next = state.t_next[:B].cpu().numpy()
    fin = state.finished[:B].cpu().numpy().astype(bool)
    f_dtype = np.float64 if (state.G.dtype == torch.float64 or (state.c_state.flags & _lib.FLAG_F64_ENERGY)) else np.float32
    # a sample that finished WITHOUT moving (rank test, error) holds the cut of its last iteration in slot t_next
    rows = np.arange(B)
    at_cut = fin & (t_next < T) & np.all(ys[rows, np.minimum(t_next, T - 1)] == y_fin, axis=1)
    filled = np.where(at_cut, t_next + 1, t_next)                     # slots 0 .. filled-1 hold evaluations of this sample
    left_at = np.where(fin, filled - 1, T - 1)                        # iteration in which the sample left the loop
    t_last = T - 1 if not fin.all() else int(left_at.max())
    f_final = None
    x_cb = np.empty_like(y_fin)
    for t in range(t_last + 1):
        have = t < filled
        slot = np.minimum(t, np.maximum(filled - 1, 0))
        f_t = fv[rows, slot].copy()
        x_cb[...] = np.where(have[:, None], ys[rows, slot], y_fin)
        moved_on = ~have & ~at_cut                                    # stall rule: the iterate moved after its last evaluation
        if moved_on.any():
            if f_final is None:
                f_final = f_at(state.y).double().cpu().numpy()
            f_t[moved_on] = f_final[moved_on]
        f_t = f_t.astype(f_dtype)
        if variant == "rl":
            callback(t, f_t)
        else:
            callback(t, f_t, x_cb)


class FusedSolver:
    """Reusable buffers for repeated fused solves of one PICNN at a fixed batch shape
    (what a training loop or the benchmark calls once per minibatch).  Everything --
    state reset, nIter x (energy+gradient kernel, dual-step kernel) -- is enqueued on the
    current stream with no host synchronisation; `solve` returns a BundleResult whose
    tensors are overwritten by the next call."""

    def __init__(self, model, batch, n_iter=10, variant="dual", device=None, flags=0, slots=None):
        dev = _pick_device(device if device is not None else model.device)
        n = model.spec.n_labels
        self.model, self.batch, self.n_iter = model, batch, n_iter
        if hasattr(model, "reserve"):
            model.reserve(batch)
        self.y = torch.empty(batch, n, dtype=torch.float64, device=dev)
        self.state = BundleState(self.y, n_iter, variant, torch.float32, flags, slots)
        self.f_work = torch.empty(max(batch, 1), dtype=torch.float32, device=dev)
        self.g_work = torch.empty(max(batch, 1), n, dtype=torch.float32, device=dev)

    def solve(self, ctx: torch.Tensor, y0=0.5):
        assert ctx.is_contiguous() and ctx.dtype == torch.float32
        assert ctx.shape == (self.batch, self.model.spec.ctx_width)
        if torch.is_tensor(y0):
            self.y.copy_(y0)
        else:
            self.y.fill_(float(y0))
        st = self.state
        st.init()
        rounds = getattr(st.lib, self.model.solve_entry)(C.byref(self.model.c_model), ctx.data_ptr(),
                                                         C.byref(st.c_state), self.f_work.data_ptr(),
                                                         self.g_work.data_ptr(), st.stream())
        if rounds < 0:
            _lib.check(rounds, self.model.solve_entry)
        st.rounds = rounds
        return BundleResult(st)


class ImplicitFeed:
    """Rows of the training feed built from a solve (device tensors): `sample[r]` is the minibatch index
    of row r (gather x with it), `y[r]` the point of the cut, `v[r]` and `c[r]` the placeholders `v_`, `c_`
    of the reference's surrogate F = c E + <dE/dy, v> (multi-label-cls/icnn_ebundle.py:148)."""

    def __init__(self, sample, y, v, c):
        self.sample, self.y, self.v, self.c = sample, y, v, c

    def as_tuple(self):
        """(sample, y, v, c): the form icnn_amd.dist.solve_sharded_feed's feed_fn returns"""
        return self.sample, self.y, self.v, self.c


def implicit_feed(res: BundleResult, true_y, loss="xent"):
    """GPU replacement of `train_step_fd` + `crossEntrGrad` / `mseGrad`
    (multi-label-cls/icnn_ebundle.py:296-314, 390-417; completion/icnn_ebundle.py:315-335, 493-522)."""
    st = res.state
    dev = st.y.device
    B, n = st.B, st.n
    t = torch.as_tensor(true_y).to(dev, torch.float64).contiguous()
    assert t.shape == (B, n)
    cnt = st.count[:B].to(torch.int64)
    offs = (torch.cumsum(cnt, 0) - cnt).to(torch.int32).contiguous()
    R = int(cnt.sum().item())
    fd_y = torch.empty(R, n, dtype=torch.float64, device=dev)
    fd_v = torch.empty(R, n, dtype=torch.float64, device=dev)
    fd_c = torch.empty(R, dtype=torch.float64, device=dev)
    fd_s = torch.empty(R, dtype=torch.int32, device=dev)
    if R:
        _lib.check(st.lib.icnn_be_implicit_feed(C.byref(st.c_state), t.data_ptr(), _lib.LOSS[loss], offs.data_ptr(),
                                                fd_y.data_ptr(), fd_v.data_ptr(), fd_c.data_ptr(), fd_s.data_ptr(),
                                                st.stream()), "icnn_be_implicit_feed")
    return ImplicitFeed(fd_s, fd_y, fd_v, fd_c)
