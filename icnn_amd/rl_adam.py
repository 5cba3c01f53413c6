"""Host mirror of the RL agent's default inner optimiser, `Agent.adam(func, obs)` (RL/src/icnn.py:160-215) with
func = `_fg_entr` (negQ minus the entropy of the action, :59-63): what `act()` runs on one observation per
environment step (:264-288) and `train()` on a minibatch of next observations (:306-317).

The reference evaluates the TensorFlow graph once per Adam iteration (up to 1000 sess.run calls per action);
here the whole loop is ONE launch of `adam_fc_kernel` (icnn_amd/csrc/be_adam.hip) through the C ABI
`icnn_be_adam_fc`.  There is no CPU fallback.
"""
import ctypes as C

import torch

from . import _lib


class AdamResult:
    """Device tensors of one call: act_best [B, n] float64 in [-1, 1] (what the reference returns), f_best [B]
    float32 = negQ_entr at act_best, iters (0-d int32) = evaluations before the stopping rule fired."""

    def __init__(self, act_best, f_best, iters):
        self.act_best, self.f_best, self.iters = act_best, f_best, iters


class AdamSolver:
    """Reusable buffers for repeated calls at one batch shape.  `model`: picnn.FCModel of negQ with
    spec.action_box False (Adam works on the action itself, not on the [0,1] re-parametrisation the bundle
    method uses)."""

    def __init__(self, model, batch, max_iter=1000):
        if model.spec.action_box:
            raise ValueError("adam() takes the action as is: build the model with action_box=False")
        self.model, self.batch, self.max_iter = model, int(batch), int(max_iter)
        self.lib = _lib.load()
        dev = model.device
        n = model.spec.n_labels
        self.act_best = torch.empty(self.batch, n, dtype=torch.float64, device=dev)
        self.f_best = torch.empty(max(self.batch, 1), dtype=torch.float32, device=dev)
        self.iters = torch.zeros((), dtype=torch.int32, device=dev)
        self.workspace = torch.empty(self.lib.icnn_be_adam_workspace_bytes(self.batch, n), dtype=torch.uint8, device=dev)

    def solve(self, ctx: torch.Tensor) -> AdamResult:
        assert ctx.is_cuda and ctx.is_contiguous() and ctx.dtype == torch.float32
        assert ctx.shape == (self.batch, self.model.spec.ctx_width)
        stream = torch.cuda.current_stream(ctx.device).cuda_stream
        _lib.check(self.lib.icnn_be_adam_fc(C.byref(self.model.c_model), ctx.data_ptr(), self.batch, self.max_iter,
                                            self.act_best.data_ptr(), self.f_best.data_ptr(), self.iters.data_ptr(),
                                            self.workspace.data_ptr(), stream), "icnn_be_adam_fc")
        self._keep = ctx
        return AdamResult(self.act_best, self.f_best[:self.batch], self.iters)

    def solve_obs(self, obs: torch.Tensor):
        """Observation -> action in ONE launch (`icnn_be_adam_fc_obs`): the x-only context rows are computed inside the
        Adam kernel.  Latency path only (at most four states per workgroup, no BatchNorm); returns None when the shape is
        outside it -- the caller then uses `solve(model.context(obs))`."""
        obs = obs.to(self.model.device, torch.float32).contiguous()
        assert obs.shape == (self.batch, self.model.spec.n_features)
        stream = torch.cuda.current_stream(obs.device).cuda_stream
        rc = self.lib.icnn_be_adam_fc_obs(C.byref(self.model.c_model), C.byref(self.model.c_ctx), obs.data_ptr(), self.batch,
                                          self.max_iter, self.act_best.data_ptr(), self.f_best.data_ptr(),
                                          self.iters.data_ptr(), self.workspace.data_ptr(), stream)
        if rc == -2:                                       # ICNN_BE_ELIMIT: not the latency path
            return None
        _lib.check(rc, "icnn_be_adam_fc_obs")
        self._keep = obs
        return AdamResult(self.act_best, self.f_best[:self.batch], self.iters)


def adam(model, obs=None, ctx=None, max_iter=1000, verbose=False, one_launch=False):
    """`Agent.adam(func, obs)`: returns act_best [B, dimA] (device float64).  `obs` [B, dimO] goes through the
    model's x-only context producer, or pass a precomputed `ctx`.  one_launch=True: observation -> action in a single
    kernel launch where the shape allows it (AdamSolver.solve_obs); measured on MI355X it is not faster than the default
    (240 us against 229 us per act() at B = 1: the three small context launches overlap with launch overhead, the
    in-kernel context is a serial prologue of one workgroup), so it is opt-in."""
    res = None
    if ctx is None:
        obs = torch.as_tensor(obs)
        solver = AdamSolver(model, obs.shape[0], max_iter)
        if one_launch:
            res = solver.solve_obs(obs)
        if res is None:
            res = solver.solve(model.context(obs).contiguous())
    else:
        res = AdamSolver(model, ctx.shape[0], max_iter).solve(ctx.contiguous())
    if verbose:
        it = int(res.iters.item())
        print("  + Adam took {} iterations".format(it) if it < max_iter else "  + Warning: Adam did not converge.")
    return res.act_best
