"""Adam inner optimiser of the RL agent: oracle vs the reference's own method (golden
vectors), then the HIP kernel vs the oracle."""
import os

import numpy as np
import pytest
import torch

import problems
from golden_util import GOLDEN_DIR
from oracle import adam_oracle


@pytest.mark.parametrize("case", sorted(problems.ADAM_CASES))
def test_oracle_equals_reference_adam(case):
    """tests/golden/adam__*.npz were produced by RL/src/icnn.py's own `adam` (oracle/gen_golden_adam.py)."""
    gold = np.load(os.path.join(GOLDEN_DIR, "adam__%s.npz" % case))
    obs, n, neg_q = problems.ADAM_CASES[case]()
    best, iters, f_best = adam_oracle.adam(adam_oracle.entropy_fg(neg_q), obs, n)
    assert iters == int(gold["iters"])
    assert np.array_equal(best, gold["act_best"])
    assert np.array_equal(f_best, gold["f_at_best"])


def test_entropy_terms_match_closed_form():
    a = np.linspace(-0.999, 0.999, 41).reshape(1, -1)
    pen, grad = adam_oracle.entropy_terms(a)
    p = (a + 1) / 2
    assert np.allclose(pen, p * np.log(p) + (1 - p) * np.log(1 - p), atol=2e-6)
    assert np.allclose(grad, 0.5 * (np.log(p) - np.log(1 - p)), atol=1e-4, rtol=1e-4)
    # outside the clip range the penalty is constant: no gradient (tf.clip_by_value)
    pen, grad = adam_oracle.entropy_terms(np.array([[1.0 - 1e-8, -1.0 + 1e-8]]))
    assert (grad == 0).all()


# ---------------------------------------------------------------------------------------------------------
# HIP kernel (icnn_be_adam_fc through the C ABI) against the oracle driven by the MFMA-order PICNN restatement
# ---------------------------------------------------------------------------------------------------------
def _negq_problem(B, seed, widths=(200, 200), n_obs=17, n_act=6, scale=1.0):
    import dataclasses

    from icnn_amd import picnn
    spec = dataclasses.replace(picnn.halfcheetah_spec(), action_box=False, szs=tuple(widths), n_features=n_obs,
                               n_labels=n_act)
    params = picnn.init_params(spec, seed, "spread", yu_bias=1.0, gate_bias=1.0)
    obs = (scale * np.random.RandomState(100 + seed).randn(max(B, 1), n_obs)).astype(np.float32)
    return spec, params, obs


def _oracle_adam(spec, params, ctx_host, max_iter):
    from oracle import picnn_oracle
    chain = picnn_oracle.make_fg_chain(params, ctx_host, list(spec.szs), spec.alpha, False)
    func = adam_oracle.entropy_fg(lambda obs, act: chain(act))
    return adam_oracle.adam(func, ctx_host, spec.n_labels, max_iter)


@pytest.mark.gpu
@pytest.mark.parametrize("B,seed,max_iter", [(1, 0, 1000), (1, 11, 1000), (2, 7, 1000), (4, 8, 1000), (3, 9, 40), (16, 1, 1000), (33, 2, 1000),
                                             (100, 3, 1000), (7, 4, 9), (256, 12, 1000), (700, 13, 60), (1100, 14, 40)])
def test_adam_kernel_matches_oracle(B, seed, max_iter):
    """One launch for the whole loop: 1-4 states per workgroup on the VALU path while the workgroups fit the GPU (one
    workgroup for the agent's act() shape, a cooperative launch with a grid barrier per iteration beyond), 16-state
    MFMA tiles for larger batches (1100 here).  Both sides evaluate negQ in the same float32 order and
    the entropy term / moments with the same operations, so the iteration count must be equal and the best
    iterates agree to float64 rounding (the bar of BASELINE.json is 1e-5)."""
    import torch

    from icnn_amd import picnn, rl_adam
    spec, params, obs = _negq_problem(B, seed)
    model = picnn.FCModel(spec, params)
    ctx = model.context(torch.from_numpy(obs))[:B].contiguous()
    res = rl_adam.AdamSolver(model, B, max_iter).solve(ctx)
    torch.cuda.synchronize()
    best, iters, f_best = _oracle_adam(spec, params, ctx.cpu().numpy(), max_iter)
    got = res.act_best.cpu().numpy()
    print("adam B=%d: %d iterations (oracle %d), max |d act_best| %.3e" % (B, int(res.iters.item()), iters,
                                                                            np.max(np.abs(got - best))))
    assert int(res.iters.item()) == iters
    assert np.max(np.abs(got - best)) <= 1e-9
    assert np.max(np.abs(res.f_best.cpu().numpy() - f_best)) <= 1e-6
    assert np.all(np.abs(got) <= 1.0 - 1e-8)


@pytest.mark.gpu
def test_adam_kernel_wider_action_and_repeatable():
    """n > 64 (two column passes per wave), three hidden layers; a second call on the same buffers is identical."""
    import torch

    from icnn_amd import picnn, rl_adam
    spec, params, obs = _negq_problem(40, 5, widths=(96, 64, 48), n_obs=11, n_act=70)
    model = picnn.FCModel(spec, params)
    ctx = model.context(torch.from_numpy(obs)).contiguous()
    solver = rl_adam.AdamSolver(model, 40, 300)
    first = solver.solve(ctx)
    a1, i1 = first.act_best.cpu().numpy().copy(), int(first.iters.item())
    second = solver.solve(ctx)
    assert int(second.iters.item()) == i1 and np.array_equal(second.act_best.cpu().numpy(), a1)
    best, iters, _ = _oracle_adam(spec, params, ctx.cpu().numpy(), 300)
    assert i1 == iters
    assert np.max(np.abs(a1 - best)) <= 1e-9


@pytest.mark.gpu
def test_adam_latency_path_three_hidden_layers():
    """batch <= 4 runs adam_rows_kernel (VALU chains in MFMA order, state in LDS and registers): a deeper, narrower
    network than the agent's, uneven widths."""
    import torch

    from icnn_amd import picnn, rl_adam
    spec, params, obs = _negq_problem(2, 10, widths=(96, 50, 33), n_obs=11, n_act=20)
    model = picnn.FCModel(spec, params)
    ctx = model.context(torch.from_numpy(obs))[:2].contiguous()
    res = rl_adam.AdamSolver(model, 2, 200).solve(ctx)
    best, iters, f_best = _oracle_adam(spec, params, ctx.cpu().numpy(), 200)
    assert int(res.iters.item()) == iters
    assert np.max(np.abs(res.act_best.cpu().numpy() - best)) <= 1e-9
    assert np.max(np.abs(res.f_best.cpu().numpy() - f_best)) <= 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("B", [1, 3, 200])
def test_adam_from_observations_in_one_launch(B):
    """act(): observation -> action in ONE launch (icnn_be_adam_fc_obs): the x-only context rows are computed inside the
    Adam kernel as k-ascending fma chains.  Against the oracle's Adam fed by the MFMA-order negQ on the context rows of
    oracle/picnn_chain.c's restatement of that order: bit-identical best actions, same iteration count."""
    from icnn_amd import picnn, rl_adam
    from oracle import picnn_oracle
    import dataclasses
    spec = dataclasses.replace(picnn.halfcheetah_spec(), action_box=False)
    params = picnn.init_params(spec, 11, "spread", yu_bias=1.0, gate_bias=1.0)
    obs = np.random.RandomState(12).randn(B, spec.n_features).astype(np.float32)
    model = picnn.FCModel(spec, params)
    solver = rl_adam.AdamSolver(model, B, 300)
    res = solver.solve_obs(torch.from_numpy(obs))
    assert res is not None, "the agent's shapes are inside the latency path"
    ctx_chain = picnn_oracle.context_rows_chain(params, obs, list(spec.szs), picnn.stage_weights(spec, params))
    # (the chain context differs from the MFMA-GEMM context of model.context by float32 rounding only)
    ctx_gemm = model.context(torch.from_numpy(obs)).cpu().numpy()
    assert np.max(np.abs(ctx_chain - ctx_gemm)) <= 2e-5 * np.abs(ctx_gemm).max()
    best, iters, f_best = _oracle_adam(spec, params, ctx_chain, 300)
    assert int(res.iters.item()) == iters
    assert np.array_equal(res.act_best.cpu().numpy(), best)
    # a BatchNorm model is outside the one-launch path: the wrapper falls back to context + Adam
    bn_spec = dataclasses.replace(picnn.bibtex_spec(), action_box=False)
    bn_model = picnn.FCModel(picnn.FCSpec(20, 5, (16, 8)), picnn.init_params(picnn.FCSpec(20, 5, (16, 8)), 1, "spread"))
    assert rl_adam.AdamSolver(bn_model, 2, 50).solve_obs(torch.randn(2, 20)) is None
    assert rl_adam.adam(bn_model, torch.randn(2, 20), max_iter=50, one_launch=True).shape == (2, 5)
    del bn_spec


@pytest.mark.gpu
def test_adam_entry_point_rejects_bad_arguments():
    import ctypes as C
    import dataclasses

    import torch

    from icnn_amd import _lib, picnn, rl_adam
    spec, params, obs = _negq_problem(4, 6)
    boxed = picnn.FCModel(dataclasses.replace(spec, action_box=True), params)
    with pytest.raises(ValueError):
        rl_adam.AdamSolver(boxed, 4)
    model = picnn.FCModel(spec, params)
    solver = rl_adam.AdamSolver(model, 4)
    ctx = model.context(torch.from_numpy(obs)).contiguous()
    lib = _lib.load()
    rc = lib.icnn_be_adam_fc(C.byref(model.c_model), ctx.data_ptr(), 4, 0, solver.act_best.data_ptr(),
                             solver.f_best.data_ptr(), solver.iters.data_ptr(), solver.workspace.data_ptr(), None)
    assert rc == -1
    # more states than a cooperative launch can hold resident: ELIMIT, nothing enqueued
    big = 16 * 256 * 2 + 1
    assert lib.icnn_be_adam_workspace_bytes(big, spec.n_labels) > 0
    huge = rl_adam.AdamSolver(model, big)
    ctx_big = ctx[:1].expand(big, -1).contiguous()
    with pytest.raises(RuntimeError, match="ELIMIT"):
        huge.solve(ctx_big)
