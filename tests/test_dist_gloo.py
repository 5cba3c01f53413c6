"""The N > 1 path on CPU: world_size-2 gloo process group, contiguous batch shards, one
all-gather of y*.  The per-shard solve is injected (here: the CPU oracle) because the
product solver needs a GPU; the sharding / collective code under test is icnn_amd.dist."""
import os
import sys

import numpy as np
import pytest
import torch

import problems
import spawn_util

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_cover_batch():
    from icnn_amd.dist import shard_bounds
    for B in (0, 1, 7, 8, 4096, 4097):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(B, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, B, out_dir):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from icnn_amd import dist as be_dist
    from icnn_amd import picnn
    from oracle import bundle_entropy_oracle as oracle
    from oracle import picnn_oracle
    r, w, _ = be_dist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)

    spec = picnn.FCSpec(40, 9, (24, 9))
    params = picnn.init_params(spec, 3, "spread")
    x = torch.from_numpy((np.random.RandomState(5).rand(B, 40) < 0.3).astype(np.float32))
    ctx_full = picnn.context(spec, params, x)          # BatchNorm statistics of the FULL batch
    y0 = torch.full((B, 9), 0.5, dtype=torch.float64)

    def solve_fn(ctx, y):
        fg = picnn_oracle.make_fg_from_context(params, ctx.numpy(), list(spec.szs))
        with np.errstate(all="ignore"):
            res = oracle.solve_batch(fg, y.numpy().copy(), 6)
        return torch.from_numpy(res.y)

    y_all = be_dist.solve_sharded(solve_fn, ctx_full, y0)
    assert y_all.shape == (B, 9)
    np.save(os.path.join(out_dir, "y_rank%d.npy" % rank), y_all.numpy())
    # the single gather to a root rank (what bench.py --gpus N times): only rank 1 receives
    y_root = be_dist.solve_sharded(solve_fn, ctx_full, y0, dst=1)
    assert (y_root is None) == (rank != 1)
    if rank == 1:
        assert torch.equal(y_root, y_all)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("B", [10, 13])
def test_two_rank_sharded_solve_equals_single_process(tmp_path, B):
    world = 2
    spawn_util.spawn(_worker, lambda port: (world, port, B, str(tmp_path)), world)
    ys = [np.load(tmp_path / ("y_rank%d.npy" % r)) for r in range(world)]
    assert np.array_equal(ys[0], ys[1]), "every rank must hold the full gathered y*"

    from icnn_amd import picnn
    from oracle import bundle_entropy_oracle as oracle
    from oracle import picnn_oracle
    spec = picnn.FCSpec(40, 9, (24, 9))
    params = picnn.init_params(spec, 3, "spread")
    x = torch.from_numpy((np.random.RandomState(5).rand(B, 40) < 0.3).astype(np.float32))
    ctx = picnn.context(spec, params, x).numpy()
    fg = picnn_oracle.make_fg_from_context(params, ctx, list(spec.szs))
    with np.errstate(all="ignore"):
        ref = oracle.solve_batch(fg, np.full((B, 9), 0.5), 6)
    assert np.array_equal(ys[0], ref.y), "sharding must not change any sample's result"


def test_context_matches_oracle_context_on_cpu():
    """Host logic: picnn.context (torch) against the oracle's NumPy context, incl. batch-stat BN."""
    from icnn_amd import picnn
    from oracle import picnn_oracle
    for spec, kw in ((picnn.FCSpec(50, 11, (32, 11)), {}),
                     (picnn.FCSpec(17, 6, (20, 20), alpha=0.01, batchnorm=False, action_box=True),
                      dict(yu_bias=1.0, gate_bias=1.0))):
        params = picnn.init_params(spec, 1, "spread", **kw)
        x = np.random.RandomState(2).randn(33, spec.n_features).astype(np.float32)
        ctx = picnn.context(spec, params, torch.from_numpy(x)).numpy()
        ref = picnn_oracle.flat_context(picnn_oracle.context(params, x, list(spec.szs), spec.batchnorm))
        assert ctx.shape == (33, spec.ctx_width)
        assert np.max(np.abs(ctx - ref)) <= 1e-5 * max(1.0, np.abs(ref).max())


def test_sharded_context_derives_the_global_row_count_itself():
    """picnn.context(all_reduce=...) without batch_total used to divide by None.  The global row count now comes
    from the same collective; two "ranks" emulated in one process (the all-reduce adds the other shard's sums) must
    reproduce the full-batch context rows of their shards, an EMPTY shard included."""
    from icnn_amd import picnn
    spec = picnn.FCSpec(50, 11, (32, 11))
    params = picnn.init_params(spec, 1, "spread")
    x = torch.from_numpy(np.random.RandomState(2).randn(21, spec.n_features).astype(np.float32))
    full = picnn.context(spec, params, x)
    for split in (8, 0):
        shards = [x[:split], x[split:]]
        # first pass: record every tensor each shard hands to all_reduce; second pass: add the partner's
        sent = [[], []]
        for r in (0, 1):
            picnn.context(spec, params, shards[r], all_reduce=lambda t, r=r: sent[r].append(t.clone()))
        outs = []
        for r in (0, 1):
            calls = iter(range(len(sent[r])))
            outs.append(picnn.context(spec, params, shards[r],
                                      all_reduce=lambda t, r=r, calls=calls: t.add_(sent[1 - r][next(calls)])))
        got = torch.cat(outs, dim=0)
        assert got.shape == full.shape
        assert torch.max(torch.abs(got - full)).item() <= 1e-5 * max(1.0, float(full.abs().max()))


def test_solve_sharded_feed_resolves_the_rank_when_only_the_world_is_given():
    """solve_sharded_feed(world=N, rank=None) passed None into shard_bounds.  Outside a process group the rank
    resolves to 0; with world = 1 the call is the single-process path."""
    from icnn_amd import dist as be_dist
    spec, params, x, true_y = _train_problem(6)
    solve_fn, feed_fn = _cpu_solve_and_feed(spec, params)
    from icnn_amd import picnn
    ctx = picnn.context(spec, params, x)
    out = be_dist.solve_sharded_feed(solve_fn, feed_fn, ctx, torch.full((6, 9), 0.5, dtype=torch.float64),
                                     torch.from_numpy(true_y), 6, world=1)
    assert out is not None and out["y"].shape == (6, 9) and out["feed"].sample.numel() == int(out["count"].sum())


# ---- a data-parallel training step: sharded context (all-reduced BatchNorm sums), local feed rows, live rows gathered ----
def _train_problem(B):
    from icnn_amd import picnn
    spec = picnn.FCSpec(40, 9, (24, 9))
    params = picnn.init_params(spec, 3, "spread")
    x = torch.from_numpy((np.random.RandomState(5).rand(B, 40) < 0.3).astype(np.float32))
    true_y = (np.random.RandomState(6).rand(B, 9) < 0.3).astype(np.float64)
    return spec, params, x, true_y


class _CpuResult:
    def __init__(self, ora):
        self.ora = ora
        self.y = torch.from_numpy(ora.y)
        self.count = torch.tensor([len(a) for a in ora.active], dtype=torch.int64)
        self.n_iters = torch.tensor(list(ora.n_iters), dtype=torch.int64)


def _cpu_solve_and_feed(spec, params):
    from oracle import bundle_entropy_oracle as oracle
    from oracle import implicit_feed_oracle as feed_oracle
    from oracle import picnn_oracle

    def solve_fn(ctx, y0):
        fg = picnn_oracle.make_fg_from_context(params, ctx.numpy(), list(spec.szs))
        with np.errstate(all="ignore"):
            return _CpuResult(oracle.solve_batch(fg, y0.numpy().copy(), 6))

    def feed_fn(res, true_y):
        _, A, _, lam, xs, _ = res.ora.as_reference_tuple()
        idx, ry, rv, rc = feed_oracle.feed_rows(res.ora.y, true_y.numpy(), A, xs, lam, "xent")
        return torch.from_numpy(idx), torch.from_numpy(ry), torch.from_numpy(rv), torch.from_numpy(rc)

    return solve_fn, feed_fn


def _train_worker(rank, world, port, B, out_dir):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from icnn_amd import dist as be_dist
    from icnn_amd import picnn
    be_dist.init_from_env(backend="gloo")
    spec, params, x, true_y = _train_problem(B)
    lo, hi = be_dist.shard_bounds(B, world, rank)
    # this rank only ever touches ITS rows of x: the BatchNorm statistics come from one all-reduce of (sum u, sum u^2)
    ctx_local = picnn.context(spec, params, x[lo:hi], all_reduce=dist.all_reduce, batch_total=float(B))
    np.save(os.path.join(out_dir, "ctx_rank%d.npy" % rank), ctx_local.numpy())
    solve_fn, feed_fn = _cpu_solve_and_feed(spec, params)
    y0 = torch.full((hi - lo, 9), 0.5, dtype=torch.float64)
    out = be_dist.solve_sharded_feed(solve_fn, feed_fn, ctx_local, y0, torch.from_numpy(true_y[lo:hi]), B, dst=0)
    assert (out is None) == (rank != 0)
    if rank == 0:
        np.savez(os.path.join(out_dir, "train.npz"), y=out["y"].numpy(), count=out["count"].numpy(),
                 n_iters=out["n_iters"].numpy(), sample=out["feed"].sample.numpy(), fy=out["feed"].y.numpy(),
                 fv=out["feed"].v.numpy(), fc=out["feed"].c.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [10, 13])
def test_two_rank_training_step_equals_single_process(tmp_path, B):
    """Sharded context with all-reduced BatchNorm sums + per-shard solve + per-shard implicit-differentiation feed + the two
    gathers of solve_sharded_feed, world size 2 over gloo, against the single-process computation on the whole batch."""
    world = 2
    spawn_util.spawn(_train_worker, lambda port: (world, port, B, str(tmp_path)), world)
    from icnn_amd import picnn
    from icnn_amd.dist import shard_bounds
    spec, params, x, true_y = _train_problem(B)
    ctx_full = picnn.context(spec, params, x).numpy()
    for r in range(world):
        lo, hi = shard_bounds(B, world, r)
        got = np.load(tmp_path / ("ctx_rank%d.npy" % r))
        assert got.shape == (hi - lo, spec.ctx_width)
        assert np.max(np.abs(got - ctx_full[lo:hi])) <= 2e-5 * max(1.0, np.abs(ctx_full).max())
    # single process, fed by the SAME (sharded-statistics) context rows so that the comparison below is exact
    ctx_rows = np.concatenate([np.load(tmp_path / ("ctx_rank%d.npy" % r)) for r in range(world)])
    solve_fn, feed_fn = _cpu_solve_and_feed(spec, params)
    res = solve_fn(torch.from_numpy(ctx_rows), torch.full((B, 9), 0.5, dtype=torch.float64))
    idx, ry, rv, rc = feed_fn(res, torch.from_numpy(true_y))
    z = np.load(tmp_path / "train.npz")
    assert np.array_equal(z["y"], res.y.numpy()) and np.array_equal(z["count"], res.count.numpy())
    assert np.array_equal(z["n_iters"], res.n_iters.numpy())
    assert np.array_equal(z["sample"], idx.numpy()), "global sample indices, in batch order"
    assert np.array_equal(z["fy"], ry.numpy()) and np.array_equal(z["fv"], rv.numpy()) and np.array_equal(z["fc"], rc.numpy())
    assert len(idx) == int(res.count.sum()) > B
