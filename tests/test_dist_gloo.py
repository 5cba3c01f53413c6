"""The N > 1 path on CPU: world_size-2 gloo process group, contiguous batch shards, one
all-gather of y*.  The per-shard solve is injected (here: the CPU oracle) because the
product solver needs a GPU; the sharding / collective code under test is icnn_amd.dist."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

import problems

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


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
    port = _free_port()
    mp.spawn(_worker, args=(world, port, B, str(tmp_path)), nprocs=world, join=True)
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
