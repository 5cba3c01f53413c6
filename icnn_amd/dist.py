"""Multi-GPU use of the bundle-entropy path: one process per GPU, samples sharded.

Samples are independent inside solveBatch (lib/bundle_entropy_dual.py:147-174 has no
cross-sample state), so a minibatch is split into contiguous shards, every rank runs
the whole solve on its shard with replicated weights and NO communication, and one
gather (to a root rank, or an all-gather when every rank wants y*; RCCL over xGMI on GPUs,
gloo on CPU for the tests) assembles y*.

The only cross-sample quantity anywhere near the path is the u-path BatchNorm, which
the reference runs in batch-statistics mode: compute the x-only context on the FULL
batch (`picnn.context`) before sharding it, as `solve_sharded` does.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Join the process group described by RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"    # "nccl" is RCCL on ROCm
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    elif torch.cuda.is_available():
        torch.cuda.set_device(local)
    return rank, world, local


def shard_bounds(batch, world, rank):
    """Contiguous shard [lo, hi) of rank `rank`; the first batch % world ranks get one extra."""
    base, extra = divmod(batch, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_rows(local: torch.Tensor, batch, world, rank, dst=None):
    """Assemble row shards (they differ by at most one row) into the full [batch, ...] tensor with ONE collective.
    dst=None: all-gather, every rank gets the result.  dst=r: gather to rank r only (returns None elsewhere) -- on
    RCCL that is one grouped send/recv, every shard travels once over its own xGMI link to the root instead of
    around a ring to all ranks; what a data-parallel training step needs (nobody else consumes foreign y*)."""
    if world == 1:
        return local
    longest = shard_bounds(batch, world, 0)[1]
    equal = batch % world == 0
    if equal:
        send = local.contiguous()
    else:
        send = torch.zeros((longest,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        send[:local.shape[0]] = local
    out = None
    if dst is None or rank == dst:
        out = torch.empty((world * longest,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    if dst is None:
        dist.all_gather_into_tensor(out, send)
    else:
        dist.gather(send, gather_list=list(out.chunk(world)) if rank == dst else None, dst=dst)
        if rank != dst:
            return None
    if equal:
        return out
    pieces = []
    for r in range(world):
        lo, hi = shard_bounds(batch, world, r)
        pieces.append(out[r * longest:r * longest + (hi - lo)])
    return torch.cat(pieces, dim=0)


def solve_sharded(solve_fn, ctx_full: torch.Tensor, y0_full: torch.Tensor, world=None, rank=None, dst=None):
    """y* for the whole batch: each rank solves its contiguous shard with `solve_fn(ctx, y0) -> y`
    and the shards are gathered (see gather_rows for `dst`).  `ctx_full` must come from the full batch (BatchNorm)."""
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    if rank is None:                         # (resolved on its own: a caller may pass the world size alone)
        rank = dist.get_rank() if dist.is_initialized() else 0
    B = y0_full.shape[0]
    lo, hi = shard_bounds(B, world, rank)
    y_local = solve_fn(ctx_full[lo:hi].contiguous(), y0_full[lo:hi].contiguous())
    return gather_rows(y_local, B, world, rank, dst)


class FeedRows:
    """Rows of the implicit-differentiation feed of a (sharded) minibatch with GLOBAL sample indices: `sample[r]` is the
    minibatch index of row r, `y[r]` the point of the cut, `v[r]`, `c[r]` the placeholders of the reference's surrogate
    (multi-label-cls/icnn_ebundle.py:296-314; icnn_amd.bundle_entropy.ImplicitFeed holds the local form)."""

    def __init__(self, sample, y, v, c):
        self.sample, self.y, self.v, self.c = sample, y, v, c


def gather_feed(sample_local, y_rows, v_rows, c_rows, lo, world, rank, dst=None):
    """Assemble the per-rank feed rows (a different number on every rank: one per ACTIVE cut) on rank `dst` (None: on every
    rank).  Two collectives: an all-gather of the row counts (`world` integers) and one padded gather of the packed rows
    [R_max, 2 n + 2] float64 -- only live rows travel, not the [B, nIter, n] bundle arrays (2 x 78 MB at
    B = 4096, nIter = 30 against a few MB of live rows)."""
    n = y_rows.shape[1]
    dev = y_rows.device
    R = int(sample_local.shape[0])
    packed = torch.empty(R, 2 * n + 2, dtype=torch.float64, device=dev)
    packed[:, :n] = y_rows
    packed[:, n:2 * n] = v_rows
    packed[:, 2 * n] = c_rows
    packed[:, 2 * n + 1] = (sample_local.to(torch.int64) + lo).to(torch.float64)        # exact up to 2^53
    if world == 1:
        return FeedRows(sample_local.to(torch.int64) + lo, y_rows, v_rows, c_rows)
    counts = torch.zeros(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(counts, torch.tensor([R], dtype=torch.int64, device=dev))
    counts = [int(v) for v in counts.tolist()]
    r_max = max(max(counts), 1)
    send = torch.zeros(r_max, 2 * n + 2, dtype=torch.float64, device=dev)
    send[:R] = packed
    out = None
    if dst is None or rank == dst:
        out = torch.empty(world * r_max, 2 * n + 2, dtype=torch.float64, device=dev)
    if dst is None:
        dist.all_gather_into_tensor(out, send)
    else:
        dist.gather(send, gather_list=list(out.chunk(world)) if rank == dst else None, dst=dst)
        if rank != dst:
            return None
    rows = torch.cat([out[r * r_max:r * r_max + counts[r]] for r in range(world)], dim=0)
    return FeedRows(rows[:, 2 * n + 1].to(torch.int64), rows[:, :n].contiguous(), rows[:, n:2 * n].contiguous(),
                    rows[:, 2 * n].contiguous())


def solve_sharded_feed(solve_fn, feed_fn, ctx_local, y0_local, true_y_local, batch, world=None, rank=None, dst=None):
    """One data-parallel TRAINING step's use of the solver (multi-label-cls/icnn_ebundle.py:225-226 + :296-314): this rank
    solves ITS shard -- `res = solve_fn(ctx_local, y0_local)`, any object with `.y [B_local, n]`, `.count [B_local]`,
    `.n_iters [B_local]` --, builds its feed rows locally -- `feed_fn(res, true_y_local)` -> (sample [R] local indices,
    y [R, n], v [R, n], c [R]) -- and the ranks exchange only what the caller consumes: y* with the per-sample counts and
    nIters folded into the same rows (ONE gather) and the live feed rows (gather_feed).  `ctx_local` must carry the
    global batch's BatchNorm statistics (picnn.FCModel.context_sharded).  Returns on rank `dst` (None: everywhere) a dict
    y [batch, n], count [batch], n_iters [batch], feed (FeedRows with global sample indices); None on the other ranks."""
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    if rank is None:                         # (resolved on its own: a caller may pass the world size alone)
        rank = dist.get_rank() if dist.is_initialized() else 0
    lo, hi = shard_bounds(batch, world, rank)
    assert y0_local.shape[0] == hi - lo
    res = solve_fn(ctx_local, y0_local)
    sample, y_rows, v_rows, c_rows = feed_fn(res, true_y_local)
    n = res.y.shape[1]
    meta = torch.empty(hi - lo, n + 2, dtype=torch.float64, device=res.y.device)
    meta[:, :n] = res.y
    meta[:, n] = res.count.to(torch.float64)
    meta[:, n + 1] = res.n_iters.to(torch.float64)
    meta_all = gather_rows(meta, batch, world, rank, dst)
    feed = gather_feed(sample, y_rows, v_rows, c_rows, lo, world, rank, dst)
    if meta_all is None:
        return None
    return {"y": meta_all[:, :n].contiguous(), "count": meta_all[:, n].to(torch.int64),
            "n_iters": meta_all[:, n + 1].to(torch.int64), "feed": feed}
