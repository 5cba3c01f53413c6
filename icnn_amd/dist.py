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
        rank = dist.get_rank() if dist.is_initialized() else 0
    B = y0_full.shape[0]
    lo, hi = shard_bounds(B, world, rank)
    y_local = solve_fn(ctx_full[lo:hi].contiguous(), y0_full[lo:hi].contiguous())
    return gather_rows(y_local, B, world, rank, dst)
