#!/usr/bin/env python3
"""bench.py's N > 1 path with the REAL workload on a one-GPU box: two ranks, both on cuda:0, gloo as the backend (RCCL refuses
two ranks on one device).  Exercises what the CPU dry run cannot: HipWorkload's sharded context producer (all-reduce of the
BatchNorm sums on device tensors), the per-shard fused solves, the gather and the per-rank timings.  GPU box only."""
import json
import os
import socket
import sys

import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(rank, world, port, out_dir):
    sys.path.insert(0, REPO)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import bench
    args = bench.parse_args(["--gpus", str(world), "--steps", "5", "--warmup", "2", "--c4-steps", "2", "--cpu-sample", "0"])
    out = bench.run(args, backend="gloo")
    if rank == 0:
        json.dump(out, open(os.path.join(out_dir, "two_ranks.json"), "w"))


if __name__ == "__main__":
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out_dir = os.path.join(REPO, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    mp.spawn(worker, args=(2, port, out_dir), nprocs=2, join=True)
    d = json.load(open(os.path.join(out_dir, "two_ranks.json")))
    print({k: d[k] for k in ("value", "ms_per_step", "n_gpus", "per_rank_solve_ms", "per_rank_gather_ms")})
    print(d["config"])
    print({k: v for k, v in d["extra"]["c4"].items() if k in ("ms_per_step", "per_rank_solve_ms", "per_rank_gather_ms", "kernel")})
