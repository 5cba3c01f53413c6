"""Stand-ins for bench.HipWorkload that run on CPU: `python bench.py --workload tests/bench_stub.py:StubWorkload`
drives bench.py's control flow (rank launch, sharding, timing, gather, JSON) without a GPU."""
import time

import torch


class _StubResult:
    def __init__(self, y):
        self.y = y


class StubWorkload:
    """Stands in for HipWorkload: y*[u] = global row index of the sample, so the gathered tensor proves the sharding."""
    name = "stub"
    n = 3

    def __init__(self, args, rank, world, local):
        from icnn_amd import dist as be_dist
        self.rank, self.world = rank, world
        if args.scaling == "strong":
            self.global_batch = args.batch
            self.lo, self.hi = be_dist.shard_bounds(args.batch, world, rank)
        else:
            self.global_batch = args.batch * world
            self.lo, self.hi = rank * args.batch, (rank + 1) * args.batch
        self.local_batch = self.hi - self.lo
        self.calls = []

    def step(self, n_iter, events=None):
        if events is not None:
            events[0].t = time.perf_counter()
        time.sleep(0.002 * (1 + self.rank))            # rank 1 is slower: the max over ranks must show it
        y = torch.arange(self.lo, self.hi, dtype=torch.float64).reshape(-1, 1).repeat(1, self.n)
        self.calls.append(n_iter)
        if events is not None:
            events[1].t = time.perf_counter()
        return _StubResult(y), y

    def new_events(self):
        class Ev:
            t = 0.0

            def elapsed_time(self, other):
                return 1e3 * (other.t - self.t)
        return Ev(), Ev()

    def sync(self):
        pass


class FailsOnRankOne(StubWorkload):
    """rank 1 dies while rank 0 is waiting in a collective: the launcher must stop rank 0 and report the failure."""

    def __init__(self, args, rank, world, local):
        super().__init__(args, rank, world, local)
        if rank == 1:
            raise RuntimeError("injected failure on rank 1")
