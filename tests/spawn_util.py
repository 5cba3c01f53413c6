"""Two-process (gloo) test launches: torch.multiprocessing.spawn on a free local port, retried on a fresh port when the
rendezvous itself fails (the port picked by binding to 0 can be taken again before the workers bind it; a loaded machine can
miss the store's timeout).  A failure inside the workers' own assertions fails every attempt and surfaces unchanged."""
import socket

import torch.multiprocessing as mp


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn(worker, make_args, nprocs, attempts=3):
    """mp.spawn(worker, args=make_args(port), nprocs=nprocs, join=True) with a fresh port per attempt."""
    last = None
    for _ in range(attempts):
        try:
            mp.spawn(worker, args=make_args(free_port()), nprocs=nprocs, join=True)
            return
        except Exception as exc:      # noqa: BLE001 -- ProcessRaisedException / ProcessExitedException / socket errors
            last = exc
    raise last
