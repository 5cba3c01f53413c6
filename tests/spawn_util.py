"""Two-process (gloo) test launches: torch.multiprocessing.spawn on a free local port, retried on a fresh port when the
rendezvous itself fails (the port picked by binding to 0 can be taken again before the workers bind it; a loaded machine can
miss the store's timeout).  Anything else -- a worker's own assertion or exception -- is raised at once, not re-run."""
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
        except Exception as exc:      # noqa: BLE001
            if not _is_rendezvous_failure(exc):
                raise                 # a worker's own assertion / exception: surface it now, do not re-run
            last = exc
    raise last


# signatures of the rendezvous itself, nothing broader (substrings like "store" or "timeout" also match a worker's own
# exception whose traceback merely mentions a TCPStore or a timeout argument, and re-running that hides a flaky test)
_RENDEZVOUS_MARKS = ("address already in use", "eaddrinuse", "diststoreerror", "distnetworkerror", "connection refused",
                     "connection reset by peer", "failed to connect to", "the server socket has failed to listen",
                     "timed out waiting for clients", "timed out after", "socket timeout")


def _is_rendezvous_failure(exc):
    """True for failures of the rendezvous itself (port taken again, store timeout), judged from the message: mp.spawn wraps
    whatever a worker raised in ProcessRaisedException with the worker's traceback as text."""
    if isinstance(exc, OSError):
        return True
    text = str(exc).lower()
    if "assertionerror" in text:
        return False
    return any(m in text for m in _RENDEZVOUS_MARKS)
