"""Helpers for the GPU parity tests (test infrastructure)."""
import numpy as np

from golden_util import flatten_slots


def result_to_host(res):
    """BundleResult (device) -> dict of NumPy arrays + ragged lists like the oracle's BundleResult."""
    B = res.state.B
    cnt = res.count[:B].cpu().numpy()
    act = res.active.cpu().numpy()
    lam = res.lam.cpu().numpy()
    n_iters = res.n_iters[:B].cpu().numpy()
    active = [list(act[u, :cnt[u]]) for u in range(B)]
    lams = [None if (cnt[u] == 0 and n_iters[u] < 0) else lam[u, :cnt[u]].copy() for u in range(B)]
    return dict(y=res.y.cpu().numpy(), G=res.G.cpu().numpy(), h=res.h.cpu().numpy(),
                ys=res.ys.cpu().numpy(), active=active, lam=lams, n_iters=list(n_iters),
                finished=res.finished[:B].cpu().numpy(), status=res.status[:B].cpu().numpy(),
                newton=res.newton_iters[:B].cpu().numpy())


def flatten_result(res, T):
    h = result_to_host(res)
    return flatten_slots(h["y"], h["G"], h["h"], h["ys"], h["active"], h["lam"], h["n_iters"], T), h


def compare_with_oracle(host, ora, what="", same_slots=True):
    """Per-sample comparison of a GPU result (result_to_host) with an oracle BundleResult.
    Returns (max |dy| per sample, samples whose discrete outcome differs).  same_slots=False: compare the SIZE of the
    active set only (more iterations than slots: the device recycles slots, the oracle's slot is the iteration number)."""
    dy = np.max(np.abs(host["y"] - ora.y), axis=1)
    B = len(dy)
    discrete = []
    for u in range(B):
        same_act = (list(host["active"][u]) == list(ora.active[u])) if same_slots else (len(host["active"][u]) == len(ora.active[u]))
        same = same_act and int(host["n_iters"][u]) == int(ora.n_iters[u])
        if not same:
            discrete.append(u)
    return dy, discrete
