"""Helpers shared by the golden-vector tests (test infrastructure)."""
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(case, variant):
    z = np.load(os.path.join(GOLDEN_DIR, "%s__%s.npz" % (case, variant)))
    return {k: z[k] for k in z.files}


def row_checksums(rows, n):
    """Same checksum as oracle/gen_golden.py: (sum, index-weighted sum) per row."""
    w = np.arange(1, n + 1, dtype=np.float64)
    out = np.zeros((len(rows), 2))
    for i, r in enumerate(rows):
        r = np.asarray(r, dtype=np.float64)
        out[i, 0] = r.sum()
        out[i, 1] = (r * w).sum()
    return out


def flatten_slots(y, G, h, ys, active, lam, n_iters, T):
    """Slot-addressed result -> the padded layout stored in the fixtures."""
    B, n = y.shape
    cnt = np.array([len(a) for a in active], dtype=np.int64)
    lam_none = np.array([l is None for l in lam], dtype=bool)
    lam_pad = np.zeros((B, T))
    b_pad = np.zeros((B, T))
    a_chk = np.zeros((B, T, 2))
    ys_chk = np.zeros((B, T, 2))
    for u in range(B):
        k = cnt[u]
        if lam[u] is not None:
            lam_pad[u, :len(lam[u])] = lam[u]
        if k:
            b_pad[u, :k] = h[u, active[u]]
            a_chk[u, :k] = row_checksums(G[u, active[u]], n)
            ys_chk[u, :k] = row_checksums(ys[u, active[u]], n)
    return dict(y=np.asarray(y), cnt=cnt, lam_none=lam_none, lam=lam_pad, b=b_pad,
                a_chk=a_chk, ys_chk=ys_chk, n_iters=np.asarray(n_iters, dtype=np.int64))


def assert_matches_golden(got, gold, y_tol, lam_tol, chk_rtol=1e-9, what=""):
    assert np.array_equal(got["n_iters"], gold["n_iters"]), what + " nIters differ"
    assert np.array_equal(got["cnt"], gold["cnt"]), what + " active-cut counts differ"
    assert np.array_equal(got["lam_none"], gold["lam_none"]), what + " lam None-ness differs"
    dy = np.max(np.abs(got["y"] - gold["y"])) if got["y"].size else 0.0
    assert dy <= y_tol, "%s max|y - y_ref| = %.3e > %.1e" % (what, dy, y_tol)
    dl = np.max(np.abs(got["lam"] - gold["lam"]))
    assert dl <= lam_tol, "%s max|lam - lam_ref| = %.3e > %.1e" % (what, dl, lam_tol)
    scale = 1.0 + np.abs(gold["b"])
    assert np.all(np.abs(got["b"] - gold["b"]) <= max(1e-9, 100 * y_tol) * scale), what + " cut offsets differ"
    for key in ("a_chk", "ys_chk"):
        scale = 1.0 + np.abs(gold[key])
        assert np.all(np.abs(got[key] - gold[key]) <= chk_rtol * scale), what + " " + key
    return dy
