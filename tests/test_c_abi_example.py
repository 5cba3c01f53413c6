"""examples/c_abi_solve.cpp: the C ABI used from plain C++ (no Python, no torch in the process) gives the same bits as
the Python host mirror on BASELINE.json configs[0] (convex quadratic over [0,1]^4, batch 32, nIter 10, float64 cuts)."""
import os
import shutil
import subprocess

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _problem(seed=0, B=32, n=4):
    rng = np.random.RandomState(seed)
    M = rng.randn(n, n)
    Q = M.T.dot(M) + 0.1 * np.eye(n)
    Q = 0.5 * (Q + Q.T)
    return Q, rng.randn(B, n), np.full((B, n), 0.5)


def _fg_like_the_example(Q, P):
    """f and g with the example's operation order (plain loops, no fused multiply-add)."""
    n = Q.shape[0]

    def fg(y):
        B = y.shape[0]
        f, g = np.zeros(B), np.zeros((B, n))
        for u in range(B):
            fu = 0.0
            for i in range(n):
                qy = 0.0
                for j in range(n):
                    qy += Q[i, j] * y[u, j]
                g[u, i] = qy + P[u, i]
                fu += y[u, i] * (0.5 * qy + P[u, i])
            f[u] = fu
        return f, g
    return fg


def test_example_source_uses_only_the_public_header():
    src = open(os.path.join(REPO, "examples", "c_abi_solve.cpp")).read()
    assert '#include "icnn_be.h"' in src and "torch" not in src.replace("no torch", "") and "Python.h" not in src


@pytest.mark.gpu
def test_c_abi_example_matches_the_python_host(tmp_path):
    from icnn_amd import bundle_entropy
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "c_abi_solve")
    lib_dir = os.path.join(REPO, "icnn_amd", "csrc")
    subprocess.run([hipcc, "-O2", "-ffp-contract=off", "-I", os.path.join(REPO, "include"),
                    os.path.join(REPO, "examples", "c_abi_solve.cpp"), "-o", exe, "-L", lib_dir, "-licnn_be",
                    "-Wl,-rpath," + lib_dir], check=True, capture_output=True, timeout=300)
    Q, P, y0 = _problem()
    B, n = P.shape
    T = 10
    prob, out = str(tmp_path / "problem.bin"), str(tmp_path / "result.bin")
    with open(prob, "wb") as fh:
        np.array([B, n, T], np.int32).tofile(fh)
        for a in (Q, P, y0):
            np.ascontiguousarray(a, np.float64).tofile(fh)
    run = subprocess.run([exe, prob, out], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stderr
    raw = np.fromfile(out, dtype=np.uint8)
    y = raw[:B * n * 8].view(np.float64).reshape(B, n)
    ints = raw[B * n * 8:].view(np.int32).reshape(3, B)
    res = bundle_entropy.solveBatch(_fg_like_the_example(Q, P), y0.copy(), nIter=T, native=True)
    assert np.array_equal(y, res.y.cpu().numpy())
    assert np.array_equal(ints[0], res.count[:B].cpu().numpy())
    assert np.array_equal(ints[1], res.n_iters[:B].cpu().numpy())
    assert not ints[2].any()
    assert "sum(y*)" in run.stdout
