#!/usr/bin/env python3
"""Replay, on the CPU, the projected-Newton solve of the completion model's slowest sample (BASELINE configs[2], x seed 5,
sample 75, outer iteration 3: four cuts) with the REFERENCE's update rule (lib/bundle_entropy_dual.py:15-85, restated in
oracle/bundle_entropy_oracle.simplex_newton) and print the iterates.  The bundle comes from the GPU state
(tools/scratch-style dump: gpurun_out/c3_straggler.npz, or the copy committed under profiles/).

What it shows: the iteration does not converge and does not cycle -- the pivot alternates between cuts 2 and 3, the free set
flips, and lam wanders through (0.18..0.97, 0.03..0.82) without repeating to better than 1e-2 -- until the cap of 100 updates
(:30) ends it.  The reference returns lam_100 of that map; so does the kernel (100 updates, ~4.6 us each: the 466 us dual launch of
that round).  No shortcut reproduces lam_100 of a non-periodic map, so this launch is the floor of the lockstep round."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(REPO, "profiles", "r05_c3_straggler_bundle.npz")
d = np.load(path)
A = d["G"][:4].astype(np.float32)
b = d["h"][:4]
k = 4
c = np.sum(A, axis=1) + b
lam = np.ones(k) / k
keep = np.ones(k)
seen = []
for it in range(100):
    a = A.T.dot(lam)
    z = 1 / (1 + np.exp(-a))
    grad = -c + A.dot(z)
    hess = (A * (z * (1 - z))).dot(A.T)
    piv = int(np.argmax(lam))
    red = lam.copy()
    red[piv] = 1
    keep[piv] = 0
    col = hess[:, piv]
    g0 = grad - keep * grad[piv]
    h0 = hess - keep[:, None] * col[None, :] - col[:, None] * keep[None, :] + hess[piv, piv] * (keep[:, None] * keep[None, :])
    bound = (red <= 1e-12) & (g0 > 0)
    bound[piv] = True
    free = ~bound
    gn = np.linalg.norm(g0[free])
    if gn < 1e-10:
        print("converged after", it, "updates")
        break
    step = np.zeros(k)
    step[free] = np.linalg.solve(h0[free][:, free], -g0[free])
    t = 1.0
    for _ in range(50):
        trial = np.maximum(red + t * step, 0)
        trial[piv] = 1
        new = trial.copy()
        new[piv] = 1 - keep.dot(trial)
        if new[piv] >= 0 or t < 1e-10:
            break
        t *= 0.5
    keep[piv] = 1
    near = min([np.abs(new - s).max() for s in seen], default=np.inf)
    seen.append(new.copy())
    print("update %3d  pivot %d  free %s  t %.3g  |g0| %.3e  lam %s  nearest earlier iterate %.2e"
          % (it + 1, piv, free.astype(int), t, gn, np.array2string(new, precision=9, floatmode="fixed"), near))
    lam = new
