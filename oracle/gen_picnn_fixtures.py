#!/usr/bin/env python3
"""(params, x, y) -> (E, dE/dy) fixtures of the three PICNNs, written by the CPU oracles.  TEST INFRASTRUCTURE.

The PICNN oracles are UNPINNED: the reference evaluates its energies with TensorFlow r0.10 + tflearn (README.md:33-35;
multi-label-cls/icnn_ebundle.py:316-388, RL/src/icnn.py:325-404, completion/icnn_ebundle.py:337-452), neither of which
exists in this image or is installable without network.  These fixtures are what someone WITH that stack needs to pin
them after the fact: every parameter under the reference's own variable-scope name ('u0/W', 'z1_zu_proj/W', 'u0/bn/gamma',
...), an input batch, and the oracle's float32 outputs.  oracle/pin_picnn_with_tflearn.py feeds them to the reference's
graph-building functions (assigning the variables by name) and compares.

Reduced layer sizes, same architecture and code paths as the full models (the oracles are shape generic); BatchNorm in
batch-statistics mode (tflearn.is_training(True), the mode the reference runs inference in).  The RL network is
stored for both readings of `tflearn.activations.leaky_relu(., FLAGS.lrelu)` (RL/src/icnn.py:330, :396): alpha = 0.01 as
written, and alpha = 0 -- some tflearn releases of that era computed leaky_relu with an integer-cast alpha, i.e. plain
ReLU; whichever the pinned run matches is the one to configure (FCSpec.alpha).

    python oracle/gen_picnn_fixtures.py          (rewrites tests/golden/picnn__*.npz; deterministic)"""
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from icnn_amd import picnn  # noqa: E402  (parameter initialisers only: host-side NumPy)
from oracle import picnn_conv_oracle, picnn_oracle  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")


def fc_case(name, spec, seed, init_kw, x, note):
    params = picnn.init_params(spec, seed, "spread", **init_kw)
    B = x.shape[0]
    y = np.random.RandomState(seed + 1).uniform(0.05, 0.95, (B, spec.n_labels))
    fg = picnn_oracle.make_fg(params, x, list(spec.szs), spec.alpha, spec.batchnorm, "action" if spec.action_box else None)
    E, g = fg(y)
    meta = dict(model=name, n_features=spec.n_features, n_labels=spec.n_labels, layer_sizes=list(spec.szs), alpha=spec.alpha,
                batchnorm=bool(spec.batchnorm), action_box=bool(spec.action_box), note=note,
                y_is="the solver's variable in [0,1]^n; with action_box the network sees 2y-1 and dE/dy is doubled "
                     "(RL/src/icnn.py:148-158)")
    np.savez_compressed(os.path.join(OUT, "picnn__%s.npz" % name), x=x, y=y, E=E, dE_dy=g, meta=json.dumps(meta),
                        **{"param:" + k: v for k, v in params.items()})
    print(name, "E[:3] =", E[:3], "|dE/dy| max", np.abs(g).max())


def main():
    rng = np.random.RandomState(0)
    spec = picnn.FCSpec(40, 9, (24, 9))
    fc_case("fc_multilabel", spec, 3, {}, (rng.rand(12, 40) < 0.3).astype(np.float32),
            "multi-label-cls/icnn_ebundle.py:316-388 with nFeatures=40, nLabels=9, layerSizes=[24]; ReLU; BatchNorm with batch "
            "statistics on the hidden u layer")
    for alpha, tag in ((0.01, "leaky"), (0.0, "relu")):
        spec = picnn.FCSpec(17, 6, (20, 20), alpha=alpha, batchnorm=False, action_box=True)
        fc_case("fc_rl_%s" % tag, spec, 4, dict(yu_bias=1.0, gate_bias=1.0), rng.randn(10, 17).astype(np.float32),
                "RL/src/icnn.py:325-404 negQ with dimO=17, dimA=6, l1size=l2size=20, icnn_bn=False, lrelu=%g "
                "(the two readings of tflearn's leaky_relu, see the module docstring)" % alpha)
    cs = picnn.ConvSpec(H=16, W=8)
    params = picnn.init_conv_params(cs, 5, "spread")
    x = rng.rand(6, cs.H, cs.W, 1).astype(np.float32)
    y = np.random.RandomState(6).uniform(0.05, 0.95, (6, cs.n_labels))
    import torch
    ctx = picnn_conv_oracle.flat_context(picnn_conv_oracle.context(params, torch.from_numpy(x)))
    E, g = picnn_conv_oracle.make_fg_from_context(params, ctx, cs.H, cs.W)(y)
    meta = dict(model="conv_completion", H=cs.H, W=cs.W, note="completion/icnn_ebundle.py:337-452 on 16x8 images (the "
                "reference: 64x32), conv 32 k8 s4 / 64 k4 s2 / 64 k3 s1, fc 512, fc 1; x is the h-flipped left half (:215), "
                "y the right half flattened row-major; BatchNorm with batch statistics on u0..u3")
    # 0.7 M random floats do not belong in a fixture: the parameters are regenerated from the seed by the same host-side
    # initialiser (picnn.init_conv_params(ConvSpec(16, 8), 5, "spread"): NumPy only); a checksum per tensor guards it
    meta["params"] = "icnn_amd.picnn.init_conv_params(ConvSpec(H=16, W=8), seed=5, regime='spread')"
    meta["param_checksums"] = {k: [float(np.asarray(v, dtype=np.float64).sum()), float(np.abs(np.asarray(v, dtype=np.float64)).sum())]
                               for k, v in params.items()}
    np.savez_compressed(os.path.join(OUT, "picnn__conv_completion.npz"), x=x, y=y, E=E, dE_dy=g, meta=json.dumps(meta))
    print("conv_completion E[:3] =", E[:3], "|dE/dy| max", np.abs(g).max())


if __name__ == "__main__":
    main()
