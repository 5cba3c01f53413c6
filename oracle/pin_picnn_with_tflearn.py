#!/usr/bin/env python3
"""Pin the PICNN oracles against the reference's own TensorFlow graph.  TEST INFRASTRUCTURE; NOT RUN IN THIS REPOSITORY'S CI:
it needs TensorFlow r0.10 and a tflearn release of that era (README.md:33-35 of the reference), neither of which is
installable in the build image.  Written against the reference's sources as they stand; whoever has the stack runs

    python oracle/pin_picnn_with_tflearn.py /path/to/locuslab-icnn [fc_multilabel|fc_rl_leaky|fc_rl_relu]

and reports max|E - E_fixture| and max|dE/dy - dE/dy_fixture| for tests/golden/picnn__<name>.npz
(oracle/gen_picnn_fixtures.py).  Agreement at float32 level (1e-5 relative) pins oracle/picnn_oracle.py -- and through the
bit-exact chain (tests/test_gpu_parity.py: kernels == oracle/picnn_chain.c == picnn_oracle up to summation order) the HIP
kernels -- to the reference.  For the RL network, whichever of fc_rl_leaky / fc_rl_relu matches tells how that tflearn
release evaluates `leaky_relu(x, 0.01)` (RL/src/icnn.py:330, :396): configure FCSpec.alpha accordingly.

How it drives the reference (no reference code is copied; the module is imported by path):
  multi-label: `Model(nFeatures, nLabels, layerSzs, sess)` builds x_, y_, E_, dE_dy_ (multi-label-cls/icnn_ebundle.py:119-146);
      every trainable variable is assigned from the fixture by its scope name ('u0/W:0' <- 'param:u0/W', the BatchNorm
      gamma/beta as tflearn names them), `tflearn.is_training(True)` selects batch statistics (:207), then
      sess.run([E_, dE_dy_], {x_: x, y_: y}) (:218-221).
  RL: `Agent` wires replay memory and an environment, so negQ is built directly: icnn.Agent.negQ is called unbound on a stub
      with the attributes it reads (RL/src/icnn.py:325-404), the action fed is 2y-1 and the gradient is doubled (:148-158).
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_by_path(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.dont_write_bytecode = True
    spec.loader.exec_module(mod)
    return mod


def assign_all(tf, sess, params):
    """fixture parameter 'a/b' -> variable 'a/b:0'; reports anything on either side that found no partner"""
    by_name = {v.name.split(":")[0]: v for v in tf.trainable_variables()}
    missing = sorted(set(by_name) - set(params))
    extra = sorted(set(params) - set(by_name))
    for name, value in params.items():
        if name in by_name:
            sess.run(by_name[name].assign(value))
    if missing or extra:
        print("variables without a fixture entry:", missing, "\nfixture entries without a variable:", extra)
    return not missing


def main():
    ref, name = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "fc_multilabel"
    import tensorflow as tf
    import tflearn
    z = np.load(os.path.join(REPO, "tests", "golden", "picnn__%s.npz" % name))
    meta = json.loads(str(z["meta"]))
    params = {k[len("param:"):]: z[k] for k in z.files if k.startswith("param:")}
    sess = tf.Session()
    if name == "fc_multilabel":
        sys.path.insert(0, os.path.join(ref, "lib"))
        mod = load_by_path(os.path.join(ref, "multi-label-cls", "icnn_ebundle.py"), "ref_icnn_ebundle")
        model = mod.Model(meta["n_features"], meta["n_labels"], meta["layer_sizes"], sess)
        sess.run(tf.initialize_all_variables())
        ok = assign_all(tf, sess, params)
        tflearn.is_training(True)
        E, g = sess.run([model.E_, model.dE_dy_], feed_dict={model.x_: z["x"], model.y_: z["y"].astype(np.float32)})
    else:
        sys.path.insert(0, os.path.join(ref, "RL", "src"))
        mod = load_by_path(os.path.join(ref, "RL", "src", "icnn.py"), "ref_rl_icnn")
        mod.FLAGS.lrelu = meta["alpha"] if meta["alpha"] > 0 else 0.01      # the fixture pair brackets both readings
        mod.FLAGS.icnn_bn = meta["batchnorm"]
        mod.FLAGS.l1size, mod.FLAGS.l2size = meta["layer_sizes"]
        stub = types.SimpleNamespace(dimA=meta["n_labels"], dimO=(meta["n_features"],))
        obs = tf.placeholder(tf.float32, [None, meta["n_features"]], "obs")
        act = tf.placeholder(tf.float32, [None, meta["n_labels"]], "act")
        negQ = mod.Agent.negQ(stub, obs, act)
        grad = tf.gradients(negQ, act)[0]
        sess.run(tf.initialize_all_variables())
        ok = assign_all(tf, sess, params)
        tflearn.is_training(True)
        a = (2.0 * z["y"] - 1.0).astype(np.float32)                          # RL/src/icnn.py:150
        E, g = sess.run([negQ, grad], feed_dict={obs: z["x"], act: a})
        g = 2.0 * g                                                          # :152
    dE = float(np.max(np.abs(np.asarray(E).reshape(-1) - z["E"])))
    dg = float(np.max(np.abs(np.asarray(g) - z["dE_dy"])))
    print("%s: all variables assigned: %s; max|E - fixture| = %.3e (scale %.3g), max|dE/dy - fixture| = %.3e (scale %.3g)"
          % (name, ok, dE, np.abs(z["E"]).max(), dg, np.abs(z["dE_dy"]).max()))
    print("PINNED" if ok and dE <= 1e-4 * (1 + np.abs(z["E"]).max()) and dg <= 1e-4 * (1 + np.abs(z["dE_dy"]).max())
          else "NOT pinned (see the numbers above)")


if __name__ == "__main__":
    main()
