"""Build libicnn_be.so (HIP, gfx950 only) in-tree with hipcc.

    python -m icnn_amd.build [--force] [--prof]

hipcc cross-compiles without a GPU; the resulting .so sits next to the sources
(git-ignored) so that it travels with the tree to the GPU box.

--prof builds the PROFILING variant into csrc/prof/libicnn_be.so: the same sources with -DICNN_BE_PROF=1, i.e. with the
cycle-counter laps behind icnn_be_debug_profile* compiled in.  The production library is built without them (they cost
the headline solve 1.8 %: 1.041 against 1.022 ms on one box); tools/*_phase_profile.py load the profiling variant
(icnn_amd._lib.use_profiling_build).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
SOURCES = ["be_api.hip", "be_dual.hip", "be_dual_small.hip", "be_picnn_fc.hip", "be_picnn_conv.hip", "be_fused.hip", "be_adam.hip",
           "be_context.hip"]
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith(".h")) + [os.path.join(INCLUDE, "icnn_be.h")]
LIB = os.path.join(CSRC, "libicnn_be.so")
PROF_DIR = os.path.join(CSRC, "prof")
PROF_LIB = os.path.join(PROF_DIR, "libicnn_be.so")
# Per-file compiler options.  be_fused.hip, be_adam.hip: MachineLICM hoists the literals of both inlined phases in front of the
# round loop of the persistent kernels, where they are spilled to scratch memory (be_fused.hip, FusedArgs comment).
EXTRA_FLAGS = {"be_fused.hip": ["-mllvm", "-disable-machine-licm"], "be_adam.hip": ["-mllvm", "-disable-machine-licm"]}


def _stale(lib=LIB):
    if not os.path.exists(lib):
        return True
    built = os.path.getmtime(lib)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [h if os.path.isabs(h) else os.path.join(CSRC, h)
                                                      for h in HEADERS]
    deps.append(os.path.abspath(__file__))            # the per-file compiler options live here
    return any(os.path.getmtime(d) > built for d in deps)


def build(force=False, verbose=False, out_dir=None, prof=False):
    """Compile every HIP source for gfx950 (one hipcc process per translation unit, in parallel) and link
    libicnn_be.so; returns the path of the shared library.  out_dir: build objects and library THERE instead of in-tree
    (a cold build that leaves the in-tree library alone: __graft_entry__.smoke)."""
    if out_dir is not None:
        force = True
    if prof and out_dir is None:                      # the profiling variant: its own directory, rebuilt when stale
        out_dir = PROF_DIR
        os.makedirs(out_dir, exist_ok=True)
        if not force and not _stale(PROF_LIB):
            return PROF_LIB
    elif not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wno-pass-failed",
             "-I" + INCLUDE, "-I" + CSRC] + (["-DICNN_BE_PROF=1"] if prof else []) + os.environ.get("ICNN_BE_EXTRA_FLAGS", "").split()
    header_time = max([os.path.getmtime(h if os.path.isabs(h) else os.path.join(CSRC, h)) for h in HEADERS]
                      + [os.path.getmtime(os.path.abspath(__file__))])
    jobs = []
    objs = []
    for src in SOURCES:
        path = os.path.join(CSRC, src)
        obj = os.path.join(out_dir or CSRC, src.replace(".hip", ".o"))
        objs.append(obj)
        if (not force and os.path.exists(obj) and os.path.getmtime(obj) > os.path.getmtime(path)
                and os.path.getmtime(obj) > header_time):
            continue
        cmd = [hipcc] + flags + EXTRA_FLAGS.get(src, []) + ["-c", path, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        jobs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = []
    for src, proc in jobs:
        out, _ = proc.communicate()
        if proc.returncode != 0:
            failed.append("%s:\n%s" % (src, out))
    if failed:
        raise RuntimeError("hipcc failed:\n" + "\n".join(failed))
    lib = LIB if out_dir is None else os.path.join(out_dir, os.path.basename(LIB))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", lib + ".tmp"]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc link failed:\n" + res.stdout)
    os.replace(lib + ".tmp", lib)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, prof="--prof" in sys.argv))
