"""ctypes binding of libicnn_be.so -- the C ABI declared in include/icnn_be.h.

There is no CPU fallback: if the HIP library cannot be loaded the import of the
solver fails loudly.  Build it with `python -m icnn_amd.build` (or
`__graft_entry__.build()`).
"""
import ctypes as C
import os

# torch first: its wheel carries its own HIP runtime (torch/lib/libamdhip64.so), and the process must end up with
# ONE runtime.  Loaded after torch, libicnn_be.so binds to the copy torch already mapped (same SONAME); loaded
# before, it maps /opt/rocm's copy and the second runtime to initialise finds "no ROCm-capable device".
import torch  # noqa: F401

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "csrc", "libicnn_be.so")
if os.environ.get("ICNN_BE_LIB"):          # diagnostic: another build of the same ABI (tools/lib_ab.py: same-box A/B of two builds)
    LIB_PATH = os.path.abspath(os.environ["ICNN_BE_LIB"])

ABI_VERSION = 10
MAX_LAYERS = 8
MAX_SLOTS = 31
MAX_ITERS = 64
MAX_ROUNDS = 128
VARIANT = {"dual": 0, "rl": 1, "pdipm": 2}
CUT_F32, CUT_F64 = 0, 1
ST_SINGULAR, ST_NONFINITE, ST_OVERFLOW, ST_UNFINISHED = 1, 2, 4, 8
FLAG_NO_CYCLE_SHORTCUT = 1
FLAG_TIME_SLICE = 2
FLAG_LOCKSTEP = 4
FLAG_TWO_KERNELS = 8
FLAG_PERSISTENT = 16
FLAG_F64_ENERGY = 32
FLAG_GLOBAL_BUNDLE = 64
FLAG_WAVE_PER_SAMPLE = 128
FLAG_MFMA_CONTRACTION = 256
LOSS = {"xent": 0, "mse": 1}
ERRORS = {-1: "ICNN_BE_EINVAL (bad argument)", -2: "ICNN_BE_ELIMIT (size beyond a compiled-in limit)",
          -3: "ICNN_BE_ELAUNCH (HIP launch failed)"}

EXPORTS = [
    "icnn_be_abi_version", "icnn_be_last_hip_error", "icnn_be_struct_size", "icnn_be_dual_lds_bytes", "icnn_be_bundle_capacity", "icnn_be_scratch_bytes",
    "icnn_be_state_init",
    "icnn_be_dual_step", "icnn_be_fc_pack_floats", "icnn_be_fc_pack", "icnn_be_fc_fg",
    "icnn_be_solve_fc", "icnn_be_conv_pack_floats", "icnn_be_conv_work_floats", "icnn_be_conv_pack", "icnn_be_conv_fg", "icnn_be_solve_conv",
    "icnn_be_implicit_feed", "icnn_be_export_active", "icnn_be_adam_workspace_bytes", "icnn_be_adam_fc", "icnn_be_adam_fc_obs",
    "icnn_be_fc_context_work_floats", "icnn_be_fc_context", "icnn_be_fc_context_stage", "icnn_be_fc_context_norm", "icnn_be_fc_clamp",
    "icnn_be_conv_context_work_floats", "icnn_be_conv_context", "icnn_be_conv_clamp",
    "icnn_be_debug_profile", "icnn_be_debug_profile_fc", "icnn_be_debug_profile_conv", "icnn_be_debug_profile_phases",
    "icnn_be_debug_fast_math", "icnn_be_debug_trace",
]
CLAMP_ABS, CLAMP_RELU, CLAMP_ABS_HALF = 0, 1, 2


class State(C.Structure):
    """struct icnn_be_state"""
    _fields_ = [
        ("batch", C.c_int), ("n", C.c_int), ("slots", C.c_int), ("cut_dtype", C.c_int),
        ("variant", C.c_int), ("flags", C.c_int),
        ("y", C.c_void_p), ("G", C.c_void_p), ("h", C.c_void_p), ("ys", C.c_void_p),
        ("lam", C.c_void_p), ("active", C.c_void_p), ("count", C.c_void_p),
        ("n_iters", C.c_void_p), ("finished", C.c_void_p), ("status", C.c_void_p),
        ("newton_iters", C.c_void_p),
        ("t_next", C.c_void_p), ("phase", C.c_void_p), ("skip_fg", C.c_void_p), ("pending", C.c_void_p),
        ("park", C.c_void_p), ("scratch", C.c_void_p), ("fvals", C.c_void_p), ("iters", C.c_int),
    ]


class FcModel(C.Structure):
    """struct icnn_be_fc_model"""
    _fields_ = [
        ("n", C.c_int), ("n_layers", C.c_int), ("width", C.c_int * MAX_LAYERS),
        ("alpha", C.c_float), ("action_box", C.c_int), ("ctx_width", C.c_int),
        ("wpack", C.c_void_p),
    ]


class FcCtx(C.Structure):
    """struct icnn_be_fc_ctx"""
    _fields_ = [
        ("n_features", C.c_int), ("n", C.c_int), ("n_layers", C.c_int), ("width", C.c_int * MAX_LAYERS),
        ("batchnorm", C.c_int), ("bn_eps", C.c_float),
        ("w_stage", C.c_void_p * MAX_LAYERS), ("b_stage", C.c_void_p * MAX_LAYERS),
        ("bn_gamma", C.c_void_p * MAX_LAYERS), ("bn_beta", C.c_void_p * MAX_LAYERS),
    ]


class ConvModel(C.Structure):
    """struct icnn_be_conv_model"""
    _fields_ = [
        ("H", C.c_int), ("W", C.c_int), ("filters", C.c_int * 3), ("ksize", C.c_int * 3),
        ("stride", C.c_int * 3), ("fc_hidden", C.c_int), ("ctx_width", C.c_int), ("wpack", C.c_void_p),
        ("work", C.c_void_p), ("work_batch", C.c_int),
    ]


class ConvCtx(C.Structure):
    """struct icnn_be_conv_ctx"""
    _fields_ = [
        ("w_stage", C.c_void_p * 7), ("b_stage", C.c_void_p * 7), ("bn_gamma", C.c_void_p * 4), ("bn_beta", C.c_void_p * 4),
        ("bn_eps", C.c_float),
    ]


_lib = None


def use_profiling_build():
    """Make load() take the PROFILING variant of the library (csrc/prof/libicnn_be.so: the same sources with the cycle-counter
    laps behind icnn_be_debug_profile* compiled in, `python -m icnn_amd.build --prof`), building it when missing or stale.
    The production library is built without the laps -- there the hooks set a pointer nothing reads.  Diagnostic tools call
    this before anything loads the library (tools/*_phase_profile.py)."""
    global LIB_PATH
    from . import build as _build
    if _lib is not None and LIB_PATH != _build.PROF_LIB:
        raise RuntimeError("the production library is already loaded in this process")
    LIB_PATH = _build.build(prof=True)
    return LIB_PATH


def load():
    """Load the shared library (once) and declare the prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s is missing: the HIP extension has not been built.  Run `python -m icnn_amd.build` "
            "(needs hipcc; there is no CPU fallback for the bundle-entropy kernels)." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    lib.icnn_be_abi_version.restype = C.c_int
    lib.icnn_be_last_hip_error.restype = C.c_char_p
    lib.icnn_be_dual_lds_bytes.argtypes = [C.c_int, C.c_int, C.c_int]
    lib.icnn_be_dual_lds_bytes.restype = C.c_int
    lib.icnn_be_bundle_capacity.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
    lib.icnn_be_bundle_capacity.restype = C.c_int
    lib.icnn_be_scratch_bytes.argtypes = [C.POINTER(State)]
    lib.icnn_be_scratch_bytes.restype = C.c_size_t
    lib.icnn_be_state_init.argtypes = [C.POINTER(State), C.c_void_p]
    lib.icnn_be_state_init.restype = C.c_int
    lib.icnn_be_dual_step.argtypes = [C.POINTER(State), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.icnn_be_dual_step.restype = C.c_int
    lib.icnn_be_fc_pack_floats.argtypes = [C.POINTER(FcModel)]
    lib.icnn_be_fc_pack_floats.restype = C.c_size_t
    lib.icnn_be_fc_pack.argtypes = [C.POINTER(FcModel), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                    C.c_void_p]
    lib.icnn_be_fc_pack.restype = C.c_int
    lib.icnn_be_fc_fg.argtypes = [C.POINTER(FcModel), C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p]
    lib.icnn_be_fc_fg.restype = C.c_int
    lib.icnn_be_solve_fc.argtypes = [C.POINTER(FcModel), C.c_void_p, C.POINTER(State), C.c_void_p,
                                     C.c_void_p, C.c_void_p]
    lib.icnn_be_solve_fc.restype = C.c_int
    lib.icnn_be_conv_pack_floats.argtypes = [C.POINTER(ConvModel)]
    lib.icnn_be_conv_pack_floats.restype = C.c_size_t
    lib.icnn_be_conv_work_floats.argtypes = [C.POINTER(ConvModel), C.c_int]
    lib.icnn_be_conv_work_floats.restype = C.c_size_t
    lib.icnn_be_conv_pack.argtypes = [C.POINTER(ConvModel)] + [C.POINTER(C.c_void_p)] * 4 + [C.c_void_p] * 3
    lib.icnn_be_conv_pack.restype = C.c_int
    lib.icnn_be_conv_fg.argtypes = [C.POINTER(ConvModel), C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p]
    lib.icnn_be_conv_fg.restype = C.c_int
    lib.icnn_be_solve_conv.argtypes = [C.POINTER(ConvModel), C.c_void_p, C.POINTER(State), C.c_void_p,
                                       C.c_void_p, C.c_void_p]
    lib.icnn_be_solve_conv.restype = C.c_int
    lib.icnn_be_implicit_feed.argtypes = [C.POINTER(State), C.c_void_p, C.c_int] + [C.c_void_p] * 6
    lib.icnn_be_implicit_feed.restype = C.c_int
    lib.icnn_be_export_active.argtypes = [C.POINTER(State)] + [C.c_void_p] * 6
    lib.icnn_be_export_active.restype = C.c_int
    lib.icnn_be_debug_fast_math.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    lib.icnn_be_debug_fast_math.restype = C.c_int
    lib.icnn_be_debug_profile_phases.restype = C.c_int
    lib.icnn_be_adam_workspace_bytes.argtypes = [C.c_int, C.c_int]
    lib.icnn_be_adam_workspace_bytes.restype = C.c_size_t
    lib.icnn_be_adam_fc.argtypes = [C.POINTER(FcModel), C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 5
    lib.icnn_be_adam_fc.restype = C.c_int
    lib.icnn_be_adam_fc_obs.argtypes = [C.POINTER(FcModel), C.POINTER(FcCtx), C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 5
    lib.icnn_be_adam_fc_obs.restype = C.c_int
    lib.icnn_be_fc_context_work_floats.argtypes = [C.POINTER(FcCtx), C.c_int]
    lib.icnn_be_fc_context_work_floats.restype = C.c_size_t
    lib.icnn_be_fc_context.argtypes = [C.POINTER(FcCtx), C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    lib.icnn_be_fc_context.restype = C.c_int
    lib.icnn_be_fc_context_stage.argtypes = [C.POINTER(FcCtx), C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                             C.c_void_p, C.c_void_p]
    lib.icnn_be_fc_context_stage.restype = C.c_int
    lib.icnn_be_fc_context_norm.argtypes = [C.POINTER(FcCtx), C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.icnn_be_fc_context_norm.restype = C.c_int
    lib.icnn_be_fc_clamp.argtypes = [C.POINTER(FcModel), C.c_int, C.c_void_p]
    lib.icnn_be_fc_clamp.restype = C.c_int
    lib.icnn_be_conv_context_work_floats.argtypes = [C.POINTER(ConvModel), C.c_int]
    lib.icnn_be_conv_context_work_floats.restype = C.c_size_t
    lib.icnn_be_conv_context.argtypes = [C.POINTER(ConvModel), C.POINTER(ConvCtx), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                         C.c_void_p]
    lib.icnn_be_conv_context.restype = C.c_int
    lib.icnn_be_conv_clamp.argtypes = [C.POINTER(ConvModel), C.c_int, C.c_void_p]
    lib.icnn_be_conv_clamp.restype = C.c_int
    lib.icnn_be_struct_size.argtypes = [C.c_int]
    lib.icnn_be_struct_size.restype = C.c_size_t
    if tuple(lib.icnn_be_struct_size(i) for i in range(5)) != (
            C.sizeof(State), C.sizeof(FcModel), C.sizeof(FcCtx), C.sizeof(ConvModel), C.sizeof(ConvCtx)):
        raise ImportError("ctypes struct layout differs from libicnn_be.so's")
    if lib.icnn_be_abi_version() != ABI_VERSION:
        raise ImportError("libicnn_be.so ABI %d != binding ABI %d; rebuild with python -m icnn_amd.build"
                          % (lib.icnn_be_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(rc, what):
    if rc == 0:
        return
    msg = ERRORS.get(rc, "error %d" % rc)
    if rc == -3:
        msg += ": " + load().icnn_be_last_hip_error().decode()
    raise RuntimeError("%s failed: %s" % (what, msg))
