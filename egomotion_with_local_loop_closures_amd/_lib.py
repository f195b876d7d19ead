"""ctypes binding of libellc_hip.so (the C ABI declared in include/ellc_abi.h).

There is no fallback: if the HIP library is missing or a GPU call fails, an exception is raised.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SO_PATH = os.path.join(CSRC, "libellc_hip.so")   # the shipping loader reads nothing from the environment; diagnostic tools call use_library()
MAX_LEVELS = 8

# every symbol include/ellc_abi.h declares (checked by tests/test_abi_symbols.py against the header text)
ABI_SYMBOLS = [
    "ellc_abi_version", "ellc_device_count", "ellc_default_config", "ellc_ctx_create", "ellc_ctx_destroy", "ellc_last_error", "ellc_sync", "ellc_stream", "ellc_ctx_counters", "ellc_ctx_set_poll_timeout_us", "ellc_ctx_set_grid_batch", "ellc_ctx_set_dense_maps", "ellc_ctx_set_persistent_schedule",
    "ellc_frame_upload", "ellc_keyframe_upload", "ellc_keyframe_from_frame", "ellc_get_image_level", "ellc_get_gradient",
    "ellc_get_max_gradient", "ellc_keyframe_set_depth", "ellc_keyframe_set_depth_level", "ellc_keyframe_get_depth_level",
    "ellc_keyframe_set_weights", "ellc_keyframe_get_weights", "ellc_keyframe_finalise_weights", "ellc_align", "ellc_align_enqueue",
    "ellc_align_fetch", "ellc_gn_iterate", "ellc_gn_display_planes", "ellc_concatenate_relative_pose", "ellc_concatenate_origin_pose", "ellc_se3_exp",
    "ellc_se3_log", "ellc_depth_set_state", "ellc_depth_get_state", "ellc_depth_set_keyframe", "ellc_depth_propagate",
    "ellc_depth_observe", "ellc_depth_fill_holes", "ellc_depth_regularize", "ellc_depth_make_inv_depth_one", "ellc_depth_regularize_fill_regularize", "ellc_depth_do_regularization",
    "ellc_depth_update_depth_image", "ellc_depth_create_keyframe", "ellc_depth_seeds", "ellc_track_frame",
    "ellc_histogram", "ellc_kl_divergence", "ellc_copy_slot", "ellc_copy_slot_across",
    "ellc_ingest_configure", "ellc_frame_ingest_bgr",
    "ellc_shard_range", "ellc_comm_unique_id", "ellc_comm_init_rccl", "ellc_comm_init_tcp", "ellc_comm_info", "ellc_comm_destroy", "ellc_comm_last_error",
    "ellc_gather_start", "ellc_gather_finish", "ellc_gather_results",
]


# what include/ellc_abi_diag.h adds (measurement hooks, device self-tests, test hooks): exported by libellc_hip_diag.so only
DIAG_SYMBOLS = [
    "ellc_profile_gn_kernel", "ellc_profile_align", "ellc_profile_depth_stage", "ellc_profile_calibrate_read", "ellc_profile_stream_read",
    "ellc_selftest_div_pair", "ellc_selftest_lu", "ellc_debug_persist_delay", "ellc_debug_set_persist_epoch", "ellc_debug_persist_counters", "ellc_debug_set_eager_lists", "ellc_debug_set_hinv_cache", "ellc_debug_set_fold_staging",
]


class EllcConfig(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("levels", C.c_int),
                ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("max_iter", C.c_int * MAX_LEVELS), ("early_exit", C.c_int),
                ("max_keyframes", C.c_int), ("max_frames", C.c_int), ("max_batch", C.c_int), ("device", C.c_int),
                ("concurrent_batches", C.c_int), ("arith", C.c_int), ("coalesce", C.c_int), ("cache_records", C.c_int), ("grid_batch", C.c_int)]


class EllcHypotheses(C.Structure):
    _fields_ = [("invDepth", C.c_void_p), ("invDepthSmoothed", C.c_void_p), ("variance", C.c_void_p), ("varianceSmoothed", C.c_void_p),
                ("validity_counter", C.c_void_p), ("blacklisted", C.c_void_p), ("isValid", C.c_void_p)]


class EllcError(RuntimeError):
    pass


def build(verbose=False):
    """Compile libellc_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    args = ["make", "-C", CSRC]
    if not verbose:
        args.insert(1, "-s")
    subprocess.check_call(args)


COMM_SO_PATH = os.path.join(CSRC, "libellc_comm.so")   # csrc/ellc_comm.cpp alone (TCP transport): loads without a GPU
COMM_SYMBOLS = ["ellc_shard_range", "ellc_comm_init_tcp", "ellc_comm_info", "ellc_comm_destroy", "ellc_comm_last_error", "ellc_gather_start", "ellc_gather_finish",
                "ellc_gather_results"]
_comm = None


def comm_lib():
    """The multi-GPU layer as a host-only library (no HIP, no RCCL): what the CPU tests of the sharded path drive."""
    global _comm
    if _comm is None:
        if not os.path.exists(COMM_SO_PATH):
            raise EllcError("libellc_comm.so is not built (%s). Run __graft_entry__.build() / make -C %s" % (COMM_SO_PATH, CSRC))
        _comm = C.CDLL(COMM_SO_PATH)
        _comm.ellc_comm_last_error.restype = C.c_char_p
        for name in COMM_SYMBOLS:
            getattr(_comm, name)
    return _comm


_lib = None
_diag = None
DIAG_SO_PATH = os.path.join(CSRC, "libellc_hip_diag.so")   # the same sources built with -DELLC_DIAG_ABI (csrc/Makefile)


def use_library(path):
    """Diagnostic tools only (tools/*.py with an A/B variant, build/libellc_hip_envdiag.so or ..._stamps.so — all built with the
    diagnostic ABI): load this build instead of the in-tree libraries, for the product entry points and the diagnostic ones
    alike. Must be called before the first lib() / diag_lib(); the path is explicit, never taken from the environment."""
    global SO_PATH, DIAG_SO_PATH, _external
    _external = True
    if _lib is not None or _diag is not None:
        raise EllcError("use_library: the library is already loaded")
    SO_PATH = DIAG_SO_PATH = os.path.abspath(path)


_external = False   # use_library(): an A/B build, possibly of an older revision that lacks the newest entry points


def _bind(path, symbols, what):
    if not os.path.exists(path):
        raise EllcError("%s is not built (%s). Run __graft_entry__.build() / make -C %s; "
                        "there is no CPU fallback for the product path." % (what, path, CSRC))
    h = C.CDLL(path)
    h.ellc_last_error.restype = C.c_char_p
    h.ellc_stream.restype = C.c_void_p
    h.ellc_kl_divergence.restype = C.c_double
    h.ellc_comm_last_error.restype = C.c_char_p
    for name in symbols:
        if _external and not hasattr(h, name):
            continue          # (diagnostic A/B builds only; the in-tree libraries must export every symbol)
        getattr(h, name)  # raises AttributeError if the library does not export it
    return h


def diag_lib():
    """libellc_hip_diag.so: everything lib() has plus include/ellc_abi_diag.h (bench.py's roofline leg, tools/, self-tests)."""
    global _diag
    if _diag is None:
        if DIAG_SO_PATH == SO_PATH:   # use_library(): one build serves both (an A/B build of an older revision may lack the newer hooks)
            _diag = lib()
        else:
            _diag = _bind(DIAG_SO_PATH, ABI_SYMBOLS + DIAG_SYMBOLS, "libellc_hip_diag.so")
    return _diag


def lib():
    global _lib
    if _lib is None:
        _lib = _bind(SO_PATH, ABI_SYMBOLS, "libellc_hip.so")
    return _lib
