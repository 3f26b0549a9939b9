"""ctypes binding of include/klt_gpu.h (libkltgpu.so).

There is no CPU fallback: if the HIP library is missing or cannot be loaded this
module raises -- the product path must fail loudly rather than silently compute
somewhere else.
"""
import ctypes as C
import os

KLT_MAX_LEVELS = 8
_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libkltgpu.so")


class KltFeat(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("val", C.c_int32), ("aux", C.c_int32)]


class KltParams(C.Structure):
    _fields_ = [
        ("mindist", C.c_int32), ("window_width", C.c_int32), ("window_height", C.c_int32),
        ("smoothBeforeSelecting", C.c_int32), ("retainTrackers", C.c_int32), ("nSkippedPixels", C.c_int32),
        ("max_iterations", C.c_int32), ("nPyramidLevels", C.c_int32), ("subsampling", C.c_int32),
        ("use_max_residue", C.c_int32),
        ("min_determinant", C.c_float), ("min_displacement", C.c_float), ("step_factor", C.c_float),
        ("max_residue", C.c_float),
        ("min_eigenvalue", C.c_double),
        ("grad_sigma", C.c_double), ("smooth_sigma", C.c_double), ("pyramid_sigma", C.c_double),
        ("borderx", C.c_double), ("bordery", C.c_double),
    ]


class KltAffineParams(C.Structure):
    _fields_ = [("mode", C.c_int32), ("window_width", C.c_int32), ("window_height", C.c_int32), ("max_iterations", C.c_int32),
                ("max_residue", C.c_float), ("min_displacement", C.c_float), ("max_displacement_differ", C.c_float)]


class KltAffineRec(C.Structure):
    _fields_ = [("aff_x", C.c_float), ("aff_y", C.c_float), ("Axx", C.c_float), ("Ayx", C.c_float), ("Axy", C.c_float),
                ("Ayy", C.c_float), ("valid", C.c_int32), ("pad", C.c_int32)]


class KltTrackStats(C.Structure):
    _fields_ = [("features", C.c_uint64), ("level_visits", C.c_uint64 * KLT_MAX_LEVELS),
                ("iterations", C.c_uint64 * KLT_MAX_LEVELS)]


class KltKernelTime(C.Structure):
    _fields_ = [("name", C.c_char * 32), ("launches", C.c_uint32), ("total_ms", C.c_float), ("bytes", C.c_double)]


# every symbol include/klt_gpu.h declares: name -> (restype, argtypes)
_P = C.c_void_p
_I = C.c_int
_PI = C.POINTER(C.c_int)
SYMBOLS = {
    "klt_abi_version": (_I, []),
    "klt_device_count": (_I, []),
    "klt_create": (_I, [_I, C.POINTER(_P)]),
    "klt_destroy": (None, [_P]),
    "klt_last_error": (C.c_char_p, [_P]),
    "klt_sync": (_I, [_P]),
    "klt_stream_handle": (_P, [_P]),
    "klt_set_params": (_I, [_P, C.POINTER(KltParams)]),
    "klt_set_kernels": (_I, [_P, _I, C.POINTER(C.c_double), _I, C.POINTER(C.c_double), _I]),
    "klt_upload_u8": (_I, [_P, _I, _P, _I, _I, _I]),
    "klt_upload_f32": (_I, [_P, _I, _P, _I, _I, _I]),
    "klt_host_alloc": (_I, [_P, C.c_size_t, C.POINTER(_P)]),
    "klt_host_free": (_I, [_P, _P]),
    "klt_upload_u8_async": (_I, [_P, _I, _P, _I, _I, _I]),
    "klt_upload_f32_async": (_I, [_P, _I, _P, _I, _I, _I]),
    "klt_upload_wait": (_I, [_P]),
    "klt_host_compare": (_I, [_P, _P, C.c_size_t]),
    "klt_host_copy": (_I, [_P, _P, C.c_size_t]),
    "klt_host_lanes": (_I, []),
    "klt_host_thread_serial": (_I, [_I]),
    "klt_host_compare_rows": (_I, [_P, _I, C.c_size_t, _P]),
    "klt_host_copy_rows": (_I, [_P, _P, _I, C.c_size_t]),
    "klt_host_sample_rows": (_I, [_P, _I, _I, _I, _I, _P, C.c_size_t]),
    "klt_host_luma_rows": (_I, [_P, _P, _I, _I]),
    "klt_slot_adopt_u8": (_I, [_P, _I, _P, _I, _I, _I]),
    "klt_device_alloc": (_I, [_P, C.c_size_t, C.POINTER(_P)]),
    "klt_device_write": (_I, [_P, _P, _P, C.c_size_t]),
    "klt_device_free": (_I, [_P, _P]),
    "klt_build_pyramids_async": (_I, [_P, _I]),
    "klt_build_pyramids_batch_async": (_I, [_P, C.POINTER(C.c_int), _I]),
    "klt_set_option": (_I, [_P, _I, _I]),
    "klt_build_pyramids": (_I, [_P, _I]),
    "klt_slot_state": (_I, [_P, _I]),
    "klt_device_memory": (_I, [_P, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "klt_slot_generation": (_I, [_P, _I, C.POINTER(C.c_uint64)]),
    "klt_slot_free": (_I, [_P, _I]),
    "klt_swap_slots": (_I, [_P, _I, _I]),
    "klt_featbuf_upload": (_I, [_P, _I, _P, _I]),
    "klt_featbuf_upload_async": (_I, [_P, _I, _P, _I]),
    "klt_featbuf_download": (_I, [_P, _I, _P, _I]),
    "klt_featbuf_download_async": (_I, [_P, _I, _P, _I]),
    "klt_download_wait": (_I, [_P]),
    "klt_download_mark_async": (_I, [_P]),
    "klt_featbuf_map_host": (_I, [_P, _I, _P, _I]),
    "klt_featbuf_alloc": (_I, [_P, _I, _I]),
    "klt_featbuf_view": (_I, [_P, _I, _I, _I, _I]),
    "klt_featbuf_devptr": (_P, [_P, _I]),
    "klt_select_async": (_I, [_P, _I, _I, _I, _I, _I]),
    "klt_select_begin_async": (_I, [_P, _I, _I, _I, _I, _I]),
    "klt_select_finish": (_I, [_P]),
    "klt_select_prepare_async": (_I, [_P, _I]),
    "klt_select": (_I, [_P, _I, _I, _I, _P, _I, _PI]),
    "klt_min_distance_walk": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _I, _PI]),
    "klt_track_async": (_I, [_P, _I, _I, _I, _I, _I]),
    "klt_track": (_I, [_P, _I, _I, _P, _I, _PI]),
    "klt_track_batch_async": (_I, [_P, _PI, _PI, _PI, _PI, _I, _I]),
    "klt_set_affine_params": (_I, [_P, C.POINTER(KltAffineParams)]),
    "klt_affine_alloc": (_I, [_P, _I, _I]),
    "klt_affine_download": (_I, [_P, _I, _P, _I]),
    "klt_affine_free": (_I, [_P, _I]),
    "klt_affine_copy_async": (_I, [_P, _I, _I, _I, _I]),
    "klt_track_affine_async": (_I, [_P, _I, _I, _I, _I, _I, _I]),
    "klt_track_affine": (_I, [_P, _I, _I, _P, _I, _I, _PI]),
    "klt_track_stats_reset": (_I, [_P]),
    "klt_track_stats_read": (_I, [_P, C.POINTER(KltTrackStats)]),
    "klt_level_dims": (_I, [_P, _I, _I, _PI, _PI]),
    "klt_download_f32": (_I, [_P, _I, _I, _I, _P]),
    "klt_select_dims": (_I, [_P, _I, _PI, _PI]),
    "klt_download_select_f32": (_I, [_P, _I, _P]),
    "klt_set_score_override": (_I, [_P, _P, _I]),
    "klt_download_sorted_candidates": (_I, [_P, _P, _P, _P, _I, _PI]),
    "klt_smooth_f32": (_I, [_P, _P, _I, _I, C.POINTER(C.c_double), _I, _P]),
    "klt_convolve_separate_f32": (_I, [_P, _P, _I, _I, C.POINTER(C.c_double), _I, C.POINTER(C.c_double), _I, _P]),
    "klt_pyramid_f32": (_I, [_P, _P, _I, _I, _I, _I, C.POINTER(C.c_double), _I, _P]),
    "klt_gradients_f32": (_I, [_P, _P, _I, _I, C.POINTER(C.c_double), _I, C.POINTER(C.c_double), _I, _P, _P]),
    "klt_scan_good_features_f32": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _I, _PI, _PI]),
    "klt_extract_patch_f32": (_I, [_P, _P, _I, _I, C.c_float, C.c_float, _I, _I, _P]),
    "klt_track_iterate_f32": (_I, [_P, C.c_float, C.c_float, _P, _P, _P, _I, _I, _P, _P, _P, _I, _I, C.c_float, C.c_float, C.c_float, _I,
                                   C.POINTER(C.c_float), C.POINTER(C.c_float), _PI, _PI]),
    "klt_comm_unique_id": (_I, [_P]),
    "klt_comm_init_rank": (_I, [_P, _I, _I, _P]),
    "klt_comm_destroy": (_I, [_P]),
    "klt_comm_info": (_I, [_P, _PI, _PI]),
    "klt_allgather_featbuf_async": (_I, [_P, _I, _I, _I]),
    "klt_gather_featbuf_async": (_I, [_P, _I, _I, _I, _I]),
    "klt_gatherv_featbuf_async": (_I, [_P, _I, _I, _PI, _I]),
    "klt_comm_set_timeout": (_I, [_P, C.c_double]),
    "klt_sendrecv_featbuf_async": (_I, [_P, _I, _I, _I, _I, _I]),
    "klt_comm_fence_async": (_I, [_P]),
    "klt_comm_fence_featbuf_async": (_I, [_P, _I]),
    "klt_comm_wait": (_I, [_P]),
    "klt_comm_allreduce_max": (_I, [_P, C.POINTER(C.c_double), _I]),
    "klt_timing_enable": (_I, [_P, _I]),
    "klt_timing_read": (_I, [_P, C.POINTER(KltKernelTime), _I]),
}

_lib = None


class KltBackendError(RuntimeError):
    pass


class KltOutOfMemory(KltBackendError, MemoryError):
    """KLT_ERR_NOMEM: a device or pinned-host allocation could not be had (the message names the size asked for).  Nothing is left
    half-allocated and the context stays usable: free something (`Context.device_free`, `slot_free`, drop frames) and repeat the call."""


class KltCommTimeout(KltBackendError):
    """KLT_ERR_TIMEOUT: a host-side wait for a collective gave up (a peer is gone or stuck).  The communicator is unusable from then
    on; the rank is expected to report and exit non-zero -- `exit_on_comm_timeout` does -- and is never restarted in place."""


def exit_on_comm_timeout(exc, code=3):
    """Print the timeout and leave the process at once (os._exit: no finalizer gets the chance to wait for a stream that is fenced
    behind the dead collective)."""
    import sys
    sys.stderr.write("pyfeaturetrack_amd: %s -- exiting with code %d\n" % (exc, code))
    sys.stderr.flush()
    os._exit(code)


def load_library(path=None):
    """Load libkltgpu.so and type every entry point.  Raises if it is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("KLT_GPU_LIB", LIB_PATH)
    if not os.path.exists(p):
        raise KltBackendError(
            "HIP backend %s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % p)
    lib = C.CDLL(p)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib
