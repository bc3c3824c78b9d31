"""ctypes loader for libproslam_hip.so (the HIP/gfx950 product library).

There is NO CPU fallback: if the shared library is missing or cannot be loaded, importing the
operators fails loudly.  Build it with `python -c "import __graft_entry__ as g; g.build()"` or
`make -C srrg2_proslam_amd/csrc`.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libproslam_hip.so")

# status codes (include/proslam_hip.h)
OK = 0
WARN_EMPTY_INPUT = 1
WARN_NO_MATCHES = 2
WARN_LOW_RATIO = 4
WARN_RETRIED = 8
WARN_TRACK_LOST = 16
WARN_NO_PROJECTION = 32
ERR_NULL = -1
ERR_CAPACITY = -2
ERR_HIP = -3
ERR_RANGE = -4
ERR_UNSUPPORTED = -5
ERR_NO_DEVICE = -6


class StereoParams(C.Structure):
    """prs_stereo_params"""
    _fields_ = [
        ("maximum_descriptor_distance", C.c_float),
        ("maximum_distance_ratio_to_second_best", C.c_float),
        ("minimum_matching_ratio", C.c_float),
        ("maximum_disparity_pixels", C.c_int32),
        ("epipolar_line_thickness_pixels", C.c_int32),
        ("image_rows", C.c_int32),
    ]


class TriangulatorParams(C.Structure):
    """prs_triangulator_params"""
    _fields_ = [
        ("fx", C.c_float),
        ("fy", C.c_float),
        ("cx", C.c_float),
        ("cy", C.c_float),
        ("b_x", C.c_float),
        ("minimum_disparity_pixels", C.c_float),
        ("infinity_depth_meters", C.c_float),
    ]


class StereoBatch(C.Structure):
    """prs_stereo_batch (device pointers)"""
    _fields_ = [
        ("batch", C.c_int32),
        ("stride", C.c_int32),
        ("left_kp", C.c_void_p),
        ("left_desc", C.c_void_p),
        ("n_left", C.c_void_p),
        ("right_kp", C.c_void_p),
        ("right_desc", C.c_void_p),
        ("n_right", C.c_void_p),
        ("matches", C.c_void_p),
        ("n_matches", C.c_void_p),
        ("status", C.c_void_p),
        ("fixed_uvuv", C.c_void_p),
        ("fixed_desc", C.c_void_p),
        ("n_fixed", C.c_void_p),
        ("fixed_xyz", C.c_void_p),
        ("triangulator", C.POINTER(TriangulatorParams)),
    ]


# every symbol include/proslam_hip.h declares: (restype, argtypes)
_vp = C.c_void_p
_i32p = C.POINTER(C.c_int32)
SYMBOLS = {
    "prs_version": (C.c_int, []),
    "prs_status_string": (C.c_char_p, [C.c_int]),
    "prs_context_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "prs_context_destroy": (C.c_int, [_vp]),
    "prs_context_set_stream": (C.c_int, [_vp, _vp]),
    "prs_context_use_own_stream": (C.c_int, [_vp]),
    "prs_context_synchronize": (C.c_int, [_vp]),
    "prs_last_error": (C.c_char_p, [_vp]),
    "prs_stereo_match": (C.c_int, [_vp, C.POINTER(StereoParams), _vp, _vp, C.c_int32, _vp, _vp, C.c_int32, _vp, C.c_int32, _i32p]),
    "prs_stereo_match_batch": (C.c_int, [_vp, C.POINTER(StereoParams), C.POINTER(StereoBatch)]),
    "prs_triangulate": (C.c_int, [_vp, C.POINTER(TriangulatorParams), _vp, C.c_int32, _vp, _vp]),
    "prs_triangulate_dev": (C.c_int, [_vp, C.POINTER(TriangulatorParams), _vp, C.c_int64, _vp]),
}

_lib = None


class ProslamHipError(RuntimeError):
    """hard error from the C-ABI (status < 0); the reference throws std::runtime_error here"""

    def __init__(self, status, message):
        super().__init__("libproslam_hip status %d: %s" % (status, message))
        self.status = status


def load():
    """dlopen the product library and bind every declared symbol; raises if it is absent"""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libproslam_hip.so is not built (%s). The HIP extension is mandatory: run "
            "`python -c 'import __graft_entry__ as g; g.build()'` -- there is no CPU fallback." % LIB_PATH)
    # torch bundles its own libamdhip64 (same SONAME as /opt/rocm's).  Import it FIRST so the
    # dynamic loader resolves our DT_NEEDED against the runtime torch already mapped: one HIP
    # runtime per process, so torch tensors / streams and this library agree.  Two runtime
    # copies in one process fail HSA initialisation for the second one.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib
