"""ctypes binding of the feature-extraction part of the CPU oracle (proslam_oracle_features.h).
TEST INFRASTRUCTURE ONLY."""
import ctypes as C

import numpy as np

from . import binding as ob

FEATURE_BORDER = 17
ERR_KEYPOINTS = -10


class ExtractorParams(C.Structure):
    _fields_ = [("detector_threshold", C.c_int32), ("enable_non_maximum_suppression", C.c_int32),
                ("target_number_of_keypoints", C.c_int32), ("number_of_detectors_vertical", C.c_int32),
                ("number_of_detectors_horizontal", C.c_int32)]


def extractor_params(threshold=15, nms=1, target=1000, vertical=3, horizontal=3):
    """defaults of configurations/kitti.conf:229-255"""
    return ExtractorParams(threshold, nms, target, vertical, horizontal)


_bound = False


def _lib():
    global _bound
    L = ob.lib()
    if not _bound:
        vp = C.c_void_p
        L.orc_brief_pattern.restype = None
        L.orc_brief_pattern.argtypes = [vp]
        L.orc_fast_scores.restype = None
        L.orc_fast_scores.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp]
        L.orc_extract_features.restype = C.c_int
        L.orc_extract_features.argtypes = [C.POINTER(ExtractorParams), vp, C.c_int, C.c_int, vp, vp, vp, C.c_int]
        _bound = True
    return L


def brief_pattern():
    p = np.zeros(1024, np.int8)
    _lib().orc_brief_pattern(p.ctypes.data)
    return p.reshape(256, 4)


def fast_scores(image, threshold):
    img = np.ascontiguousarray(image, np.uint8)
    out = np.zeros_like(img)
    _lib().orc_fast_scores(img.ctypes.data, img.shape[0], img.shape[1], int(threshold), out.ctypes.data)
    return out


def extract_features(params, image, capacity=4096):
    """-> (uv [n,2] f32, intensity [n] f32, descriptors [n,32] u8) or raises on overflow"""
    img = np.ascontiguousarray(image, np.uint8)
    uv = np.zeros((capacity, 2), np.float32)
    inten = np.zeros(capacity, np.float32)
    desc = np.zeros((capacity, 32), np.uint8)
    n = _lib().orc_extract_features(C.byref(params), img.ctypes.data, img.shape[0], img.shape[1], uv.ctypes.data, inten.ctypes.data, desc.ctypes.data, capacity)
    if n < 0:
        raise RuntimeError("orc_extract_features error %d" % n)
    return uv[:n].copy(), inten[:n].copy(), desc[:n].copy()
