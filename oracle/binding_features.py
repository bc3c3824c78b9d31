"""ctypes binding of the feature-extraction part of the CPU oracle (proslam_oracle_features.h).
TEST INFRASTRUCTURE ONLY."""
import ctypes as C

import numpy as np

from . import binding as ob

FEATURE_BORDER = 31  # cv::ORB edgeThreshold
ERR_KEYPOINTS = -10
SELECT_CANONICAL, SELECT_LIBSTDCXX = 0, 1


class ExtractorParams(C.Structure):
    _fields_ = [("detector_threshold", C.c_int32), ("enable_non_maximum_suppression", C.c_int32),
                ("target_number_of_keypoints", C.c_int32), ("number_of_detectors_vertical", C.c_int32),
                ("number_of_detectors_horizontal", C.c_int32), ("selection_order", C.c_int32)]


def extractor_params(threshold=15, nms=1, target=1000, vertical=3, horizontal=3, selection_order=SELECT_CANONICAL):
    """defaults of configurations/kitti.conf:229-255; selection_order: tie handling of the per-region cut
    (SELECT_LIBSTDCXX = the permutation of GNU std::sort, what the reference's pinned counts come from)"""
    return ExtractorParams(threshold, nms, target, vertical, horizontal, selection_order)


_bound = False


def _lib():
    global _bound
    L = ob.lib()
    if not _bound:
        vp = C.c_void_p
        L.orc_orb_pattern.restype = None
        L.orc_orb_pattern.argtypes = [vp]
        L.orc_gaussian_blur7.restype = None
        L.orc_gaussian_blur7.argtypes = [vp, C.c_int, C.c_int, vp]
        L.orc_std_sort_desc.restype = None
        L.orc_std_sort_desc.argtypes = [vp, C.c_int, vp]
        L.orc_fast_scores.restype = None
        L.orc_fast_scores.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp]
        L.orc_extract_features.restype = C.c_int
        L.orc_extract_features.argtypes = [C.POINTER(ExtractorParams), vp, C.c_int, C.c_int, vp, vp, vp, C.c_int]
        _bound = True
    return L


def orb_pattern():
    p = np.zeros(1024, np.int8)
    _lib().orc_orb_pattern(p.ctypes.data)
    return p.reshape(256, 4)


def gaussian_blur7(image):
    img = np.ascontiguousarray(image, np.uint8)
    out = np.zeros_like(img)
    _lib().orc_gaussian_blur7(img.ctypes.data, img.shape[0], img.shape[1], out.ctypes.data)
    return out


def std_sort_desc(response):
    """order[i] = original position of the element GNU std::sort (comparator a.response > b.response) leaves at i"""
    r = np.ascontiguousarray(response, np.int32)
    out = np.zeros(len(r), np.int32)
    _lib().orc_std_sort_desc(r.ctypes.data, len(r), out.ctypes.data)
    return out


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
