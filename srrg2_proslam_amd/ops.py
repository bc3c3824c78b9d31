"""Python host layer over the C-ABI (include/proslam_hip.h).

torch is used for device memory and streams only (plumbing); every operator below is a thin
call into libproslam_hip.so.  Host-array entry points mirror what a srrg2 plugin adapter calls
once per compute(); `*_batch` entry points keep B independent frames resident in HBM.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import ProslamHipError, StereoBatch, StereoParams, TriangulatorParams

CORR_DTYPE = np.dtype([("fixed_idx", np.int32), ("moving_idx", np.int32), ("response", np.float32)])


def _check(ctx, rc, what):
    if rc < 0:
        lib = _lib.load()
        msg = lib.prs_last_error(ctx._h) if ctx is not None and ctx._h else b""
        raise ProslamHipError(rc, "%s: %s (%s)" % (what, lib.prs_status_string(rc).decode(), (msg or b"").decode()))
    return rc


class Context:
    """prs_context: one device, one stream, scratch. Not re-entrant (like the reference's finders)."""

    def __init__(self, device=0):
        lib = _lib.load()
        h = C.c_void_p()
        rc = lib.prs_context_create(int(device), C.byref(h))
        if rc < 0:
            raise ProslamHipError(rc, "prs_context_create(device=%d): %s" % (device, lib.prs_status_string(rc).decode()))
        self._h = h
        self.device = int(device)

    def close(self):
        if getattr(self, "_h", None):
            _lib.load().prs_context_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def use_torch_stream(self):
        """enqueue on torch's current stream so torch ops and these kernels are ordered"""
        import torch
        s = torch.cuda.current_stream(self.device).cuda_stream
        _check(self, _lib.load().prs_context_set_stream(self._h, C.c_void_p(s)), "prs_context_set_stream")

    def synchronize(self):
        _check(self, _lib.load().prs_context_synchronize(self._h), "prs_context_synchronize")


def stereo_params(cfg_matcher, image_rows):
    return StereoParams(
        float(cfg_matcher["maximum_descriptor_distance"]),
        float(cfg_matcher["maximum_distance_ratio_to_second_best"]),
        float(cfg_matcher["minimum_matching_ratio"]),
        int(cfg_matcher["maximum_disparity_pixels"]),
        int(cfg_matcher["epipolar_line_thickness_pixels"]),
        int(image_rows),
    )


def triangulator_params(cfg):
    cam, tri = cfg["camera"], cfg["triangulator"]
    return TriangulatorParams(cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["fx"] * cam["baseline_m"],
                              tri["minimum_disparity_pixels"], tri["infinity_depth_meters"])


def _np(a, dtype, shape):
    a = np.ascontiguousarray(a, dtype=dtype)
    return a.reshape(shape)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def stereo_match(ctx, params, uv_left, desc_left, uv_right, desc_right):
    """host arrays, one frame -> (correspondences [CORR_DTYPE], warning bits)
    fixed = left keypoints, moving = right keypoints (raw_data_preprocessor_stereo_projective.cpp:99-102)"""
    uvl, uvr = _np(uv_left, np.float32, (-1, 2)), _np(uv_right, np.float32, (-1, 2))
    dl, dr = _np(desc_left, np.uint8, (-1, 32)), _np(desc_right, np.uint8, (-1, 32))
    nl, nr = uvl.shape[0], uvr.shape[0]
    out = np.zeros(max(nl, 1), dtype=CORR_DTYPE)
    n = C.c_int32(0)
    rc = _lib.load().prs_stereo_match(ctx._h, C.byref(params), _p(uvl), _p(dl), nl, _p(uvr), _p(dr), nr, _p(out), out.shape[0], C.byref(n))
    _check(ctx, rc, "prs_stereo_match")
    return out[: n.value].copy(), rc


def triangulate(ctx, params, uvuv):
    """host arrays -> (xyz [n,3], valid [n])"""
    uvuv = _np(uvuv, np.float32, (-1, 4))
    n = uvuv.shape[0]
    xyz = np.zeros((max(n, 1), 3), dtype=np.float32)
    valid = np.zeros(max(n, 1), dtype=np.uint8)
    rc = _lib.load().prs_triangulate(ctx._h, C.byref(params), _p(uvuv), n, _p(xyz), _p(valid))
    _check(ctx, rc, "prs_triangulate")
    return xyz[:n].copy(), valid[:n].copy()


class StereoFrames:
    """B stereo pairs resident in HBM (torch tensors own the memory) + the outputs of the
    batched matcher with its fused adaptor/triangulator epilogue."""

    def __init__(self, device, batch, stride, epilogue=True):
        import torch
        dev = torch.device("cuda", device)
        self.batch, self.stride = int(batch), int(stride)
        self.left_kp = torch.zeros((batch, stride, 2), dtype=torch.float32, device=dev)
        self.right_kp = torch.zeros((batch, stride, 2), dtype=torch.float32, device=dev)
        self.left_desc = torch.zeros((batch, stride, 32), dtype=torch.uint8, device=dev)
        self.right_desc = torch.zeros((batch, stride, 32), dtype=torch.uint8, device=dev)
        self.n_left = torch.zeros((batch,), dtype=torch.int32, device=dev)
        self.n_right = torch.zeros((batch,), dtype=torch.int32, device=dev)
        self.matches = torch.zeros((batch, stride, 3), dtype=torch.int32, device=dev)  # prs_corr as 3 x 32 bit
        self.n_matches = torch.zeros((batch,), dtype=torch.int32, device=dev)
        self.status = torch.zeros((batch,), dtype=torch.int32, device=dev)
        self.epilogue = epilogue
        if epilogue:
            self.fixed_uvuv = torch.zeros((batch, stride, 4), dtype=torch.float32, device=dev)
            self.fixed_desc = torch.zeros((batch, stride, 32), dtype=torch.uint8, device=dev)
            self.n_fixed = torch.zeros((batch,), dtype=torch.int32, device=dev)
            self.fixed_xyz = torch.zeros((batch, stride, 4), dtype=torch.float32, device=dev)

    def upload(self, b, uv_left, desc_left, uv_right, desc_right):
        import torch
        nl, nr = len(uv_left), len(uv_right)
        if nl:
            self.left_kp[b, :nl] = torch.from_numpy(np.ascontiguousarray(uv_left, dtype=np.float32))
            self.left_desc[b, :nl] = torch.from_numpy(np.ascontiguousarray(desc_left, dtype=np.uint8))
        if nr:
            self.right_kp[b, :nr] = torch.from_numpy(np.ascontiguousarray(uv_right, dtype=np.float32))
            self.right_desc[b, :nr] = torch.from_numpy(np.ascontiguousarray(desc_right, dtype=np.uint8))
        self.n_left[b] = nl
        self.n_right[b] = nr

    def descriptor(self, tri_params=None):
        d = StereoBatch()
        d.batch, d.stride = self.batch, self.stride
        d.left_kp, d.left_desc, d.n_left = self.left_kp.data_ptr(), self.left_desc.data_ptr(), self.n_left.data_ptr()
        d.right_kp, d.right_desc, d.n_right = self.right_kp.data_ptr(), self.right_desc.data_ptr(), self.n_right.data_ptr()
        d.matches, d.n_matches, d.status = self.matches.data_ptr(), self.n_matches.data_ptr(), self.status.data_ptr()
        if self.epilogue and tri_params is not None:
            d.fixed_uvuv, d.fixed_desc = self.fixed_uvuv.data_ptr(), self.fixed_desc.data_ptr()
            d.n_fixed, d.fixed_xyz = self.n_fixed.data_ptr(), self.fixed_xyz.data_ptr()
            d.triangulator = C.pointer(tri_params)
        return d

    def matches_of(self, b):
        """download frame b's correspondences as a CORR_DTYPE array"""
        n = int(self.n_matches[b].item())
        raw = self.matches[b, :n].cpu().numpy()
        out = np.zeros(n, dtype=CORR_DTYPE)
        out["fixed_idx"] = raw[:, 0]
        out["moving_idx"] = raw[:, 1]
        out["response"] = raw[:, 2].view(np.float32)
        return out


def stereo_match_batch(ctx, params, frames, tri_params=None):
    """enqueue the batched matcher on the context stream (asynchronous)"""
    d = frames.descriptor(tri_params)
    rc = _lib.load().prs_stereo_match_batch(ctx._h, C.byref(params), C.byref(d))
    _check(ctx, rc, "prs_stereo_match_batch")
    return rc
