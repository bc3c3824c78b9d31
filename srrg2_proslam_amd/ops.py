"""Python host layer over the C-ABI (include/proslam_hip.h).

torch is used for device memory and streams only (plumbing); every operator below is a thin
call into libproslam_hip.so.  Host-array entry points mirror what a srrg2 plugin adapter calls
once per compute(); `*_batch` entry points keep B independent frames resident in HBM.
"""
import ctypes as C
import weakref

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
    """prs_context: one device, one stream, scratch. Not re-entrant (like the reference's finders).

    Stream policy: the `*_batch` operators read and write torch tensors, whose fills and copies run on torch's current
    stream; a context that launched on its own (non-blocking) stream would not be ordered against them.  By default the
    context therefore enqueues on torch's current stream of `device` (stream="torch").  stream="own" keeps the
    context's private stream: the caller then orders the two streams itself (events) or only uses host-array entry
    points, which synchronise internally."""

    def __init__(self, device=0, stream="torch"):
        lib = _lib.load()
        h = C.c_void_p()
        rc = lib.prs_context_create(int(device), C.byref(h))
        if rc < 0:
            raise ProslamHipError(rc, "prs_context_create(device=%d): %s" % (device, lib.prs_status_string(rc).decode()))
        self._h = h
        self.device = int(device)
        self._children = weakref.WeakSet()  # handles that hold a pointer to this context
        if stream == "torch":
            self.use_torch_stream()
        elif stream != "own":
            raise ValueError("stream must be 'torch' or 'own'")

    def close(self):
        if getattr(self, "_h", None):
            for child in list(self._children):  # a finder handle must never outlive its context
                child.close()
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

    def set_bruteforce_dense_phase(self, mode):
        """BF_DENSE_POPCOUNT (default) / BF_DENSE_MATRIX_WHEN_FULL / BF_DENSE_MATRIX: which kernels score the brute-force matcher's
        N_f x N_m pairs (prs_context_set_bruteforce_dense_phase; same results, the matrix cores pay only when candidates are rare)"""
        _check(self, _lib.load().prs_context_set_bruteforce_dense_phase(self._h, int(mode)), "prs_context_set_bruteforce_dense_phase")

    def enable_timing(self, on=True):
        """HIP-event timing of the aligner's two kernels inside align_batch (measurement only)"""
        _check(self, _lib.load().prs_context_enable_timing(self._h, 1 if on else 0), "prs_context_enable_timing")

    def align_timing(self):
        """-> dict(search_ms, gn_ms, search_launches, gn_launches) accumulated since enable_timing()"""
        a, b, c, d = C.c_double(0), C.c_double(0), C.c_int64(0), C.c_int64(0)
        _check(self, _lib.load().prs_context_get_align_timing(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)), "prs_context_get_align_timing")
        return {"search_ms": a.value, "gn_ms": b.value, "search_launches": c.value, "gn_launches": d.value}

    def align_round_timing(self):
        """-> (search_ms[16], gn_ms[16], batches): the timed launches by round of the batch (mean per batch = value / batches)"""
        a, b, n = (C.c_double * 16)(), (C.c_double * 16)(), C.c_int64(0)
        _check(self, _lib.load().prs_context_get_align_round_timing(self._h, a, b, C.byref(n)), "prs_context_get_align_round_timing")
        return list(a), list(b), n.value


def stereo_params(cfg_matcher, image_rows, image_cols=0):
    """image_cols is reserved (ignored)"""
    return StereoParams(
        float(cfg_matcher["maximum_descriptor_distance"]),
        float(cfg_matcher["maximum_distance_ratio_to_second_best"]),
        float(cfg_matcher["minimum_matching_ratio"]),
        int(cfg_matcher["maximum_disparity_pixels"]),
        int(cfg_matcher["epipolar_line_thickness_pixels"]),
        int(image_rows),
        int(image_cols),
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


# =================================================================================================
# projective correspondence finder + Gauss-Newton aligner
# =================================================================================================
from ._lib import AlignBatch, AlignerParams, AlignResult, PcfParams, PcfState, Projector  # noqa: E402


def pcf_params(cfg, **overrides):
    """prs_pcf_params from a configs.* dictionary (projective_finder + projector + camera)"""
    cam, pr = cfg["camera"], cfg["projector"]
    f = dict(cfg["projective_finder"])
    f.update(overrides)
    proj = Projector(cam["fx"], cam["fy"], cam["cx"], cam["cy"], int(cam["cols"]), int(cam["rows"]),
                     pr["range_min"], pr["range_max"])
    return PcfParams(f["maximum_descriptor_distance"], f["maximum_distance_ratio_to_second_best"],
                     f["minimum_matching_ratio"], f["minimum_descriptor_distance"],
                     f["descriptor_distance_step_size_pixels"], int(f["maximum_search_radius_pixels"]),
                     int(f["minimum_search_radius_pixels"]), int(f["search_radius_step_size_pixels"]),
                     int(f["minimum_number_of_iterations"]), f["maximum_estimate_change_norm_for_convergence"],
                     int(f["number_of_solver_iterations_per_projection"]), int(f["search_type"]), proj)


def aligner_params(cfg, mean_disparity=-1.0, stop_at_fixed_point=1, **overrides):
    """prs_aligner_params from a configs.* dictionary; mean_disparity < 0 = computed on the device"""
    cam, al = cfg["camera"], dict(cfg["aligner"])
    al.update(overrides)
    p = AlignerParams()
    p.factor_type = int(al["factor_type"])
    p.fx, p.fy, p.cx, p.cy = cam["fx"], cam["fy"], cam["cx"], cam["cy"]
    p.image_cols, p.image_rows = cam["cols"], cam["rows"]
    p.baseline_left_in_right_px[0] = -cam["fx"] * cam.get("baseline_m", 0.0)  # K * t_left_in_right
    for i in range(3):
        p.diagonal_info[i] = al["diagonal_info"][i]
    p.chi_threshold = al["chi_threshold"]
    p.enable_inverse_depth_weighting = int(al["enable_inverse_depth_weighting"])
    p.mean_disparity = mean_disparity
    p.damping = al["damping"]
    p.max_iterations = int(al["max_iterations"])
    p.min_num_inliers = int(al["min_num_inliers"])
    p.min_num_correspondences = int(al["min_num_correspondences"])
    p.stop_at_fixed_point = int(stop_at_fixed_point)
    # MultiAligner3DQR flags of the RGB-D configurations (icl.conf:50-64, tum.conf:90-104)
    p.enable_inlier_only_runs = int(al.get("enable_inlier_only_runs", 0))
    p.keep_only_inlier_correspondences = int(al.get("keep_only_inlier_correspondences", 0))
    p.inlier_only_iterations = int(al.get("inlier_only_iterations", 0))
    # readings of the external srrg2_solver arithmetic (0 = shipped; include/proslam_hip.h)
    p.kernel_weight_form = int(al.get("kernel_weight_form", 0))
    p.damping_form = int(al.get("damping_form", 0))
    p.translation_weight_form = int(al.get("translation_weight_form", 0))
    p.step_norm_exit = float(al.get("step_norm_exit", 0.0))  # opt-in: does less work than the reference
    if al.get("sensor_in_robot") is not None:
        set_sensor_in_robot(p, al["sensor_in_robot"])
    if al.get("motion_prior_info") is not None:
        set_motion_prior(p, al["motion_prior_info"])
    return p


def set_sensor_in_robot(p, S):
    """...WithSensor factor variants (aligner_slice_processor_projective.h:80-83): X is the robot's movingInFixed"""
    p.with_sensor = 1
    for i, v in enumerate(np.asarray(S, np.float32).reshape(16)):
        p.sensor_in_robot[i] = float(v)
    return p


def set_motion_prior(p, info=(1.0,) * 6):
    """AlignerSliceMotionModel3D stand-in (kitti.conf:747-772): prior on movingInFixed with diagonal information"""
    p.enable_motion_prior = 1
    for i in range(6):
        p.motion_prior_info[i] = float(info[i])
    return p


def info_scale_from_nopt(n_opt):
    n_opt = np.ascontiguousarray(n_opt, dtype=np.uint32)
    out = np.zeros(max(len(n_opt), 1), dtype=np.float32)
    _lib.load().prs_info_scale_from_nopt(_p(n_opt), len(n_opt), _p(out))
    return out[: len(n_opt)].copy()


def gn_step(ctx, H, b, damping, X, damping_form=0):
    H, b = _np(H, np.float32, (36,)), _np(b, np.float32, (6,))
    X = _np(X, np.float32, (16,)).copy()
    rc = _lib.load().prs_gn_step_ex(ctx._h, _p(H), _p(b), float(damping), int(damping_form), _p(X))
    _check(ctx, rc, "prs_gn_step")
    return X.reshape(4, 4), rc


class ProjectiveFinder:
    """host-array handle mirroring CorrespondenceFinderProjective{KDTree,Square,Circle,Rhombus}:
    set_fixed / set_moving / set_local_map_in_sensor / compute, state carried across calls"""

    def __init__(self, ctx, params):
        self.ctx = ctx
        h = C.c_void_p()
        _check(ctx, _lib.load().prs_pcf_create(ctx._h, C.byref(params), C.byref(h)), "prs_pcf_create")
        self._h = h
        self._n_fixed = 0
        ctx._children.add(self)

    def close(self):
        if getattr(self, "_h", None):
            if getattr(self.ctx, "_h", None):
                _lib.load().prs_pcf_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_params(self, params):
        _check(self.ctx, _lib.load().prs_pcf_set_params(self._h, C.byref(params)), "prs_pcf_set_params")

    def set_fixed(self, coords, desc):
        coords = np.ascontiguousarray(coords, dtype=np.float32)
        if coords.ndim != 2:
            coords = coords.reshape(-1, 2)
        d = _np(desc, np.uint8, (-1, 32))
        self._n_fixed = coords.shape[0]
        _check(self.ctx, _lib.load().prs_pcf_set_fixed(self._h, _p(coords), coords.shape[1], _p(d), coords.shape[0]), "prs_pcf_set_fixed")

    def set_moving(self, xyz, desc, info_scale=None):
        xyz = _np(xyz, np.float32, (-1, 3))
        d = _np(desc, np.uint8, (-1, 32))
        sc = None if info_scale is None else _np(info_scale, np.float32, (-1,))
        _check(self.ctx, _lib.load().prs_pcf_set_moving(self._h, _p(xyz), None if sc is None else _p(sc), _p(d), xyz.shape[0]), "prs_pcf_set_moving")

    def set_local_map_in_sensor(self, T):
        T = _np(T, np.float32, (16,))
        _check(self.ctx, _lib.load().prs_pcf_set_local_map_in_sensor(self._h, _p(T)), "prs_pcf_set_local_map_in_sensor")

    def set_search_radius(self, r):
        _check(self.ctx, _lib.load().prs_pcf_set_search_radius(self._h, int(r)), "prs_pcf_set_search_radius")

    def set_descriptor_distance(self, d):
        _check(self.ctx, _lib.load().prs_pcf_set_descriptor_distance(self._h, float(d)), "prs_pcf_set_descriptor_distance")

    def state(self):
        st = PcfState()
        _check(self.ctx, _lib.load().prs_pcf_get_state(self._h, C.byref(st)), "prs_pcf_get_state")
        return st

    @property
    def search_radius(self):
        return int(self.state().search_radius_pixels)

    @property
    def descriptor_distance(self):
        return float(self.state().descriptor_distance)

    @property
    def iteration(self):
        return int(self.state().current_iteration)

    @property
    def has_converged(self):
        return bool(self.state().has_converged)

    @property
    def num_recomputes(self):
        return int(self.state().num_recomputes)

    def local_map_in_sensor(self):
        return np.array(self.state().local_map_in_sensor, dtype=np.float32).reshape(4, 4)

    def compute(self):
        out = np.zeros(max(self._n_fixed, 1), dtype=CORR_DTYPE)
        n = C.c_int32(0)
        rc = _lib.load().prs_pcf_compute(self._h, _p(out), out.shape[0], C.byref(n))
        _check(self.ctx, rc, "prs_pcf_compute")
        return out[: n.value].copy(), rc

    def set_motion_prior_mean(self, Z):
        """mean of the motion prior (None = identity)"""
        z = None if Z is None else _np(Z, np.float32, (16,))
        _check(self.ctx, _lib.load().prs_pcf_set_motion_prior_mean(self._h, None if z is None else _p(z)), "prs_pcf_set_motion_prior_mean")

    def align(self, params, X_init, prior=None):
        """the whole per-frame loop; returns (X [4,4], correspondences, prs_align_result, warnings)"""
        X0 = _np(X_init, np.float32, (16,))
        X = np.zeros(16, dtype=np.float32)
        out = np.zeros(max(self._n_fixed, 1), dtype=CORR_DTYPE)
        n = C.c_int32(0)
        res = AlignResult()
        pr = None if prior is None else _np(np.concatenate([np.ravel(prior[0]), np.ravel(prior[1])]), np.float32, (42,))
        rc = _lib.load().prs_pcf_align(self._h, C.byref(params), _p(X0), None if pr is None else _p(pr), _p(X), _p(out), out.shape[0], C.byref(n), C.byref(res))
        _check(self.ctx, rc, "prs_pcf_align")
        return X.reshape(4, 4), out[: n.value].copy(), res, rc

    def linearize(self, params, X, corr):
        X = _np(X, np.float32, (16,))
        corr = np.ascontiguousarray(corr, dtype=CORR_DTYPE)
        res = AlignResult()
        rc = _lib.load().prs_pcf_linearize(self._h, C.byref(params), _p(X), _p(corr), len(corr), C.byref(res))
        _check(self.ctx, rc, "prs_pcf_linearize")
        return res


class AlignFrames:
    """B independent frames (sequences) resident in HBM for the fused finder + aligner kernel"""

    def __init__(self, device, batch, fixed_stride, moving_stride, with_prior=False):
        import torch
        dev = torch.device("cuda", device)
        self.batch, self.fixed_stride, self.moving_stride = int(batch), int(fixed_stride), int(moving_stride)
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)  # noqa: E731
        self.fixed = z((batch, fixed_stride, 4), torch.float32)
        self.fixed_desc = z((batch, fixed_stride, 32), torch.uint8)
        self.n_fixed = z((batch,), torch.int32)
        self.moving = z((batch, moving_stride, 4), torch.float32)
        self.moving_desc = z((batch, moving_stride, 32), torch.uint8)
        self.n_moving = z((batch,), torch.int32)
        self.inputs_changed = torch.ones((batch,), dtype=torch.uint8, device=dev)
        self.state = z((batch, C.sizeof(PcfState)), torch.uint8)
        self.X = z((batch, 16), torch.float32)
        self.corr = z((batch, fixed_stride, 3), torch.int32)
        self.n_corr = z((batch,), torch.int32)
        self.result = z((batch, C.sizeof(AlignResult)), torch.uint8)
        self.prior = z((batch, 42), torch.float32) if with_prior else None
        self.prior_mean = None  # optional [batch, 16] float32 device tensor: mean of the motion prior
        self.max_fixed = 0  # 0 = fixed_stride; smaller bound = less LDS per frame = more frames per CU
        self.reset_state()

    def reset_state(self):
        """fresh finder objects: zeroed dynamic thresholds, config_changed = 1, identity transforms"""
        import torch
        st = PcfState()
        st.config_changed = 1
        for i in (0, 5, 10, 15):
            st.local_map_in_sensor[i] = 1.0
            st.local_map_in_sensor_previous[i] = 1.0
        raw = np.frombuffer(bytes(st), dtype=np.uint8)
        self.state[:] = torch.from_numpy(raw.copy()).to(self.state.device)
        self.n_corr.zero_()

    def upload(self, b, fixed, fixed_desc, moving_xyz, info_scale, moving_desc, X_init):
        import torch
        fixed = np.ascontiguousarray(fixed, dtype=np.float32)
        nf, nm = fixed.shape[0], len(moving_xyz)
        f4 = np.zeros((nf, 4), dtype=np.float32)
        f4[:, : fixed.shape[1]] = fixed
        m4 = np.ones((nm, 4), dtype=np.float32)
        m4[:, :3] = moving_xyz
        if info_scale is not None:
            m4[:, 3] = info_scale
        if nf:
            self.fixed[b, :nf] = torch.from_numpy(f4)
            self.fixed_desc[b, :nf] = torch.from_numpy(np.ascontiguousarray(fixed_desc, dtype=np.uint8))
        if nm:
            self.moving[b, :nm] = torch.from_numpy(m4)
            self.moving_desc[b, :nm] = torch.from_numpy(np.ascontiguousarray(moving_desc, dtype=np.uint8))
        self.n_fixed[b] = nf
        self.n_moving[b] = nm
        self.X[b] = torch.from_numpy(np.ascontiguousarray(X_init, dtype=np.float32).reshape(16))

    def descriptor(self):
        d = AlignBatch()
        d.batch, d.fixed_stride, d.moving_stride = self.batch, self.fixed_stride, self.moving_stride
        d.fixed, d.fixed_desc, d.n_fixed = self.fixed.data_ptr(), self.fixed_desc.data_ptr(), self.n_fixed.data_ptr()
        d.moving, d.moving_desc, d.n_moving = self.moving.data_ptr(), self.moving_desc.data_ptr(), self.n_moving.data_ptr()
        d.inputs_changed, d.state, d.X = self.inputs_changed.data_ptr(), self.state.data_ptr(), self.X.data_ptr()
        d.corr, d.n_corr, d.result = self.corr.data_ptr(), self.n_corr.data_ptr(), self.result.data_ptr()
        d.prior = self.prior.data_ptr() if self.prior is not None else None
        d.prior_mean = self.prior_mean.data_ptr() if self.prior_mean is not None else None
        d.max_fixed = int(self.max_fixed)
        return d

    def result_of(self, b):
        raw = self.result[b].cpu().numpy().tobytes()
        return AlignResult.from_buffer_copy(raw)

    def state_of(self, b):
        raw = self.state[b].cpu().numpy().tobytes()
        return PcfState.from_buffer_copy(raw)

    def corr_of(self, b):
        n = int(self.n_corr[b].item())
        raw = self.corr[b, :n].cpu().numpy()
        out = np.zeros(n, dtype=CORR_DTYPE)
        out["fixed_idx"], out["moving_idx"] = raw[:, 0], raw[:, 1]
        out["response"] = raw[:, 2].view(np.float32)
        return out


def align_batch(ctx, finder_params, aligner_params_, frames, mode=_lib.MODE_ALIGN):
    """prs_align_batch_run: the finder + aligner loop of every frame of the batch (mode ALIGN blocks until it is complete)"""
    d = frames.descriptor()
    rc = _lib.load().prs_align_batch_run(ctx._h, C.byref(finder_params), C.byref(aligner_params_), C.byref(d), int(mode))
    _check(ctx, rc, "prs_align_batch_run")
    return rc


def align_batch_enqueue(ctx, finder_params, aligner_params_, frames, rounds=0):
    """prs_align_batch_enqueue: `rounds` search + Gauss-Newton rounds on the context stream, no host synchronisation"""
    d = frames.descriptor()
    rc = _lib.load().prs_align_batch_enqueue(ctx._h, C.byref(finder_params), C.byref(aligner_params_), C.byref(d), int(rounds))
    _check(ctx, rc, "prs_align_batch_enqueue")
    return rc


def align_batch_finish(ctx):
    """prs_align_batch_finish: wait for the enqueued batch, run more rounds for frames that are still pending"""
    rc = _lib.load().prs_align_batch_finish(ctx._h)
    _check(ctx, rc, "prs_align_batch_finish")
    return rc


def align_batch_rearm(ctx, replay_stream=None):
    """prs_align_batch_rearm{,_on}: after replaying a HIP graph captured around align_batch_enqueue, so that finish checks the replay
    (replay_stream: the hipStream_t handle the graph was launched on when that is not the capture stream)"""
    if replay_stream is None:
        rc = _lib.load().prs_align_batch_rearm(ctx._h)
    else:
        rc = _lib.load().prs_align_batch_rearm_on(ctx._h, C.c_void_p(int(replay_stream)))
    _check(ctx, rc, "prs_align_batch_rearm")
    return rc


# ---- scene clipper (SceneClipperProjective3D, mapping/scene_clipper_projective_3d.cpp:9-67) ----
def projector_params(cfg):
    """prs_projector from a config's camera + projector section"""
    return pcf_params(cfg).projector


def scene_clip(ctx, projector, robot_in_local_map, sensor_in_robot, scene_xyzw, scene_desc=None):
    """host arrays, one scene -> (clipped_xyzw [m,4], clipped_desc [m,32] | None, global_indices [m], warning bits)"""
    xyzw = _np(scene_xyzw, np.float32, (-1, 4))
    n = xyzw.shape[0]
    desc = None if scene_desc is None else _np(scene_desc, np.uint8, (-1, 32))
    R = _np(robot_in_local_map, np.float32, (16,))
    S = _np(sensor_in_robot, np.float32, (16,))
    out_xyzw = np.zeros((max(n, 1), 4), dtype=np.float32)
    out_desc = None if desc is None else np.zeros((max(n, 1), 32), dtype=np.uint8)
    out_idx = np.zeros(max(n, 1), dtype=np.int32)
    m = C.c_int32(0)
    rc = _lib.load().prs_scene_clip(ctx._h, C.byref(projector), _p(R), _p(S), _p(xyzw), None if desc is None else _p(desc), n,
                                    _p(out_xyzw), None if out_desc is None else _p(out_desc), _p(out_idx), out_idx.shape[0], C.byref(m))
    _check(ctx, rc, "prs_scene_clip")
    k = m.value
    return out_xyzw[:k].copy(), (None if out_desc is None else out_desc[:k].copy()), out_idx[:k].copy(), rc


class ClipScenes:
    """B local maps resident in HBM + the clipper's outputs (laid out like the aligner's moving cloud)"""

    def __init__(self, device, batch, stride, with_desc=True):
        import torch
        dev = torch.device("cuda", device)
        self.batch, self.stride = int(batch), int(stride)
        self.scene_xyzw = torch.zeros((batch, stride, 4), dtype=torch.float32, device=dev)
        self.scene_desc = torch.zeros((batch, stride, 32), dtype=torch.uint8, device=dev) if with_desc else None
        self.n_scene = torch.zeros((batch,), dtype=torch.int32, device=dev)
        self.robot_in_local_map = torch.eye(4, dtype=torch.float32, device=dev).repeat(batch, 1, 1).contiguous()
        self.clipped_xyzw = torch.zeros((batch, stride, 4), dtype=torch.float32, device=dev)
        self.clipped_desc = torch.zeros((batch, stride, 32), dtype=torch.uint8, device=dev) if with_desc else None
        self.global_indices = torch.zeros((batch, stride), dtype=torch.int32, device=dev)
        self.n_clipped = torch.zeros((batch,), dtype=torch.int32, device=dev)
        self.status = torch.zeros((batch,), dtype=torch.int32, device=dev)

    def upload(self, b, xyzw, desc, robot_in_local_map):
        import torch
        n = len(xyzw)
        dev = self.scene_xyzw.device
        if n:
            self.scene_xyzw[b, :n] = torch.from_numpy(np.ascontiguousarray(xyzw, dtype=np.float32).reshape(n, 4)).to(dev)
            if self.scene_desc is not None:
                self.scene_desc[b, :n] = torch.from_numpy(np.ascontiguousarray(desc, dtype=np.uint8).reshape(n, 32)).to(dev)
        self.n_scene[b] = n
        self.robot_in_local_map[b] = torch.from_numpy(np.ascontiguousarray(robot_in_local_map, dtype=np.float32).reshape(4, 4)).to(dev)

    def descriptor(self):
        d = _lib.ClipBatch()
        d.batch, d.stride = self.batch, self.stride
        d.scene_xyzw = self.scene_xyzw.data_ptr()
        d.scene_desc = self.scene_desc.data_ptr() if self.scene_desc is not None else None
        d.n_scene = self.n_scene.data_ptr()
        d.robot_in_local_map = self.robot_in_local_map.data_ptr()
        d.clipped_xyzw = self.clipped_xyzw.data_ptr()
        d.clipped_desc = self.clipped_desc.data_ptr() if self.clipped_desc is not None else None
        d.global_indices = self.global_indices.data_ptr()
        d.n_clipped = self.n_clipped.data_ptr()
        d.status = self.status.data_ptr()
        d.scene_n_opt = self.scene_n_opt.data_ptr() if getattr(self, "scene_n_opt", None) is not None else None
        return d

    def clipped_of(self, b):
        m = int(self.n_clipped[b].item())
        return (self.clipped_xyzw[b, :m].cpu().numpy(),
                None if self.clipped_desc is None else self.clipped_desc[b, :m].cpu().numpy(),
                self.global_indices[b, :m].cpu().numpy(), int(self.status[b].item()))


def scene_clip_batch(ctx, projector, sensor_in_robot, scenes):
    """enqueue the clipper for every scene of the batch on the context stream (asynchronous)"""
    S = _np(sensor_in_robot, np.float32, (16,))
    d = scenes.descriptor()
    rc = _lib.load().prs_scene_clip_batch(ctx._h, C.byref(projector), _p(S), C.byref(d))
    _check(ctx, rc, "prs_scene_clip_batch")
    return rc


# ---- bijective brute-force matcher (CF/correspondence_finder_descriptor_based_bruteforce_impl.cpp) ----
def bruteforce_params(maximum_descriptor_distance=50.0, maximum_distance_ratio_to_second_best=0.9, minimum_matching_ratio=0.25):
    """defaults of CF/correspondence_finder_descriptor_based_bruteforce.h:22-36"""
    return _lib.BruteforceParams(maximum_descriptor_distance, maximum_distance_ratio_to_second_best, minimum_matching_ratio)


def bruteforce_match(ctx, params, desc_fixed, desc_moving):
    """host arrays, one pair of clouds -> (correspondences [CORR_DTYPE] ordered by (response, fixed), warning bits)"""
    df, dm = _np(desc_fixed, np.uint8, (-1, 32)), _np(desc_moving, np.uint8, (-1, 32))
    nf, nm = df.shape[0], dm.shape[0]
    out = np.zeros(max(min(nf, nm), 1), dtype=CORR_DTYPE)
    n = C.c_int32(0)
    rc = _lib.load().prs_bruteforce_match(ctx._h, C.byref(params), _p(df), nf, _p(dm), nm, _p(out), out.shape[0], C.byref(n))
    _check(ctx, rc, "prs_bruteforce_match")
    return out[: n.value].copy(), rc


class BruteforceClouds:
    """B (fixed, moving) descriptor cloud pairs resident in HBM + the matcher's outputs"""

    def __init__(self, device, batch, fixed_stride, moving_stride, candidate_capacity=0):
        import torch
        dev = torch.device("cuda", device)
        self.batch, self.fixed_stride, self.moving_stride = int(batch), int(fixed_stride), int(moving_stride)
        self.candidate_capacity = int(candidate_capacity)
        self.fixed_desc = torch.zeros((batch, fixed_stride, 32), dtype=torch.uint8, device=dev)
        self.moving_desc = torch.zeros((batch, moving_stride, 32), dtype=torch.uint8, device=dev)
        self.n_fixed = torch.zeros((batch,), dtype=torch.int32, device=dev)
        self.n_moving = torch.zeros((batch,), dtype=torch.int32, device=dev)
        self.out_stride = min(self.fixed_stride, self.moving_stride)
        self.matches = torch.zeros((batch, self.out_stride, 3), dtype=torch.int32, device=dev)
        self.n_matches = torch.zeros((batch,), dtype=torch.int32, device=dev)
        self.status = torch.zeros((batch,), dtype=torch.int32, device=dev)

    def upload(self, b, desc_fixed, desc_moving):
        import torch
        dev = self.fixed_desc.device
        nf, nm = len(desc_fixed), len(desc_moving)
        if nf:
            self.fixed_desc[b, :nf] = torch.from_numpy(np.ascontiguousarray(desc_fixed, dtype=np.uint8).reshape(nf, 32)).to(dev)
        if nm:
            self.moving_desc[b, :nm] = torch.from_numpy(np.ascontiguousarray(desc_moving, dtype=np.uint8).reshape(nm, 32)).to(dev)
        self.n_fixed[b], self.n_moving[b] = nf, nm

    def descriptor(self):
        d = _lib.BruteforceBatch()
        d.batch, d.fixed_stride, d.moving_stride = self.batch, self.fixed_stride, self.moving_stride
        d.fixed_desc, d.n_fixed = self.fixed_desc.data_ptr(), self.n_fixed.data_ptr()
        d.moving_desc, d.n_moving = self.moving_desc.data_ptr(), self.n_moving.data_ptr()
        d.matches, d.n_matches, d.status = self.matches.data_ptr(), self.n_matches.data_ptr(), self.status.data_ptr()
        d.candidate_capacity = self.candidate_capacity
        return d

    def matches_of(self, b):
        n = int(self.n_matches[b].item())
        raw = self.matches[b, :n].cpu().numpy()
        out = np.zeros(n, dtype=CORR_DTYPE)
        out["fixed_idx"], out["moving_idx"] = raw[:, 0], raw[:, 1]
        out["response"] = raw[:, 2].view(np.float32)
        return out


def bruteforce_match_batch(ctx, params, clouds):
    """enqueue the matcher for every cloud pair of the batch on the context stream (asynchronous)"""
    d = clouds.descriptor()
    rc = _lib.load().prs_bruteforce_match_batch(ctx._h, C.byref(params), C.byref(d))
    _check(ctx, rc, "prs_bruteforce_match_batch")
    return rc


# ---- landmark estimators + projective mergers (mapping/mergers, mapping/landmarks) ----
EST_WEIGHTED_MEAN, EST_EKF, EST_SMOOTHER = 0, 1, 2
MERGER_STEREO_TRIANGULATION, MERGER_STEREO_EKF, MERGER_DEPTH_EKF = 0, 1, 2


class MapBatch:
    """B local maps resident in HBM (structure of arrays, torch tensors own the memory) + the
    per-frame inputs of the merger.  Row layouts follow include/proslam_hip.h prs_merge_batch."""

    MEAS_WORDS = 7    # prs_camera_measurement: 3 + 3 floats + frame index
    POSE_WORDS = 24   # prs_frame_pose: two 3x4 transforms

    def __init__(self, device, batch, capacity, max_measurements, max_frames, measurement_stride, corr_stride):
        import torch
        dev = torch.device("cuda", device)
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)
        self.batch, self.capacity = int(batch), int(capacity)
        self.max_measurements, self.max_frames = int(max_measurements), int(max_frames)
        self.measurement_stride, self.corr_stride = int(measurement_stride), int(corr_stride)
        self.coords = z((batch, capacity, 4), torch.float32)
        self.desc = z((batch, capacity, 32), torch.uint8)
        self.state = z((batch, capacity, 4), torch.float32)
        self.covariance = z((batch, capacity, 9), torch.float32)
        self.n_opt = z((batch, capacity), torch.int32)
        self.inlier = z((batch, capacity), torch.uint8)
        self.n_meas = z((batch, capacity), torch.int32)
        self.meas = z((batch, capacity, max(max_measurements, 1), self.MEAS_WORDS), torch.int32)
        self.poses = z((batch, max_frames, self.POSE_WORDS), torch.float32)
        self.n_points = z((batch,), torch.int32)
        self.measurement = z((batch, measurement_stride, 4), torch.float32)
        self.measurement_desc = z((batch, measurement_stride, 32), torch.uint8)
        self.n_measured = z((batch,), torch.int32)
        self.corr = z((batch, corr_stride, 3), torch.int32)
        self.n_corr = z((batch,), torch.int32)
        self.scene_index_map = None
        self.corr_from_aligner = 0
        self.measurement_in_world = torch.eye(4, dtype=torch.float32, device=dev).repeat(batch, 1, 1).contiguous()
        self.measurement_in_scene = torch.eye(4, dtype=torch.float32, device=dev).repeat(batch, 1, 1).contiguous()
        self.frame = z((batch,), torch.int32)
        self.result = z((batch, 3), torch.int32)

    def descriptor(self):
        d = _lib.MergeBatch()
        d.batch, d.capacity, d.max_measurements, d.max_frames = self.batch, self.capacity, self.max_measurements, self.max_frames
        for name in ("coords", "desc", "state", "covariance", "n_opt", "inlier", "n_meas", "poses", "n_points", "measurement",
                     "measurement_desc", "n_measured", "corr", "n_corr", "measurement_in_world", "measurement_in_scene", "frame", "result"):
            setattr(d, name, getattr(self, name).data_ptr())
        d.meas = self.meas.data_ptr() if self.max_measurements > 0 else None
        d.measurement_stride, d.corr_stride = self.measurement_stride, self.corr_stride
        d.scene_index_map = self.scene_index_map.data_ptr() if self.scene_index_map is not None else None
        d.corr_from_aligner = int(self.corr_from_aligner)
        return d


def merge_batch(ctx, params, maps):
    """enqueue MergerProjective_::compute for every (map, frame) pair of the batch (asynchronous)"""
    d = maps.descriptor()
    rc = _lib.load().prs_merge_batch_run(ctx._h, C.byref(params), C.byref(d))
    _check(ctx, rc, "prs_merge_batch_run")
    return rc


class MapHandle:
    """host-side merger object: one device-resident local map (prs_map), mirrors setScene / setMeasurement / compute of
    mapping/mergers/merger_projective.h"""

    def __init__(self, ctx, capacity, max_measurements=0, max_frames=64, max_measured=2048):
        self._ctx = ctx
        self._h = C.c_void_p()
        self.capacity = int(capacity)
        rc = _lib.load().prs_map_create(ctx._h, int(capacity), int(max_measurements), int(max_frames), int(max_measured), C.byref(self._h))
        _check(ctx, rc, "prs_map_create")
        ctx._children.add(self)  # the handle points into the context: Context.close() closes it first

    def close(self):
        if getattr(self, "_h", None):
            if getattr(self._ctx, "_h", None):  # (a context that is gone has already released the device memory's owner)
                _lib.load().prs_map_destroy(self._h)
            self._h = C.c_void_p()

    def reserve(self, capacity):
        """grow the landmark arrays in place (every landmark keeps its state, covariance and history)"""
        _check(self._ctx, _lib.load().prs_map_reserve(self._h, int(capacity)), "prs_map_reserve")
        self.capacity = max(self.capacity, int(capacity))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def clear(self):
        _check(self._ctx, _lib.load().prs_map_clear(self._h), "prs_map_clear")

    def size(self):
        n, f = C.c_int32(0), C.c_int32(0)
        _check(self._ctx, _lib.load().prs_map_size(self._h, C.byref(n), C.byref(f)), "prs_map_size")
        return n.value, f.value

    def set_scene(self, coords, desc, state=None, covariance=None, n_opt=None, first_measurement=None):
        c = _np(coords, np.float32, (-1, 3))
        n = c.shape[0]
        d = _np(desc, np.uint8, (-1, 32))
        st = None if state is None else _np(state, np.float32, (-1, 3))
        cov = None if covariance is None else _np(covariance, np.float32, (-1, 9))
        no = None if n_opt is None else _np(n_opt, np.uint32, (-1,))
        fm = None if first_measurement is None else np.ascontiguousarray(first_measurement)
        rc = _lib.load().prs_map_set_scene(self._h, _p(c), None if st is None else _p(st), None if cov is None else _p(cov), _p(d),
                                           None if no is None else _p(no), None if fm is None else fm.ctypes.data, n)
        _check(self._ctx, rc, "prs_map_set_scene")

    def set_frame_pose(self, frame, sensor_in_world):
        T = _np(sensor_in_world, np.float32, (16,))
        _check(self._ctx, _lib.load().prs_map_set_frame_pose(self._h, int(frame), _p(T)), "prs_map_set_frame_pose")

    def merge(self, params, measurement_in_world, measurement_in_scene, measurement, measurement_desc, corr, scene_index_map=None, corr_from_aligner=0):
        """-> (n_merged, n_added, status bits)"""
        dim = int(params.estimator.measurement_dim)
        z = _np(measurement, np.float32, (-1, dim))
        d = _np(measurement_desc, np.uint8, (-1, 32))
        c = np.ascontiguousarray(corr, dtype=CORR_DTYPE)
        im = None if scene_index_map is None else _np(scene_index_map, np.int32, (-1,))
        Tw, Ts = _np(measurement_in_world, np.float32, (16,)), _np(measurement_in_scene, np.float32, (16,))
        res = (C.c_int32 * 3)()
        rc = _lib.load().prs_map_merge(self._h, C.byref(params), _p(Tw), _p(Ts), _p(z), _p(d), z.shape[0], _p(c) if len(c) else None, len(c),
                                       None if im is None else _p(im), int(corr_from_aligner), res)
        _check(self._ctx, rc, "prs_map_merge")
        return int(res[0]), int(res[1]), int(res[2])

    def scene(self):
        """-> dict(coords [n,3], state [n,3], desc [n,32], n_opt [n], inlier [n])"""
        cap = self.capacity
        coords, state = np.zeros((cap, 3), np.float32), np.zeros((cap, 3), np.float32)
        desc, n_opt, inl = np.zeros((cap, 32), np.uint8), np.zeros(cap, np.uint32), np.zeros(cap, np.uint8)
        n = C.c_int32(0)
        rc = _lib.load().prs_map_get_scene(self._h, cap, _p(coords), _p(state), _p(desc), _p(n_opt), _p(inl), C.byref(n))
        _check(self._ctx, rc, "prs_map_get_scene")
        k = n.value
        return dict(coords=coords[:k].copy(), state=state[:k].copy(), desc=desc[:k].copy(), n_opt=n_opt[:k].copy(), inlier=inl[:k].copy())


def pose_compose_batch(ctx, prediction, X, pose_out):
    """pose_out[b] = prediction[b] * X[b]^-1 on device tensors of shape [B, 16] / [B, 4, 4] (asynchronous)"""
    batch = int(prediction.shape[0])
    rc = _lib.load().prs_pose_compose_batch(ctx._h, batch, prediction.data_ptr(), X.data_ptr(), pose_out.data_ptr())
    _check(ctx, rc, "prs_pose_compose_batch")
    return rc


def motion_predict_batch(ctx, pose_prev2, pose_prev1, pose_pred):
    """MotionModelConstantVelocity3D on the device: pose_pred = pose_prev1 * (pose_prev2^-1 * pose_prev1), [B, 4, 4] float32"""
    batch = int(pose_prev1.shape[0])
    rc = _lib.load().prs_motion_predict_batch(ctx._h, batch, pose_prev2.data_ptr(), pose_prev1.data_ptr(), pose_pred.data_ptr())
    _check(ctx, rc, "prs_motion_predict_batch")
    return rc


# ---- intensity feature extraction (sensor_processing/feature_extractors) ----
SELECT_CANONICAL, SELECT_LIBSTDCXX = 0, 1
BF_DENSE_POPCOUNT, BF_DENSE_MATRIX_WHEN_FULL, BF_DENSE_MATRIX = 0, 1, 2  # include/proslam_hip.h PRS_BF_DENSE_*


def extractor_params(threshold=15, nms=1, target=1000, vertical=3, horizontal=3, selection_order=SELECT_CANONICAL, max_raw_detections=0):
    """defaults of configurations/kitti.conf:229-255.  selection_order: tie handling of the per-region cut
    (SELECT_LIBSTDCXX = GNU std::sort's permutation, what a GCC build of the reference does; slower);
    max_raw_detections: FAST detections per image the selection holds (0 = 8192, at most 32768)"""
    return _lib.ExtractorParams(threshold, nms, target, vertical, horizontal, selection_order, max_raw_detections)


def extract_features(ctx, params, image, capacity=4096):
    """host arrays, one 8-bit image [rows, cols] -> (uv [n, 2] f32, intensity [n] f32, descriptors [n, 32] u8); synchronises"""
    img = np.ascontiguousarray(image, dtype=np.uint8)
    rows, cols = img.shape
    uv = np.zeros((capacity, 2), dtype=np.float32)
    inten = np.zeros(capacity, dtype=np.float32)
    desc = np.zeros((capacity, 32), dtype=np.uint8)
    n = C.c_int32(0)
    rc = _lib.load().prs_extract_features(ctx._h, C.byref(params), _p(img), rows, cols, cols, _p(uv), _p(inten), _p(desc), capacity, C.byref(n))
    _check(ctx, rc, "prs_extract_features")
    k = n.value
    return uv[:k].copy(), inten[:k].copy(), desc[:k].copy()


def selftest_reciprocal(ctx):
    """(operands that differ from 1.0f / x, operands that took the short form) over all 2^32 float bit patterns (prs_selftest_reciprocal)"""
    counts = np.zeros(2, dtype=np.uint64)
    rc = _lib.load().prs_selftest_reciprocal(ctx._h, _p(counts))
    _check(ctx, rc, "prs_selftest_reciprocal")
    return int(counts[0]), int(counts[1])


def selection_order(ctx, response):
    """responses (1..255) of one region's keypoints in detection order -> the permutation the reference's std::sort leaves
    (order[k] = keypoint at position k; intensity_feature_extractor_binned.cpp:182-186); host arrays, synchronises"""
    r = np.ascontiguousarray(response, dtype=np.uint8)
    order = np.zeros(len(r), dtype=np.int32)
    rc = _lib.load().prs_selection_order(ctx._h, _p(r), len(r), _p(order))
    _check(ctx, rc, "prs_selection_order")
    return order


def extract_features_batch(ctx, params, images, keypoints, descriptors, n_features, status, intensity=None):
    """images: uint8 device tensor [B, rows, cols(pitch)]; outputs are device tensors laid out like the stereo
    matcher's inputs (keypoints [B, stride, 2] f32, descriptors [B, stride, 32] u8, n_features [B] i32)."""
    d = _lib.ExtractBatch()
    d.batch, d.rows, d.cols, d.pitch = int(images.shape[0]), int(images.shape[1]), int(images.shape[2]), int(images.stride(1))
    d.images = images.data_ptr()
    d.stride = int(keypoints.shape[1])
    d.keypoints, d.descriptors = keypoints.data_ptr(), descriptors.data_ptr()
    d.intensity = intensity.data_ptr() if intensity is not None else None
    d.n_features, d.status = n_features.data_ptr(), status.data_ptr()
    rc = _lib.load().prs_extract_features_batch(ctx._h, C.byref(params), C.byref(d))
    _check(ctx, rc, "prs_extract_features_batch")
    return rc
