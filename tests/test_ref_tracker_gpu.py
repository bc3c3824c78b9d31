"""The tracker scenarios of tests/test_trackers.cpp on the reference's own images, every stage on the HIP path through the
C-ABI (extractor, epipolar matcher, scene clipper, projective finder + aligner, merger; all buffers device-resident between
stages), against the reference's bounds AND frame by frame against the CPU checker running the same loop."""
import ctypes as C

import numpy as np
import pytest
import torch

import ref_tracker as rt
from srrg2_proslam_amd import _lib, ops
from test_ref_pins_gpu import HipBackend
from test_ref_tracker import SUCCESS, OracleStages
from test_ref_pins import OracleBackend

pytestmark = pytest.mark.gpu
STRIDE = 1024


class HipStages:
    name = "hip"

    def __init__(self, ctx):
        self.ctx = ctx
        ctx.use_torch_stream()
        self.dev = torch.device("cuda", ctx.device)
        self.af = None

    def reset(self):
        self.af = None

    def _t(self, a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(self.dev)

    def predict(self, prev, pose):
        out = torch.zeros((1, 4, 4), dtype=torch.float32, device=self.dev)
        ops.motion_predict_batch(self.ctx, self._t(np.asarray(prev, np.float32).reshape(1, 4, 4)), self._t(np.asarray(pose, np.float32).reshape(1, 4, 4)), out)
        return out[0].cpu().numpy()

    def compose(self, guess, X):
        out = torch.zeros((1, 4, 4), dtype=torch.float32, device=self.dev)
        ops.pose_compose_batch(self.ctx, self._t(np.asarray(guess, np.float32).reshape(1, 4, 4)), self._t(np.asarray(X, np.float32).reshape(1, 16)), out)
        return out[0].cpu().numpy()

    def new_map(self, cfg, merger, capacity):
        po = OracleStages().new_map(cfg, merger, capacity)["params"]  # parameter block only: same layout on both sides
        assert C.sizeof(_lib.MergerParams) == C.sizeof(type(po))
        pg = _lib.MergerParams()
        C.memmove(C.byref(pg), C.byref(po), C.sizeof(pg))
        maps = ops.MapBatch(self.ctx.device, 1, capacity, merger.get("max_measurements", 0), 16, STRIDE, STRIDE)
        clip = ops.ClipScenes(self.ctx.device, 1, capacity)
        clip.scene_xyzw, clip.scene_desc, clip.n_scene, clip.scene_n_opt = maps.coords, maps.desc, maps.n_points, maps.n_opt
        af = ops.AlignFrames(self.ctx.device, 1, STRIDE, capacity)
        af.moving, af.moving_desc, af.n_moving = clip.clipped_xyzw, clip.clipped_desc, clip.n_clipped
        maps.measurement, maps.measurement_desc, maps.n_measured = af.fixed, af.fixed_desc, af.n_fixed
        maps.corr, maps.corr_from_aligner, maps.scene_index_map = af.corr, 1, clip.global_indices
        self.af = af
        return dict(params=pg, maps=maps, clip=clip, af=af, frame=0)

    def map_size(self, m):
        return int(m["maps"].n_points[0].item())

    def _set_measurement(self, af, fixed, desc):
        n = len(fixed)
        f4 = np.zeros((n, 4), np.float32)
        f4[:, : fixed.shape[1]] = fixed
        af.fixed[0, :n] = self._t(f4)
        af.fixed_desc[0, :n] = self._t(desc)
        af.n_fixed[0] = n

    def clip(self, cfg, pose, m):
        clip = m["clip"]
        clip.robot_in_local_map[0] = self._t(np.asarray(pose, np.float32).reshape(4, 4))
        ops.scene_clip_batch(self.ctx, ops.pcf_params(cfg).projector, rt.I4, clip)
        xyzw, desc, idx, status = clip.clipped_of(0)
        assert status >= 0
        return xyzw, desc, idx

    def align(self, cfg, fixed, desc, xyzw, cdesc, guess, prior_info):
        af = self.af  # the moving cloud is the clipper's output, already in place
        self._set_measurement(af, fixed, desc)
        if cfg.get("bruteforce_finder"):
            # descriptor-based brute-force finder in the slice: one correspondence vector (brute-force kernel), then
            # max_iterations x (prs_pcf_linearize, prs_gn_step) through the host-pointer entry points
            corr = ops.bruteforce_match(self.ctx, ops.bruteforce_params(*cfg["bruteforce_finder"]), desc, cdesc)[0]
            gf = ops.ProjectiveFinder(self.ctx, ops.pcf_params(cfg))
            gf.set_fixed(fixed, desc)
            gf.set_moving(xyzw[:, :3], cdesc, xyzw[:, 3])
            ap = ops.aligner_params(cfg, mean_disparity=-1.0, stop_at_fixed_point=0)
            X = np.asarray(guess, np.float32).reshape(4, 4).copy()
            for _ in range(cfg["aligner"]["max_iterations"]):
                res = gf.linearize(ap, X, corr)
                X, _ = ops.gn_step(self.ctx, np.array(res.H, np.float32), np.array(res.b, np.float32), cfg["aligner"]["damping"], X)
            gf.close()
            n = len(corr)  # the merger reads the aligner's correspondence vector in place
            af.corr[0, :n] = self._t(np.stack([corr["fixed_idx"], corr["moving_idx"], corr["response"].view(np.int32)], axis=1).astype(np.int32))
            af.n_corr[0] = n
            return X, corr, int(res.num_inliers >= cfg["aligner"]["min_num_inliers"]), res.num_inliers
        af.X[0] = self._t(np.asarray(guess, np.float32).reshape(16))
        ap = ops.aligner_params(cfg, mean_disparity=-1.0, stop_at_fixed_point=0)
        if prior_info > 0:
            ops.set_motion_prior(ap, (prior_info,) * 6)
        ops.align_batch(self.ctx, ops.pcf_params(cfg), ap, af)  # finder state carried in af.state
        self.ctx.synchronize()
        res = af.result_of(0)
        assert res.warnings >= 0
        return af.X[0].cpu().numpy().reshape(4, 4), af.corr_of(0), res.status, res.num_inliers

    def merge(self, m, pose, fixed, desc, corr, idx):
        maps, af = m["maps"], m["af"]
        self._set_measurement(af, fixed, desc)
        maps.n_corr = af.n_corr if corr is not None else torch.zeros((1,), dtype=torch.int32, device=self.dev)
        T = self._t(np.asarray(pose, np.float32).reshape(4, 4))
        maps.measurement_in_world[0], maps.measurement_in_scene[0] = T, T
        maps.frame.fill_(m["frame"])
        ops.merge_batch(self.ctx, m["params"], maps)
        self.ctx.synchronize()
        got = maps.result[0].cpu().numpy()
        assert int(got[2]) >= 0, got
        m["frame"] += 1
        return int(got[0]), int(got[1])


@pytest.fixture(scope="module")
def B(hip_ctx):
    return HipBackend(hip_ctx)


@pytest.fixture(scope="module")
def S(hip_ctx):
    return HipStages(hip_ctx)


def _same_as_checker(log, ref):
    for k, (e, o) in enumerate(zip(log, ref)):
        for key in ("n_measured", "map_size", "merged", "added", "status", "inliers", "n_corr", "n_clipped"):
            assert e.get(key) == o.get(key), (k, key, e.get(key), o.get(key))
        assert np.array_equal(np.asarray(e["pose"], np.float32).view(np.uint32), np.asarray(o["pose"], np.float32).view(np.uint32)), (k, e["pose"], o["pose"])


@pytest.mark.parametrize("dataset", ["kitti", "icl"])
def test_same_frame_three_times(S, B, oracle, dataset):
    log = rt.same_frame_three_times(S, B, dataset)
    assert 0 < log[0]["map_size"] <= log[0]["n_measured"]
    for e in log[1:]:
        assert e["n_measured"] == log[0]["n_measured"] and e["map_size"] == log[0]["map_size"] and e["status"] == SUCCESS
        assert np.linalg.norm(rt.rp.t2tnq(np.asarray(e["pose"], np.float64))) < 1e-5
    _same_as_checker(log, rt.same_frame_three_times(OracleStages(), OracleBackend(), dataset))


@pytest.mark.parametrize("no_merges", [True, False])
@pytest.mark.parametrize("motion_model", [False, True])
def test_kitti_00_to_04(S, B, oracle, no_merges, motion_model):
    kw = dict(use_prediction=True, prior_info=1.0) if motion_model else dict(use_prediction=False)
    log, error = rt.kitti_00_to_04(S, B, no_merges, **kw)
    assert all(e["status"] == SUCCESS for e in log[1:])
    from test_ref_tracker import REFERENCE_BOUND
    assert np.all(np.abs(error[:3]) < REFERENCE_BOUND) and np.all(np.abs(error[3:]) < 0.01), error
    ref, _ = rt.kitti_00_to_04(OracleStages(), OracleBackend(), no_merges, **kw)
    _same_as_checker(log, ref)


@pytest.mark.parametrize("kind", ["ekf", "smoother", "bruteforce_ekf"])
def test_kitti_00_to_04_other_mergers(S, B, oracle, kind):
    """tests/test_trackers.cpp:473-576, :578-682, :684-783 on the HIP path (bounds: tests/test_ref_tracker.py), frame by frame equal to the CPU checker"""
    from test_ref_tracker import REFERENCE_BOUND
    log, error = rt.kitti_00_to_04(S, B, False, kind)
    assert all(e["status"] == SUCCESS for e in log[1:])
    assert np.all(np.abs(error[:3]) < REFERENCE_BOUND) and np.all(np.abs(error[3:]) < 0.01), error
    ref, _ = rt.kitti_00_to_04(OracleStages(), OracleBackend(), False, kind)
    _same_as_checker(log, ref)


def test_icl_00_01_50(S, B, oracle):
    log, error = rt.icl_00_01_50(S, B)
    assert all(e["status"] == SUCCESS for e in log[1:])
    assert np.all(np.abs(error[:3]) < 0.02) and np.all(np.abs(error[3:]) < 0.01), error
    ref, _ = rt.icl_00_01_50(OracleStages(), OracleBackend())
    _same_as_checker(log, ref)
