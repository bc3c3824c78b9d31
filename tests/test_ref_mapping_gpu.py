"""HIP scene clipper and mergers on the reference's gtest scenarios (tests/ref_mapping.py): the reference's asserted counts on the
device results, and every output equal to the CPU checker's."""
import numpy as np
import pytest
import torch

import ref_mapping as rm
from srrg2_proslam_amd import ops
from test_mapping_gpu import _assert_map_equal, _gpu_params, _upload_frame, _upload_map
from test_ref_mapping import oracle_clip, run_oracle_case
from test_ref_pins_gpu import HipBackend

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def B(hip_ctx):
    return HipBackend(hip_ctx)


def test_scene_clipper_counts(B, hip_ctx, oracle):
    for name, K, rmin, rmax, T, pts, want in rm.clipper_cases(B):
        proj = ops.Projector(K["fx"], K["fy"], K["cx"], K["cy"], K["cols"], K["rows"], rmin, rmax)
        xyzw = np.concatenate([pts, np.ones((len(pts), 1), np.float32)], axis=1).astype(np.float32)
        cx, _, gi, rc = ops.scene_clip(hip_ctx, proj, T, rm.I4, xyzw, None)
        assert rc >= 0
        if isinstance(want, tuple):
            assert want[0] <= len(cx) <= want[1], (name, len(cx))
        else:
            assert len(cx) == want, (name, len(cx), want)
        ocx, ogi = oracle_clip(K, rmin, rmax, T, pts)
        assert np.array_equal(gi, ogi) and np.array_equal(cx.view(np.uint32), ocx.view(np.uint32)), name


def test_merger_cases(B, hip_ctx, oracle):
    for case in rm.merger_cases(B):
        m0 = case["map"]
        n0 = m0.n_points
        before = m0.coords[:n0, :3].astype(np.float64).copy()
        om_, poses, res = run_oracle_case(case)
        maps = ops.MapBatch(0, 1, m0.capacity, m0.max_measurements, case["n_frames"], 1024, 1024)
        poses0 = poses.copy()
        poses0[1:] = 0
        _upload_map(maps, 0, m0, poses0)
        _upload_frame(maps, 0, case["fixed"], case["desc"], case["corr"], case["T"], case["T"], 1)
        ops.merge_batch(hip_ctx, _gpu_params(case["params"]), maps)
        hip_ctx.synchronize()
        got = maps.result[0].cpu().numpy()
        assert (int(got[0]), int(got[1]), int(got[2])) == (res.n_merged, res.n_added, res.flags), (case["name"], got)
        n = int(maps.n_points[0].item())
        rm.check_merge_result(case, n0, before, n, maps.coords[0, :n].cpu().numpy(), len(case["fixed"]))
        _assert_map_equal(maps, 0, om_, poses, 2)


def test_merger_cases_through_the_host_handle(B, hip_ctx, oracle):
    """the same gtest scenarios through prs_map_* (the stateful object an adapter's merger owns): setScene, compute, read back"""
    from oracle import binding_mapping as om
    for case in rm.merger_cases(B):
        m0 = case["map"]
        n0 = m0.n_points
        before = m0.coords[:n0, :3].astype(np.float64).copy()
        om_, poses, res = run_oracle_case(case)
        h = ops.MapHandle(hip_ctx, m0.capacity, m0.max_measurements, max_frames=case["n_frames"], max_measured=1024)
        first = None
        if m0.max_measurements > 0:
            first = np.ascontiguousarray(m0.meas[:n0, 0])
            assert first.dtype.itemsize == 28
        h.set_scene(m0.coords[:n0, :3], m0.desc[:n0], state=m0.state[:n0, :3], covariance=m0.covariance[:n0], n_opt=m0.n_opt[:n0], first_measurement=first)
        h.set_frame_pose(0, rm.I4)
        assert h.size() == (n0, 1)
        got = h.merge(_gpu_params(case["params"]), case["T"], case["T"], case["fixed"], case["desc"], case["corr"])
        assert got == (res.n_merged, res.n_added, res.flags), (case["name"], got)
        sc = h.scene()
        n = len(sc["coords"])
        assert h.size() == (n, 2) and n == om_.n_points
        rm.check_merge_result(case, n0, before, n, sc["coords"], len(case["fixed"]))
        bits = lambda a: np.ascontiguousarray(a, np.float32).view(np.uint32)  # noqa: E731
        assert np.array_equal(bits(sc["coords"]), bits(om_.coords[:n, :3])) and np.array_equal(bits(sc["state"]), bits(om_.state[:n, :3])), case["name"]
        assert np.array_equal(sc["desc"], om_.desc[:n]) and np.array_equal(sc["n_opt"], om_.n_opt[:n]) and np.array_equal(sc["inlier"], om_.inlier[:n])
        h.close()
    # a map grown in place between two merges keeps every landmark's state, covariance, counters and history
    cases = {c["name"]: c for c in rm.merger_cases(B)}
    for kind in ("weighted_mean", "smoother", "stereo_ekf"):
        c0, c1 = cases["kitti_%s_00_to_00" % kind], cases["kitti_%s_00_to_01" % kind]
        m0, n0 = c0["map"], c0["map"].n_points
        out = []
        for grow in (False, True):
            h = ops.MapHandle(hip_ctx, n0 + 300 if grow else 4096, m0.max_measurements, max_frames=4, max_measured=1024)
            first = np.ascontiguousarray(m0.meas[:n0, 0]) if m0.max_measurements > 0 else None
            h.set_scene(m0.coords[:n0, :3], m0.desc[:n0], state=m0.state[:n0, :3], covariance=m0.covariance[:n0], n_opt=m0.n_opt[:n0], first_measurement=first)
            h.set_frame_pose(0, rm.I4)
            h.merge(_gpu_params(c0["params"]), c0["T"], c0["T"], c0["fixed"], c0["desc"], c0["corr"])
            if grow:
                h.reserve(4096)
            got = h.merge(_gpu_params(c1["params"]), c1["T"], c1["T"], c1["fixed"], c1["desc"], c1["corr"])
            out.append((got, h.scene()))
            h.close()
        assert out[0][0] == out[1][0], kind
        for key in ("coords", "state", "desc", "n_opt", "inlier"):
            assert np.array_equal(np.ascontiguousarray(out[0][1][key]).view(np.uint8), np.ascontiguousarray(out[1][1][key]).view(np.uint8)), (kind, key)
    # loud errors: more measurements than the handle holds, a full pose table
    h = ops.MapHandle(hip_ctx, 64, 0, max_frames=1, max_measured=8)
    case = rm.merger_cases(B)[0]
    with pytest.raises(ops.ProslamHipError) as ei:
        h.merge(_gpu_params(case["params"]), rm.I4, rm.I4, case["fixed"], case["desc"], case["corr"][:4])
    assert ei.value.status == ops._lib.ERR_CAPACITY
    h.close()
