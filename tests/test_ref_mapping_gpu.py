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
