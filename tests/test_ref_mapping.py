"""CPU checker on the scene-clipper and merger gtests of the reference, on its own KITTI / ICL data (scenarios: tests/ref_mapping.py)."""
import numpy as np
import pytest

import ref_mapping as rm
from oracle import binding as ob, binding_mapping as om
from test_ref_pins import OracleBackend


@pytest.fixture(scope="module")
def B():
    return OracleBackend()


def oracle_clip(K, rmin, rmax, T, xyz):
    proj = ob.Projector(K["fx"], K["fy"], K["cx"], K["cy"], K["cols"], K["rows"], rmin, rmax)
    xyzw = np.concatenate([xyz, np.ones((len(xyz), 1), np.float32)], axis=1).astype(np.float32)
    cx, _, gi, flags = ob.scene_clip(proj, T, rm.I4, xyzw, None)
    assert flags >= 0
    return cx, gi


def test_scene_clipper_counts(B):
    """tests/test_scene_clippers.cpp:7-462: 49872 / 136022 / 51 / 242 / 52 visible points, nothing or everything for the trivial motions"""
    for name, K, rmin, rmax, T, pts, want in rm.clipper_cases(B):
        cx, gi = oracle_clip(K, rmin, rmax, T, pts)
        if isinstance(want, tuple):
            assert want[0] <= len(cx) <= want[1], (name, len(cx))
        else:
            assert len(cx) == want, (name, len(cx), want)
        assert np.all(cx[:, 2] > 0) and np.all(np.diff(gi) > 0)  # :33-36: visible points lie ahead of the camera; source order kept
        assert len(pts) in (321, 145, 307200)  # "this operation must not modify the world points"


def run_oracle_case(case):
    m = case["map"].copy()
    poses = om.pose_table(case["n_frames"])
    om.set_pose(poses, 0, rm.I4)
    rc, res = om.merge(case["params"], case["T"], case["T"], poses, 1, m, case["fixed"], case["desc"], case["corr"])
    assert rc == 0
    return m, poses, res


def test_merger_cases(B):
    """tests/test_mergers.cpp:248-780: ICL depth EKF 321 -> 321 / 337 points, KITTI triangulation (weighted mean, smoother) and stereo EKF mergers:
    a cloud merged with its own measurements stays put (1e-5), merging frame 01 grows the scene by no more than the measurements"""
    for case in rm.merger_cases(B):
        n0 = case["map"].n_points
        before = case["map"].coords[:n0, :3].astype(np.float64).copy()
        m, _, res = run_oracle_case(case)
        rm.check_merge_result(case, n0, before, m.n_points, m.coords, len(case["fixed"]))
        assert res.n_merged > 0
