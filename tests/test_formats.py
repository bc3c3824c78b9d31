"""Harness formats: trajectory writers in the reference's KITTI / TUM layouts
(apps/app_benchmark.cpp:195-262) and the `.conf` reader subset.  CPU only."""
import os

import numpy as np
import pytest

from srrg2_proslam_amd import configs, formats

REFERENCE_CONFS = "/root/reference/configurations"


def _pose(rng):
    a = rng.normal(size=3)
    a *= rng.uniform(0.0, 3.1) / np.linalg.norm(a)
    th = np.linalg.norm(a)
    k = a / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    T = np.eye(4)
    T[:3, :3] = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K
    T[:3, 3] = rng.normal(size=3) * 10
    return T.astype(np.float32)


def test_kitti_line_layout_is_the_reference_s():
    T = np.eye(4, dtype=np.float32)
    T[0, 3], T[1, 3], T[2, 3] = 1.5, -0.25, 1.0 / 3.0
    line = formats.kitti_line(T)
    # std::fixed, setprecision(9), one space after EVERY value (also the last), float32 widened to double
    assert line == "1.000000000 0.000000000 0.000000000 1.500000000 0.000000000 1.000000000 0.000000000 -0.250000000 " \
                   "0.000000000 0.000000000 1.000000000 0.333333343 "


def test_tum_line_layout_and_quaternion_order():
    T = np.eye(4, dtype=np.float32)
    T[:3, :3] = [[0, -1, 0], [1, 0, 0], [0, 0, 1]]  # +90 degrees about z: q = (0, 0, sin 45, cos 45)
    T[:3, 3] = [1, 2, 3]
    line = formats.tum_line(1403636579.763555527, T)
    v = line.split(" ")
    assert v[-1] == "" and len(v) == 9  # trailing space
    assert v[0] == "1403636579.763555527" and v[1:4] == ["1.000000000", "2.000000000", "3.000000000"]
    q = np.array(v[4:8], np.float64)
    assert np.allclose(q, [0, 0, np.sqrt(0.5), np.sqrt(0.5)], atol=1e-7)


def test_quaternion_of_every_branch_round_trips(tmp_path):
    rng = np.random.default_rng(3)
    poses = [_pose(rng) for _ in range(200)]
    # rotations by ~180 degrees exercise the three largest-diagonal branches
    for axis in range(3):
        R = -np.eye(3)
        R[axis, axis] = 1.0
        T = np.eye(4, dtype=np.float32)
        T[:3, :3] = R
        poses.append(T)
    stamped = [(100.0 + 0.1 * i, T) for i, T in enumerate(poses)]
    for T in poses:
        q = formats.rotation_to_quaternion_xyzw(T[:3, :3])
        assert abs(np.linalg.norm(q) - 1.0) < 1e-5
    formats.write_trajectory_tum(tmp_path / "t.txt", stamped)
    ts, back = formats.read_trajectory_tum(tmp_path / "t.txt")
    assert np.allclose(ts, [s for s, _ in stamped])
    assert np.allclose(back, np.array(poses, np.float64), atol=2e-5)
    formats.write_trajectory_kitti(tmp_path / "k.txt", stamped)
    back = formats.read_trajectory_kitti(tmp_path / "k.txt")
    assert np.allclose(back, np.array(poses, np.float64), atol=1e-9)  # nine decimals of a float32


def test_unroll_orders_by_timestamp_and_first_insert_wins():
    rng = np.random.default_rng(4)
    K0, K1 = _pose(rng), _pose(rng)
    a, b, c, d = (_pose(rng) for _ in range(4))
    out = formats.unroll_trajectory([(K1, [(3.0, c), (2.0, b)]), (K0, [(1.0, a), (2.0, d)])])
    assert [s for s, _ in out] == [1.0, 2.0, 3.0]
    assert np.array_equal(out[0][1], (K0 @ a).astype(np.float32))
    assert np.array_equal(out[1][1], (K1 @ b).astype(np.float32))  # timestamp 2.0 was inserted by the first map; K0 * d is dropped
    assert np.array_equal(out[2][1], (K1 @ c).astype(np.float32))


def test_writers_order_by_timestamp_and_fail_loudly(tmp_path):
    rng = np.random.default_rng(5)
    a, b = _pose(rng), _pose(rng)
    formats.write_trajectory_kitti(tmp_path / "k.txt", [(2.0, b), (1.0, a)])
    lines = open(tmp_path / "k.txt").read().split("\n")
    assert lines[0] == formats.kitti_line(a) and lines[1] == formats.kitti_line(b) and lines[2] == ""
    with pytest.raises(OSError):
        formats.write_trajectory_tum(tmp_path / "no_such_dir" / "t.txt", [(1.0, a)])
    (tmp_path / "bad.txt").write_text("1 2 3\n")
    with pytest.raises(ValueError):
        formats.read_trajectory_kitti(tmp_path / "bad.txt")


SMALL_CONF = """
"PointProjectorPinhole" {
  "#id" : 7,
  // near plane
  "range_min" : 0.5,
  "range_max" : 40
 }

"RobustifierSaturated" { "#id" : 9, "chi_threshold" : 12.5 }

"CorrespondenceFinderProjectiveSquare4D3D" {
  "#id" : 3,
  "name" : "finder // not a comment",
  "maximum_descriptor_distance" : 60,
  "projector" : { "#pointer" : 7 }
 }

"AlignerSliceProcessorProjectiveStereo" {
  "#id" : 2,
  "diagonal_info_matrix" : [ 1, 2, 1 ],
  "min_num_correspondences" : 4,
  "finder" : { "#pointer" : 3 },
  "projector" : { "#pointer" : 7 },
  "robustifier" : { "#pointer" : 9 }
 }

"MultiAligner3DQR" {
  "#id" : 1,
  "name" : "aligner",
  "max_iterations" : 20,
  "min_num_inliers" : 5,
  "slice_processors" : [ { "#pointer" : 2 }, { "#pointer" : -1 } ],
  "termination_criteria" : { "#pointer" : -1 }
 }
"""


def test_conf_reader_records_pointers_and_comments():
    c = formats.parse_conf(SMALL_CONF)
    assert [r.class_name for r in c.records] == ["PointProjectorPinhole", "RobustifierSaturated", "CorrespondenceFinderProjectiveSquare4D3D",
                                                 "AlignerSliceProcessorProjectiveStereo", "MultiAligner3DQR"]
    assert c.by_id[7]["range_min"] == 0.5 and c.by_name["aligner"].id == 1
    assert c.by_id[3]["name"] == "finder // not a comment"  # `//` inside a string is text
    assert c.deref(c.by_id[1]["termination_criteria"]) is None  # -1 = no object
    assert c.follow(c.by_id[2], "finder", "projector") is c.by_id[7]
    hp = formats.hot_path_params(c)
    assert hp["projective_finder"] == {"maximum_descriptor_distance": 60, "search_type": configs.SEARCH_SQUARE}
    assert hp["projector"] == {"range_min": 0.5, "range_max": 40}
    assert hp["aligner"] == {"factor_type": configs.FACTOR_STEREO, "diagonal_info": (1, 2, 1), "min_num_correspondences": 4,
                             "chi_threshold": 12.5, "max_iterations": 20, "min_num_inliers": 5}
    with pytest.raises(ValueError):
        formats.parse_conf('"Broken" { "a" : }')
    with pytest.raises(ValueError):
        formats.parse_conf('{ "no_class_name" : 1 }')


@pytest.mark.skipif(not os.path.isdir(REFERENCE_CONFS), reason="the reference tree is only present in the build container")
@pytest.mark.parametrize("name", ["kitti", "euroc", "icl", "tum"])
def test_shipped_configurations_agree_with_the_carried_parameter_sets(name):
    """`configs.py` carries the hot-path values of the shipped `.conf` files by hand; this reads the files themselves."""
    hp = formats.hot_path_params(formats.read_conf(os.path.join(REFERENCE_CONFS, name + ".conf")))
    carried = configs.get(name)
    checked = 0
    for group, values in hp.items():
        for key, value in values.items():
            if key not in carried.get(group, {}):
                continue
            have = carried[group][key]
            if isinstance(value, tuple):
                assert tuple(float(x) for x in have) == tuple(float(x) for x in value), (group, key)
            else:
                assert float(have) == pytest.approx(float(value), rel=1e-6), (group, key)
            checked += 1
    assert checked >= 20
