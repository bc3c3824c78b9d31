"""Oracle pins for the scene clipper: the behaviour gates of
tests/test_scene_clippers.cpp:7-462 restated on a synthetic dense ICL-shaped cloud (the reference
uses the lr-kt0 depth image, which is not in the repository: its exact counts are not reproducible,
the properties are)."""
import numpy as np

from tests import helpers as hp

I4 = np.eye(4, dtype=np.float32)


def test_C1_no_motion_keeps_everything_in_front(oracle):
    xyzw = hp.icl_dense_scene()
    assert xyzw.shape[0] == 307200  # tests/test_scene_clippers.cpp:28
    before = xyzw.copy()
    out, _, idx, flags = oracle.scene_clip(hp.clip_projector(oracle), I4, I4, xyzw)
    assert np.array_equal(xyzw, before)  # :27-28 the world points are not modified
    # :30-33: "some points might be lost due to numerical imprecision"
    assert 0.995 * 307200 <= len(out) <= 307200 and flags == 0
    assert np.all(out[:, 2] > 0)  # :34-37
    assert np.all(np.diff(idx) > 0)  # the projector's sequential loop: ascending source order
    assert np.array_equal(out[:, :3], xyzw[idx, :3])  # identity pose: camera frame = local map frame
    assert np.array_equal(out[:, 3], xyzw[idx, 3])


def test_C2_half_turn_about_optical_axis_keeps_the_cloud(oracle):
    # :40-72 image flipped upside down, all points still visible (principal point is the centre)
    xyzw = hp.icl_dense_scene()
    out, _, idx, _ = oracle.scene_clip(hp.clip_projector(oracle), hp.rot("z", np.pi), I4, xyzw)
    assert 0.99 * 307200 <= len(out) <= 307200
    assert np.all(out[:, 2] > 0)


def test_C3_half_turn_about_x_sees_nothing(oracle):
    # :74-100
    xyzw = hp.icl_dense_scene()
    out, _, idx, flags = oracle.scene_clip(hp.clip_projector(oracle), hp.rot("x", np.pi), I4, xyzw)
    assert len(out) == 0 and flags == oracle.WARN_NO_PROJECTION


def test_C4_quarter_roll_sees_a_strict_subset(oracle):
    # :102-128 (49872 of 307200 on the real image)
    xyzw = hp.icl_dense_scene()
    out, _, idx, _ = oracle.scene_clip(hp.clip_projector(oracle), hp.rot("x", np.pi / 4), I4, xyzw)
    assert 0 < len(out) < 307200 // 2


def test_C5_translation_backward_keeps_all_forward_loses_some(oracle):
    # :130-184
    xyzw = hp.icl_dense_scene()
    back, fwd = I4.copy(), I4.copy()
    back[2, 3], fwd[2, 3] = -1.0, 1.0
    proj = hp.clip_projector(oracle)
    out_b, _, _, _ = oracle.scene_clip(proj, back, I4, xyzw)
    out_f, _, idx_f, _ = oracle.scene_clip(proj, fwd, I4, xyzw)
    assert len(out_b) == 307200  # :155
    assert 0 < len(out_f) < 307200  # :183
    # the kept points are expressed in the moved camera
    assert np.allclose(out_f[:, 2], xyzw[idx_f, 2] - 1.0, atol=1e-6)


def test_C6_descriptors_indices_and_sensor_offset(oracle):
    # :186-219 sparse cloud with descriptors; global indices (scene_clipper_projective_3d.h:32-34)
    rng = np.random.default_rng(5)
    full = hp.icl_dense_scene(1)
    pick = rng.choice(len(full), 321, replace=False)
    xyzw = full[np.sort(pick)].copy()
    xyzw[:, 3] = rng.uniform(1, 4, len(xyzw)).astype(np.float32)
    desc = rng.integers(0, 256, (len(xyzw), 32), dtype=np.uint8)
    proj = hp.clip_projector(oracle)
    R = hp.rot("y", 0.3)
    R[0, 3] = 0.4
    out, odesc, idx, _ = oracle.scene_clip(proj, R, I4, xyzw, desc)
    assert 0 < len(out) < 321
    assert np.array_equal(odesc, desc[idx]) and np.array_equal(out[:, 3], xyzw[idx, 3])
    # a sensor offset: same visibility decided by robot_in_local_map * sensor_in_robot, output in the robot frame
    S = hp.rot("z", 0.1)
    S[:3, 3] = (0.2, -0.1, 0.05)
    out_s, _, idx_s, _ = oracle.scene_clip(proj, R, S, xyzw, desc)
    cam = np.linalg.inv(R.astype(np.float64) @ S.astype(np.float64))
    pc = (cam[:3, :3] @ xyzw[idx_s, :3].T.astype(np.float64)).T + cam[:3, 3]
    pr = (S[:3, :3].astype(np.float64) @ pc.T).T + S[:3, 3]
    assert np.allclose(out_s[:, :3], pr, atol=1e-4)


def test_C7_empty_scene_is_a_warning_and_touches_nothing(oracle):
    # scene_clipper_projective_3d.cpp:21-28
    out, _, idx, flags = oracle.scene_clip(hp.clip_projector(oracle), I4, I4, np.zeros((0, 4), np.float32))
    assert len(out) == 0 and flags == oracle.WARN_EMPTY_INPUT
