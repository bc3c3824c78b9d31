"""GPU parity of the scene clipper through the C-ABI: clipped coordinates bit-exact
as float32 bit patterns, descriptors, carried column and global indices identical INCLUDING order."""
import numpy as np
import pytest

from srrg2_proslam_amd import _lib, ops
from tests import helpers as hp

pytestmark = pytest.mark.gpu
I4 = np.eye(4, dtype=np.float32)


def _proj(oracle):
    po = hp.clip_projector(oracle)
    return po, _lib.Projector(po.fx, po.fy, po.cx, po.cy, po.canvas_cols, po.canvas_rows, po.range_min, po.range_max)


def _same(a, b):
    assert len(a[0]) == len(b[0])
    assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32))
    if a[1] is not None:
        assert np.array_equal(a[1], b[1])
    assert np.array_equal(a[2], b[2])


@pytest.mark.parametrize("pose", ["identity", "half_z", "half_x", "quarter_x", "back", "forward", "general"])
def test_dense_cloud_tile_parallel_path(oracle, hip_ctx, pose):
    # 307200 points, one scene: the count + scatter launch shape (tests/test_scene_clippers.cpp:7-184)
    xyzw = hp.icl_dense_scene()
    R = {"identity": I4, "half_z": hp.rot("z", np.pi), "half_x": hp.rot("x", np.pi), "quarter_x": hp.rot("x", np.pi / 4)}.get(pose)
    if R is None:
        R = I4.copy()
        if pose == "back":
            R[2, 3] = -1.0
        elif pose == "forward":
            R[2, 3] = 1.0
        else:
            R = hp.rot("y", 0.2) @ hp.rot("x", -0.1)
            R[:3, 3] = (0.3, -0.2, 0.5)
    po, pg = _proj(oracle)
    ref = oracle.scene_clip(po, R, I4, xyzw)
    got = ops.scene_clip(hip_ctx, pg, R, I4, xyzw)
    _same(ref, got)
    assert ref[3] == got[3]


def test_sparse_with_descriptors_and_sensor_offset(oracle, hip_ctx):
    rng = np.random.default_rng(11)
    full = hp.icl_dense_scene(2)
    for n in (1, 63, 64, 65, 321, 1023, 1024, 1025, 2047, 2049, 5000):
        xyzw = full[np.sort(rng.choice(len(full), n, replace=False))].copy()
        xyzw[:, 3] = rng.uniform(1, 4, n).astype(np.float32)
        desc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        R = hp.rot("y", 0.3)
        R[0, 3] = 0.4
        S = hp.rot("z", 0.1)
        S[:3, 3] = (0.2, -0.1, 0.05)
        po, pg = _proj(oracle)
        for sensor in (I4, S):
            ref = oracle.scene_clip(po, R, sensor, xyzw, desc)
            got = ops.scene_clip(hip_ctx, pg, R, sensor, xyzw, desc)
            _same(ref, got)
            assert ref[3] == got[3]


def test_batched_walk_ragged_empty_and_blind_scenes(oracle, hip_ctx):
    rng = np.random.default_rng(3)
    full = hp.icl_dense_scene(4)
    sizes = [0, 1, 700, 2000, 2600, 3000, 1024, 0, 2048]
    B, stride = len(sizes), 3000
    scenes = ops.ClipScenes(0, B, stride)
    inputs = []
    for b, n in enumerate(sizes):
        xyzw = full[np.sort(rng.choice(len(full), n, replace=False))].copy() if n else np.zeros((0, 4), np.float32)
        desc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        R = hp.rot("y", rng.uniform(-0.4, 0.4)) @ hp.rot("x", rng.uniform(-0.2, 0.2))
        R[:3, 3] = rng.uniform(-0.5, 0.5, 3)
        if b == 4:
            R = hp.rot("x", np.pi)  # sees nothing
        inputs.append((xyzw, desc, R.astype(np.float32)))
        scenes.upload(b, xyzw, desc, R)
    S = hp.rot("z", -0.05)
    S[:3, 3] = (0.1, 0.0, -0.02)
    po, pg = _proj(oracle)
    scenes.n_clipped.fill_(-7)
    ops.scene_clip_batch(hip_ctx, pg, S, scenes)
    hip_ctx.synchronize()
    for b, (xyzw, desc, R) in enumerate(inputs):
        ref = oracle.scene_clip(po, R, S, xyzw, desc)
        if len(xyzw) == 0:
            # empty full scene: warning, nothing touched (scene_clipper_projective_3d.cpp:21-28)
            assert int(scenes.status[b].item()) == _lib_warn("EMPTY") and int(scenes.n_clipped[b].item()) == -7
            continue
        got = scenes.clipped_of(b)
        _same(ref, got)
        assert got[3] == ref[3]
    assert int(scenes.status[4].item()) == _lib_warn("NOPROJ") and int(scenes.n_clipped[4].item()) == 0


def _lib_warn(which):
    return {"EMPTY": 1, "NOPROJ": 32}[which]


def test_few_long_scenes_take_the_tile_parallel_shape(oracle, hip_ctx):
    rng = np.random.default_rng(8)
    full = hp.icl_dense_scene(6)
    sizes = [20000, 0, 12345]
    scenes = ops.ClipScenes(0, len(sizes), 20000, with_desc=False)
    inputs = []
    for b, n in enumerate(sizes):
        xyzw = full[np.sort(rng.choice(len(full), n, replace=False))].copy() if n else np.zeros((0, 4), np.float32)
        R = hp.rot("y", 0.25 * (b + 1))
        inputs.append((xyzw, R))
        scenes.upload(b, xyzw, None, R)
    po, pg = _proj(oracle)
    ops.scene_clip_batch(hip_ctx, pg, I4, scenes)
    hip_ctx.synchronize()
    for b, (xyzw, R) in enumerate(inputs):
        if len(xyzw) == 0:
            assert int(scenes.status[b].item()) == 1
            continue
        ref = oracle.scene_clip(po, R, I4, xyzw)
        _same(ref, scenes.clipped_of(b))


def test_error_contract(hip_ctx):
    # missing projector / clipped scene / global scene throw (scene_clipper_projective_3d.cpp:12-20)
    L = _lib.load()
    import ctypes as C
    n = C.c_int32(0)
    xyzw = np.zeros((4, 4), np.float32)
    idx = np.zeros(4, np.int32)
    pg = _lib.Projector(1, 1, 0, 0, 10, 10, 0.1, 10)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    assert L.prs_scene_clip(hip_ctx._h, None, p(I4), p(I4), p(xyzw), None, 4, p(xyzw), None, p(idx), 4, C.byref(n)) == -1
    assert b"projector" in L.prs_last_error(hip_ctx._h)
    assert L.prs_scene_clip(hip_ctx._h, C.byref(pg), p(I4), p(I4), p(xyzw), None, 4, None, None, p(idx), 4, C.byref(n)) == -1
    assert b"clipped scene" in L.prs_last_error(hip_ctx._h)
    assert L.prs_scene_clip(hip_ctx._h, C.byref(pg), p(I4), p(I4), None, None, 4, p(xyzw), None, p(idx), 4, C.byref(n)) == -1
    assert b"global scene" in L.prs_last_error(hip_ctx._h)
    assert L.prs_scene_clip(hip_ctx._h, C.byref(pg), p(I4), p(I4), p(xyzw), None, 0, p(xyzw), None, p(idx), 4, C.byref(n)) == 1
    assert L.prs_scene_clip(hip_ctx._h, C.byref(pg), p(I4), p(I4), p(xyzw), None, 4, p(xyzw), None, p(idx), 2, C.byref(n)) == -2
