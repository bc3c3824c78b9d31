"""Seeded synthetic inputs shaped like the reference's datasets.

There is no network and no feature detector in the image, so the hot path is fed with
keypoints + 256-bit descriptors generated here:
  * landmarks are sampled in the left camera frustum, projected into the left and right
    image (rectified stereo, tests/fixtures.hpp:360-366), coordinates rounded to integer pixels
    (FAST yields integer keypoints; the matcher truncates anyway, epipolar_impl.cpp:10-11);
  * descriptors are i.i.d. Bernoulli(0.5) per landmark, each observation flips bits with
    p = 0.04 (true pairs ~10 bit apart, random pairs ~128);
  * every image is padded to N keypoints with uniformly random outliers;
  * the "moving" cloud (local map) holds the same landmarks expressed in the previous camera
    frame plus unrelated map points, with per-point optimisation counts in [0, 30].
All arrays are numpy; nothing here touches the GPU.
"""
import numpy as np

from . import configs

SEED_BASE = 20200300  # seed = 20200300 + 1000*config + sequence


def seed_for(config_index, sequence):
    return SEED_BASE + 1000 * int(config_index) + int(sequence)


def random_descriptors(rng, n):
    return rng.integers(0, 256, size=(n, 32), dtype=np.uint8)


def flip_bits(rng, desc, p):
    """flip every bit of desc [n,32] u8 independently with probability p"""
    if p <= 0:
        return desc.copy()
    n = desc.shape[0]
    flips = rng.random((n, 256)) < p
    mask = np.packbits(flips, axis=1)
    return np.bitwise_xor(desc, mask)


def rot_xyz(rx, ry, rz):
    cx, sx = np.cos(rx), np.sin(rx)
    cy, sy = np.cos(ry), np.sin(ry)
    cz, sz = np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rx @ Ry @ Rz


def make_transform(t, r):
    T = np.eye(4)
    T[:3, :3] = rot_xyz(*r)
    T[:3, 3] = t
    return T


def sample_landmarks(rng, cam, depth, n):
    """n points in the left camera frame whose left projection is inside the image"""
    u = rng.uniform(0.0, cam["cols"] - 1.0, n)
    v = rng.uniform(0.0, cam["rows"] - 1.0, n)
    z = rng.uniform(depth["min"], depth["max"], n)
    x = (u - cam["cx"]) * z / cam["fx"]
    y = (v - cam["cy"]) * z / cam["fy"]
    return np.stack([x, y, z], axis=1)


def project_left_right(cam, pts):
    """rectified stereo projection: right = (K p - (fx*b, 0, 0)) / z"""
    z = pts[:, 2]
    u = cam["fx"] * pts[:, 0] / z + cam["cx"]
    v = cam["fy"] * pts[:, 1] / z + cam["cy"]
    ur = u - cam["fx"] * cam["baseline_m"] / z
    return u, v, ur


def stereo_frame(rng, cfg, n_keypoints=2000, visible_fraction=0.6, flip_p=0.04, row_jitter_fraction=0.0,
                 landmarks=None, landmark_desc=None):
    """one KITTI-/EuRoC-shaped stereo pair.

    returns dict with uv_left/right [N,2] f32, desc_left/right [N,32] u8 (keypoint order is a random
    permutation) and ground truth: lm_of_left / lm_of_right (landmark id or -1 per keypoint),
    landmarks [L,3], landmark_desc [L,32].
    """
    cam, depth = cfg["camera"], cfg["depth"]
    n_lm = int(round(n_keypoints * visible_fraction))
    if landmarks is None:
        landmarks = sample_landmarks(rng, cam, depth, n_lm)
        landmark_desc = random_descriptors(rng, n_lm)
    n_lm = landmarks.shape[0]
    u, v, ur = project_left_right(cam, landmarks)
    ul, vl, urr = np.rint(u), np.rint(v), np.rint(ur)
    vr = vl.copy()
    if row_jitter_fraction > 0:
        jit = rng.random(n_lm) < row_jitter_fraction
        vr = vr + jit * rng.choice([-1.0, 1.0], n_lm)
    in_left = (ul >= 0) & (ul < cam["cols"]) & (vl >= 0) & (vl < cam["rows"]) & (landmarks[:, 2] > 0)
    in_right = in_left & (urr >= 0) & (urr < cam["cols"]) & (vr >= 0) & (vr < cam["rows"])

    def build(mask, uu, vv):
        ids = np.nonzero(mask)[0]
        if ids.shape[0] > n_keypoints:
            ids = ids[:n_keypoints]
        n_out = n_keypoints - ids.shape[0]
        uv = np.empty((n_keypoints, 2), dtype=np.float32)
        uv[: ids.shape[0], 0] = uu[ids]
        uv[: ids.shape[0], 1] = vv[ids]
        uv[ids.shape[0]:, 0] = rng.integers(0, cam["cols"], n_out)
        uv[ids.shape[0]:, 1] = rng.integers(0, cam["rows"], n_out)
        desc = np.empty((n_keypoints, 32), dtype=np.uint8)
        desc[: ids.shape[0]] = flip_bits(rng, landmark_desc[ids], flip_p)
        desc[ids.shape[0]:] = random_descriptors(rng, n_out)
        lm = np.full(n_keypoints, -1, dtype=np.int64)
        lm[: ids.shape[0]] = ids
        perm = rng.permutation(n_keypoints)
        return uv[perm], desc[perm], lm[perm]

    uv_l, d_l, lm_l = build(in_left, ul, vl)
    uv_r, d_r, lm_r = build(in_right, urr, vr)
    return {"uv_left": uv_l, "desc_left": d_l, "uv_right": uv_r, "desc_right": d_r,
            "lm_of_left": lm_l, "lm_of_right": lm_r, "landmarks": landmarks, "landmark_desc": landmark_desc}


def default_motion(rng, cfg):
    """inter-frame motion local_map(previous camera) -> current camera, KITTI-like: the car drives
    ~0.86 m forward per frame (test_data/kitti/city/gt.txt rows 1-3) with a small yaw"""
    if cfg["name"] == "kitti":
        t = np.array([rng.normal(0, 0.02), rng.normal(0, 0.01), -0.86 + rng.normal(0, 0.08)])
        r = np.array([rng.normal(0, 0.002), rng.normal(0, 0.01), rng.normal(0, 0.002)])
    elif cfg["name"] == "euroc":
        t = rng.normal(0, 0.03, 3)
        r = rng.normal(0, 0.01, 3)
    else:
        t = rng.normal(0, 0.015, 3)
        r = rng.normal(0, 0.008, 3)
    return make_transform(t, r)


def local_map(rng, cfg, frame, T_map_to_sensor, n_moving=2000, tracked_fraction=0.7, flip_p=0.04,
              position_noise=0.01):
    """moving cloud for the projective finder / aligner: landmarks of `frame` expressed in the
    map (previous camera) frame + unrelated map points; n_opt in [0,30]"""
    lm = frame["landmarks"]
    n_lm = lm.shape[0]
    n_tracked = min(int(round(n_lm * tracked_fraction)), n_moving)
    ids = rng.permutation(n_lm)[:n_tracked]
    T_inv = np.linalg.inv(T_map_to_sensor)
    pts = lm[ids] @ T_inv[:3, :3].T + T_inv[:3, 3]
    pts = pts + rng.normal(0, position_noise, pts.shape) * (lm[ids, 2:3] / 10.0)
    n_other = n_moving - n_tracked
    other = sample_landmarks(rng, cfg["camera"], cfg["depth"], n_other)
    xyz = np.concatenate([pts, other], axis=0).astype(np.float32)
    desc = np.concatenate([flip_bits(rng, frame["landmark_desc"][ids], flip_p),
                           random_descriptors(rng, n_other)], axis=0)
    lm_of_moving = np.concatenate([ids, np.full(n_other, -1, dtype=np.int64)])
    perm = rng.permutation(n_moving)
    n_opt = rng.integers(0, 31, n_moving).astype(np.uint32)
    return {"xyz": xyz[perm], "desc": desc[perm], "lm_of_moving": lm_of_moving[perm], "n_opt": n_opt}


def rgbd_frame(rng, cfg, n_keypoints=1000, visible_fraction=0.7, flip_p=0.04):
    """TUM/ICL-shaped fixed cloud: (u, v, d) + descriptor (raw_data_preprocessor_monocular_depth.cpp:174)"""
    cam, depth = cfg["camera"], cfg["depth"]
    n_lm = int(round(n_keypoints * visible_fraction))
    landmarks = sample_landmarks(rng, cam, depth, n_lm)
    landmark_desc = random_descriptors(rng, n_lm)
    z = landmarks[:, 2]
    u = np.rint(cam["fx"] * landmarks[:, 0] / z + cam["cx"])
    v = np.rint(cam["fy"] * landmarks[:, 1] / z + cam["cy"])
    ok = (u >= 0) & (u < cam["cols"]) & (v >= 0) & (v < cam["rows"])
    ids = np.nonzero(ok)[0]
    n_out = n_keypoints - ids.shape[0]
    uvd = np.empty((n_keypoints, 3), dtype=np.float32)
    uvd[: ids.shape[0], 0] = u[ids]
    uvd[: ids.shape[0], 1] = v[ids]
    uvd[: ids.shape[0], 2] = z[ids] + rng.normal(0, 0.005, ids.shape[0])
    uvd[ids.shape[0]:, 0] = rng.integers(0, cam["cols"], n_out)
    uvd[ids.shape[0]:, 1] = rng.integers(0, cam["rows"], n_out)
    uvd[ids.shape[0]:, 2] = rng.uniform(depth["min"], depth["max"], n_out)
    desc = np.empty((n_keypoints, 32), dtype=np.uint8)
    desc[: ids.shape[0]] = flip_bits(rng, landmark_desc[ids], flip_p)
    desc[ids.shape[0]:] = random_descriptors(rng, n_out)
    lm = np.full(n_keypoints, -1, dtype=np.int64)
    lm[: ids.shape[0]] = ids
    perm = rng.permutation(n_keypoints)
    return {"fixed": uvd[perm], "desc_fixed": desc[perm], "lm_of_fixed": lm[perm],
            "landmarks": landmarks, "landmark_desc": landmark_desc}


def perturb(rng, T, sigma_t, sigma_r):
    """initial guess = truth composed with a small error (stand-in for the constant-velocity prediction)"""
    return (make_transform(rng.normal(0, sigma_t, 3), rng.normal(0, sigma_r, 3)) @ T).astype(np.float32)


def stereo_images(rng, cfg, n_rectangles=140, max_disparity=90):
    """a rectified stereo pair of 8-bit images: a blocky random background at disparity 2 and large rectangles with
    their own 8-px block texture at integer disparities, painted far to near.  Corners inside a surface see the same
    neighbourhood in both images (their descriptors agree bit for bit); corners at depth discontinuities do not.
    Returns (left, right, rectangles) with rectangles = [(x, y, w, h, disparity)] in painting order."""
    cam = cfg["camera"]
    rows, cols = int(cam["rows"]), int(cam["cols"])

    def blocks(h, w, size):
        b = rng.integers(20, 236, ((h + size - 1) // size, (w + size - 1) // size)).astype(np.uint8)
        t = np.kron(b, np.ones((size, size), np.uint8))[:h, :w].copy()
        # small high-contrast squares: the corner-like structures a FAST detector answers to
        n_blobs = max(h * w // 220, 1)
        ys, xs = rng.integers(0, max(h - 4, 1), n_blobs), rng.integers(0, max(w - 4, 1), n_blobs)
        gs = rng.integers(0, 2, n_blobs) * 255
        for y, x, g in zip(ys, xs, gs):
            t[y: y + 4, x: x + 4] = g
        return t

    world = blocks(rows, cols + 4, 16)
    left = np.ascontiguousarray(world[:, 2: cols + 2])   # left[x] = world[x + 2]
    right = np.ascontiguousarray(world[:, 4: cols + 4])  # right[x] = world[x + 4] = left[x + 2]: uL - uR = 2
    d = np.sort(rng.integers(3, max_disparity, n_rectangles))
    rects = []
    for k in range(n_rectangles):
        w, h = int(rng.integers(60, 220)), int(rng.integers(40, 140))
        x, y = int(rng.integers(0, cols - w)), int(rng.integers(0, rows - h))
        tex = blocks(h, w, 8)
        left[y: y + h, x: x + w] = tex
        xr = x - int(d[k])
        x0, x1 = max(xr, 0), min(xr + w, cols)
        if x1 > x0:
            right[y: y + h, x0: x1] = tex[:, x0 - xr: x1 - xr]
        rects.append((x, y, w, h, int(d[k])))
    return left, right, rects


def stereo_image_sequence(rng, cfg, n_frames, n_rectangles=140, max_disparity=88):
    """rectified stereo image pairs of ONE static layered scene seen from a camera that steps sideways by a quarter of
    the baseline per frame: every disparity is a multiple of 4, so a layer moves by disparity / 4 whole pixels from
    frame to frame and patches stay identical pixel for pixel.  Returns ([(left, right)] per frame, step in metres):
    camera k sits at x = k * step in the frame of camera 0."""
    cam = cfg["camera"]
    rows, cols = int(cam["rows"]), int(cam["cols"])
    margin = (max_disparity // 4) * n_frames + max_disparity + 8  # content that scrolls into view

    def blocks(h, w, size):
        b = rng.integers(20, 236, ((h + size - 1) // size, (w + size - 1) // size)).astype(np.uint8)
        t = np.kron(b, np.ones((size, size), np.uint8))[:h, :w].copy()
        n_blobs = max(h * w // 220, 1)
        ys, xs = rng.integers(0, max(h - 4, 1), n_blobs), rng.integers(0, max(w - 4, 1), n_blobs)
        gs = rng.integers(0, 2, n_blobs) * 255
        for y, x, g in zip(ys, xs, gs):
            t[y: y + 4, x: x + 4] = g
        return t

    wide = cols + 2 * margin
    background = blocks(rows, wide, 16)  # disparity 4: one pixel per frame
    d = np.sort(rng.integers(2, max_disparity // 4 + 1, n_rectangles)) * 4
    rects = []
    for k in range(n_rectangles):
        w, h = int(rng.integers(60, 220)), int(rng.integers(40, 140))
        x, y = int(rng.integers(0, wide - w)), int(rng.integers(0, rows - h))
        rects.append((x, y, blocks(h, w, 8), int(d[k])))

    def render(shift_of):
        """image whose layer with disparity dd is displaced by shift_of(dd) pixels to the left"""
        img = np.empty((rows, cols), np.uint8)
        s0 = shift_of(4)
        img[:] = background[:, margin + s0: margin + s0 + cols]
        for x, y, tex, dd in rects:  # far to near
            xs = x - margin - shift_of(dd)
            x0, x1 = max(xs, 0), min(xs + tex.shape[1], cols)
            if x1 > x0:
                img[y: y + tex.shape[0], x0: x1] = tex[:, x0 - xs: x1 - xs]
        return img

    frames = []
    for k in range(n_frames):
        left = render(lambda dd: k * dd // 4)
        right = render(lambda dd: k * dd // 4 + dd)
        frames.append((left, right))
    return frames, float(cam["baseline_m"]) / 4.0
