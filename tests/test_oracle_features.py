"""Behaviour pins for SURVEY.md 8f row 3 (feature extraction); the reference's own count gates
(tests/test_feature_extractors.cpp:7-262) on its own images are in tests/test_ref_pins.py.  Here: FAST finds rectangle
corners and nothing on edges or flat areas,
responses grow with contrast, the in-repo region binning (intensity_feature_extractor_binned.cpp:47-196), translation
covariance of the descriptors, and that a synthetic stereo pair extracts into clouds the epipolar matcher associates
at the painted disparities."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest

from oracle import binding as ob
from oracle import binding_features as of
from srrg2_proslam_amd import configs, synthetic as syn
from tests import helpers as hp


def _rect_image(contrast=100, rows=120, cols=160):
    img = np.full((rows, cols), 60, np.uint8)
    img[40:80, 50:110] = 60 + contrast
    return img


def test_F1_fast_fires_on_corners_only():
    s = of.fast_scores(_rect_image(), 15)
    rr, cc = np.nonzero(s)
    assert len(rr) > 0
    corners = np.array([(40, 50), (40, 109), (79, 50), (79, 109)])
    # every response sits within 3 px of a rectangle corner; edges and flat regions are silent
    dist = np.min(np.abs(rr[:, None] - corners[:, 0]) + np.abs(cc[:, None] - corners[:, 1]), axis=1)
    assert dist.max() <= 4
    for r0, c0 in corners:
        assert s[max(r0 - 3, 0): r0 + 4, max(c0 - 3, 0): c0 + 4].max() > 0
    assert s[:3].max() == 0 and s[:, :3].max() == 0  # the outermost three pixels are not examined


def test_F2_response_grows_with_contrast_and_respects_the_threshold():
    lo, hi = of.fast_scores(_rect_image(40), 15), of.fast_scores(_rect_image(120), 15)
    assert hi.max() > lo.max() > 15
    assert of.fast_scores(_rect_image(40), 60).max() == 0  # contrast below the threshold
    # response = largest threshold that still detects: re-running at that threshold keeps the pixel, one above drops it
    r, c = np.unravel_index(np.argmax(lo), lo.shape)
    assert of.fast_scores(_rect_image(40), int(lo[r, c]))[r, c] == lo[r, c]
    assert of.fast_scores(_rect_image(40), int(lo[r, c]) + 1)[r, c] == 0


def test_F3_region_binning():
    rng = np.random.default_rng(1)
    cfg = configs.get("kitti")
    left, _, _ = syn.stereo_images(rng, cfg)
    p = of.extractor_params(target=450)  # 50 per region
    uv, inten, desc = of.extract_features(p, left)
    rows, cols = left.shape
    reg = (np.floor(uv[:, 1] / np.float32(rows / 3)) * 3 + (uv[:, 0] / np.float32(cols / 3)).astype(np.int64)).astype(np.int64)
    assert np.all(np.diff(reg) >= 0)  # regions in order (intensity_feature_extractor_binned.cpp:171-196)
    assert np.bincount(reg, minlength=9).max() <= 50 and len(uv) > 300
    assert np.all(inten == left[uv[:, 1].astype(int), uv[:, 0].astype(int)])
    # nothing within the descriptor border
    assert uv[:, 0].min() >= of.FEATURE_BORDER and uv[:, 0].max() < cols - of.FEATURE_BORDER
    # with a huge target every region keeps all its keypoints in raster order
    uv_all, _, _ = of.extract_features(of.extractor_params(target=10 ** 6), left, capacity=60000)
    reg_all = (np.floor(uv_all[:, 1] / np.float32(rows / 3)) * 3 + (uv_all[:, 0] / np.float32(cols / 3)).astype(np.int64)).astype(np.int64)
    for g in range(9):
        sel = uv_all[reg_all == g]
        key = sel[:, 1].astype(np.int64) * cols + sel[:, 0].astype(np.int64)
        assert np.all(np.diff(key) > 0)


def test_F4_descriptors_follow_a_translation():
    rng = np.random.default_rng(2)
    cfg = configs.get("kitti")
    left, _, _ = syn.stereo_images(rng, cfg)
    shifted = np.roll(left, (5, 9), axis=(0, 1))
    p = of.extractor_params(target=10 ** 6)
    uv0, _, d0 = of.extract_features(p, left, capacity=60000)
    uv1, _, d1 = of.extract_features(p, shifted, capacity=60000)
    lut = {(int(u), int(v)): i for i, (u, v) in enumerate(uv1)}
    hits = 0
    for i, (u, v) in enumerate(uv0):
        if 40 < u < left.shape[1] - 50 and 40 < v < left.shape[0] - 50:
            j = lut.get((int(u) + 9, int(v) + 5))
            assert j is not None
            assert np.array_equal(d0[i], d1[j])
            hits += 1
    assert hits > 200
    pat = of.orb_pattern()
    assert pat.min() >= -13 and pat.max() <= 13 and len({tuple(r) for r in pat}) == 256


def test_F5_stereo_pair_extracts_into_matchable_clouds():
    rng = np.random.default_rng(3)
    cfg = configs.get("kitti")
    left, right, rects = syn.stereo_images(rng, cfg)
    p = of.extractor_params()
    uvl, _, dl = of.extract_features(p, left)
    uvr, _, dr = of.extract_features(p, right)
    assert 600 < len(uvl) <= 1000 and 600 < len(uvr) <= 1000
    corr, flags = ob.stereo_match(uvl, dl, uvr, dr, hp.oracle_stereo_params(ob, cfg["stereo_matcher"]))
    assert len(corr) > 0.3 * len(uvl)
    disp = uvl[corr["fixed_idx"], 0] - uvr[corr["moving_idx"], 0]
    painted = {2} | {r[4] for r in rects}
    good = np.isin(disp.astype(int), list(painted))
    assert good.mean() > 0.9 and np.median(corr["response"]) <= 16  # random pairs sit at 128 +- 8


def test_F6_gaussian_blur_fixed_point():
    """cv::GaussianBlur(7x7, sigma 2) on 8-bit data: kernel round(256 g) = 18 34 49 55 49 34 18 (sum 257), reflect-101
    borders, (sum + 2^15) >> 16 -- against a direct numpy evaluation"""
    rng = np.random.default_rng(4)
    img = rng.integers(0, 256, (37, 53), dtype=np.uint8)
    k = np.array([18, 34, 49, 55, 49, 34, 18], np.int64)
    pad = np.pad(img.astype(np.int64), 3, mode="reflect")
    h = sum(k[i] * pad[:, i:i + 53] for i in range(7))
    v = sum(k[i] * h[i:i + 37, :] for i in range(7))
    want = np.minimum((v + (1 << 15)) >> 16, 255).astype(np.uint8)
    assert np.array_equal(of.gaussian_blur7(img), want)
    assert of.gaussian_blur7(np.full((20, 20), 255, np.uint8)).min() == 255  # the 257/256 gain saturates
    assert np.array_equal(of.gaussian_blur7(np.full((9, 9), 100, np.uint8)), np.full((9, 9), 101, np.uint8))


_STD_SORT_REF = r"""
#include <algorithm>
#include <vector>
struct KP { float x, y, size, angle, response; int octave, class_id; int order; };  // cv::KeyPoint-sized element
extern "C" void sort_desc(int n, const int* resp, int* order_out) {
  std::vector<KP> v(n);
  for (int i = 0; i < n; ++i) { v[i].response = (float) resp[i]; v[i].order = i; }
  std::sort(v.begin(), v.end(), [](const KP& a, const KP& b) { return a.response > b.response; });
  for (int i = 0; i < n; ++i) order_out[i] = v[i].order;
}
"""


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++ (GNU libstdc++) to compile the std::sort reference")
def test_F7_std_sort_restatement_equals_libstdcxx(tmp_path):
    """orc_std_sort_desc must leave exactly the permutation GNU std::sort leaves (the comparator of
    intensity_feature_extractor_binned.cpp:182-186 only looks at the response): ties, sorted / reversed runs and
    inputs that exhaust the depth limit (heapsort fallback)"""
    src = tmp_path / "std_sort_ref.cpp"
    src.write_text(_STD_SORT_REF)
    so = tmp_path / "libstd_sort_ref.so"
    subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-o", str(so), str(src)])
    L = C.CDLL(str(so))

    def ref(x):
        r = np.ascontiguousarray(x, np.int32)
        o = np.zeros(len(r), np.int32)
        L.sort_desc(len(r), r.ctypes.data_as(C.c_void_p), o.ctypes.data_as(C.c_void_p))
        return o

    rng = np.random.default_rng(5)
    cases = []
    for n in list(range(0, 40)) + [100, 257, 1000, 1650, 6000]:
        for vals in (2, 5, 30, 250, 10 ** 6):
            cases.append(rng.integers(0, vals, n))
        cases += [np.arange(n), np.arange(n)[::-1].copy(), np.zeros(n, int)]
    for n in (200, 1000, 5000):  # structured inputs, many of which reach the heapsort fallback
        cases.append(np.concatenate([np.arange(0, n, 2), np.arange(1, n, 2)]))
        for m in (3, 17, 101):
            cases.append((np.arange(n) * 7919) % m)
        cases.append(np.abs(np.arange(n) - n // 2) + rng.integers(0, 2, n))
        cases.append(-np.abs(np.arange(n) - n // 2) + rng.integers(0, 2, n))
    for x in cases:
        got = of.std_sort_desc(x)
        assert np.array_equal(got, ref(x))
        assert np.all(np.diff(np.asarray(x)[got]) <= 0)
