#!/usr/bin/env python3
"""Turns the DATA files the reference's own tests read (/root/reference/test_data, loaded by
srrg2_proslam/tests/fixtures.hpp:555-1152) into the committed fixtures tests/golden/ref_*.npz.

Run in the build container only (the reference tree does not exist on the GPU box); the tests read the
.npz files, never /root/reference.  What is stored is data: decoded pixels, depth, ground-truth poses and
the sparse disparity ground truth -- no source text of the reference.

    python tools/make_ref_fixtures.py [/root/reference]

Decoding follows what the reference's fixtures do with OpenCV:
  * KITTI PNGs are 8-bit grayscale already: cv::imread(.., CV_LOAD_IMAGE_GRAYSCALE) returns the bytes
    (fixtures.hpp:1063-1066).
  * ICL / SceneFlow PNGs are RGB: cv::imread(.., GRAYSCALE) lets libpng convert (png_set_rgb_to_gray with
    0.299 / 0.587, i.e. the 15-bit coefficients 9797 / 19234 / 3737, truncating; a pixel with r == g == b keeps
    its value).  tests/test_ref_pins.py shows this choice reproduces the feature counts the reference pins
    on the ICL images (test_feature_extractors.cpp:111-135), which PIL's or cvtColor's weights do not.
  * ICL depth: 16-bit PGM, millimetres (fixtures.hpp:730-740 converts with 1e-3).
"""
import os
import sys

import numpy as np
from PIL import Image

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
DATA = os.path.join(REF, "test_data")
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def gray_u8(path):
    im = Image.open(path)
    if im.mode == "L":
        return np.array(im)
    assert im.mode == "RGB", im.mode
    rgb = np.array(im).astype(np.int64)
    r, g, b = rgb[..., 0], rgb[..., 1], rgb[..., 2]
    y = (r * 9797 + g * 19234 + b * 3737) >> 15
    return np.where((r == g) & (g == b), r, y).astype(np.uint8)


def depth_u16(path):
    d = np.array(Image.open(path))
    assert d.min() >= 0 and d.max() < 65536
    return d.astype(np.uint16)


def poses(path):
    p = np.loadtxt(path)
    assert p.shape[1] == 12
    return p.astype(np.float64)


def scene_flow_gt(path):
    """rows of `row col_left row col_right disparity` (fixtures.hpp SceneFlow::SetUp)"""
    return np.loadtxt(path).astype(np.float32)


def main():
    k = os.path.join(DATA, "kitti")
    np.savez_compressed(
        os.path.join(OUT, "ref_kitti.npz"),
        city_left=np.stack([gray_u8(os.path.join(k, "city", "image_left_%d.png" % i)) for i in range(5)]),
        city_right=np.stack([gray_u8(os.path.join(k, "city", "image_right_%d.png" % i)) for i in range(5)]),
        highway_left=np.stack([gray_u8(os.path.join(k, "highway", "image_left_%d.png" % i)) for i in (274, 275)]),
        highway_right=np.stack([gray_u8(os.path.join(k, "highway", "image_right_%d.png" % i)) for i in (274, 275)]),
        highway_first_index=274)
    # KITTI 00 / 01 ground-truth trajectories (12 floats per line, row-major 3x4); float32 holds the 7 printed digits
    np.savez_compressed(os.path.join(OUT, "ref_kitti_gt.npz"),
                        city=poses(os.path.join(k, "city", "gt.txt")).astype(np.float32),
                        highway=poses(os.path.join(k, "highway", "gt.txt")).astype(np.float32),
                        city_head_f64=poses(os.path.join(k, "city", "gt.txt"))[:8],
                        highway_274_f64=poses(os.path.join(k, "highway", "gt.txt"))[274:276])
    i = os.path.join(DATA, "icl")
    np.savez_compressed(os.path.join(OUT, "ref_icl.npz"),
                        gray=np.stack([gray_u8(os.path.join(i, "image_rgb_%d.png" % n)) for n in (0, 1, 50)]),
                        depth_mm=np.stack([depth_u16(os.path.join(i, "image_depth_%d.pgm" % n)) for n in (0, 1, 50)]),
                        frame_numbers=np.array([0, 1, 50]))
    s = os.path.join(DATA, "scene_flow")
    np.savez_compressed(os.path.join(OUT, "ref_scene_flow.npz"),
                        left=gray_u8(os.path.join(s, "image_left.png")), right=gray_u8(os.path.join(s, "image_right.png")),
                        gt_threshold_10=scene_flow_gt(os.path.join(s, "gt_stereo_matching_threshold-10.txt")),
                        gt_threshold_100=scene_flow_gt(os.path.join(s, "gt_stereo_matching_threshold-100.txt")))
    for f in sorted(os.listdir(OUT)):
        if f.startswith("ref_"):
            print(f, os.path.getsize(os.path.join(OUT, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
