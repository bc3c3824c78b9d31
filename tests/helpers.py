"""shared helpers of the test-suite (inputs, comparisons)"""
import numpy as np

from srrg2_proslam_amd import configs, synthetic as syn


def kitti_frame(seed, n=2000, **kw):
    cfg = configs.get("kitti")
    rng = np.random.default_rng(seed)
    return cfg, syn.stereo_frame(rng, cfg, n, **kw)


def corr_equal(a, b):
    """bit-exact equality of two correspondence vectors INCLUDING order"""
    return (len(a) == len(b) and np.array_equal(a["fixed_idx"], b["fixed_idx"])
            and np.array_equal(a["moving_idx"], b["moving_idx"])
            and np.array_equal(a["response"].view(np.uint32), b["response"].view(np.uint32)))


def corr_set(c):
    return set(zip(c["fixed_idx"].tolist(), c["moving_idx"].tolist(), c["response"].tolist()))


def oracle_stereo_params(ob, m):
    return ob.StereoParams(m["maximum_descriptor_distance"], m["maximum_distance_ratio_to_second_best"],
                           m["minimum_matching_ratio"], m["maximum_disparity_pixels"],
                           m["epipolar_line_thickness_pixels"])


def oracle_tri_params(ob, cfg):
    cam, tri = cfg["camera"], cfg["triangulator"]
    return ob.TriangulatorParams(cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["fx"] * cam["baseline_m"],
                                 tri["minimum_disparity_pixels"], tri["infinity_depth_meters"])
