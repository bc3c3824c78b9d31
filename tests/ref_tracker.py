"""The tracker loop of tests/test_trackers.cpp restated on the pieces of the hot path: adaptor -> (scene clipper ->
projective finder + aligner) -> merger, on the reference's own KITTI / ICL images (tests/golden/ref_*.npz).

`Tracker` is written against a small stage interface (`stages`) so that the CPU checker (tests/test_ref_tracker.py) and the
HIP path (tests/test_ref_tracker_gpu.py) run the same loop; the scenarios and the bounds the reference asserts live here.
robotInLocalMap of frame k = prediction * X^-1, X = local map in sensor as estimated by the aligner
(multi_tracker: the measurement is the fixed cloud, the clipped local map the moving one).  The prediction is the constant-velocity
one of the aligner's motion-model slice (kitti.conf:257-260,747-772; `use_prediction`, default since round 4: tools/sweep_a13.py)."""
import numpy as np

import ref_pins as rp

I4 = np.eye(4, dtype=np.float32)


def kitti_measurements(B, i):
    """KITTI fixture adaptor (fixtures.hpp:800-1050): FAST 15, 500 keypoints in 3x3 bins, epipolar matcher (50, Lowe 0.8)"""
    key = ("kitti_meas", B.name, i)
    if key not in rp._cache:
        pts, desc, _ = rp.stereo_adaptor(B, rp.kitti_image("left", i), rp.kitti_image("right", i), 15, "epipolar", 50.0, 0.8)
        rp._cache[key] = (pts, desc)
    return rp._cache[key]


def icl_measurements(B, k):
    m = rp.icl_measurements(B, k)
    return np.concatenate([m["uv"], m["depth"][:, None]], axis=1).astype(np.float32), m["desc"]


def kitti_setup(no_merges, kind="weighted_mean"):
    """KITTI 00To04_Tracker_* (tests/test_trackers.cpp:258-783): kitti.conf tracker, robustifier chi 1000, circle finder
    with distance 25..100, Lowe 0.5, radius 50..5, matching ratio 0.1.  `no_merges` (:260-360): appearance and geometry thresholds 0.
    kind (merger + landmark estimator the test plugs in):
      "weighted_mean"  :362-470  merger_triangulation, appearance 50, landmark_estimator_weighted_mean with 25 m^2
      "ekf"            :473-576  merger_ekf (MergerRigidStereoProjectiveEKF + landmark_estimator_ekf, kitti.conf:1-18,552-590), appearance 50, 25 m^2
      "smoother"       :578-682  merger_triangulation + landmark_estimator_smoother (kitti.conf:503-521), appearance 50, 25 m^2
      "bruteforce_ekf" :684-783  merger_ekf as configured (appearance 100, 25 m^2) and a descriptor-based brute-force finder
                                 (distance 100, Lowe 0.5) in the aligner slice instead of the projective one"""
    from srrg2_proslam_amd import configs
    cfg = dict(configs.get("kitti"))
    f = dict(cfg["projective_finder"])
    f.update(minimum_descriptor_distance=25.0, maximum_descriptor_distance=100.0, maximum_distance_ratio_to_second_best=0.5,
             minimum_search_radius_pixels=5, maximum_search_radius_pixels=50, minimum_matching_ratio=0.1, search_type=rp.CIRCLE)
    a = dict(cfg["aligner"])
    a["chi_threshold"] = 1000.0
    cfg["projective_finder"], cfg["aligner"] = f, a
    if kind == "bruteforce_ekf":
        cfg["bruteforce_finder"] = (100.0, 0.5)  # :722-726
    if no_merges:
        merger = dict(variant="stereo_triangulation", estimator="weighted_mean", max_appearance=0.0, max_dist2=0.0)
    elif kind == "weighted_mean":
        merger = dict(variant="stereo_triangulation", estimator="weighted_mean", max_appearance=50.0, max_dist2=25.0)
    elif kind == "smoother":
        merger = dict(variant="stereo_triangulation", estimator="smoother", max_appearance=50.0, max_dist2=25.0, max_measurements=8)
    elif kind in ("ekf", "bruteforce_ekf"):
        merger = dict(variant="stereo_ekf", estimator="ekf4", max_appearance=50.0 if kind == "ekf" else 100.0, max_dist2=25.0)
    else:
        raise ValueError(kind)
    return cfg, merger


def icl_setup():
    """ICL 00To50_Tracker_ProjectiveBruteforce (tests/test_trackers.cpp:90-162): icl.conf tracker + cf_projective_circle"""
    from srrg2_proslam_amd import configs
    cfg = dict(configs.get("icl"))
    f = dict(cfg["projective_finder"])
    f["search_type"] = rp.CIRCLE
    cfg["projective_finder"] = f
    return cfg, dict(variant="depth_ekf", estimator="ekf3", max_appearance=50.0, max_dist2=1.0)


class Tracker:
    """stages: object with
         clip(cfg, pose, map) -> (xyzw, desc, scene indices)
         align(cfg, fixed, fixed_desc, xyzw, desc, guess, prior_info) -> (X, corr (fixed = measurement, moving = clipped), status, inliers)
               (cfg["bruteforce_finder"] = (distance, ratio): the slice's finder is the descriptor-based brute-force one, whose
                correspondences do not depend on the estimate: max_iterations x (linearize, GN step) on one vector)
         new_map(cfg, merger, capacity) -> map;  merge(map, pose, fixed, fixed_desc, corr, scene_indices) -> (n_merged, n_added)
         map_size(map), predict(prev, pose) -> pose, compose(guess, X) -> guess * X^-1, reset()"""

    def __init__(self, stages, cfg, merger, capacity=4096, use_prediction=True, prior_info=0.0):
        self.s, self.cfg, self.prior_info, self.use_prediction = stages, cfg, prior_info, use_prediction
        stages.reset()  # a new tracker: new finder state
        self.map = stages.new_map(cfg, merger, capacity)
        self.pose, self.prev = I4.copy(), I4.copy()
        self.frames = 0
        self.log = []

    def process(self, fixed, desc):
        s = self.s
        entry = dict(n_measured=len(fixed))
        corr, idx = None, None
        if self.frames > 0:
            if self.use_prediction == "estimate":
                # the scene is clipped at the last pose and the aligner's estimate starts at (last pose)^-1 * prediction, inverted
                guess = self.pose
                pred = s.predict(self.prev, self.pose)
                X0 = np.linalg.inv(np.linalg.inv(np.asarray(self.pose, np.float64)) @ np.asarray(pred, np.float64)).astype(np.float32)
            else:
                guess = s.predict(self.prev, self.pose) if self.use_prediction else self.pose
                X0 = I4
            xyzw, cdesc, idx = s.clip(self.cfg, guess, self.map)
            X, corr, status, inliers = s.align(self.cfg, fixed, desc, xyzw, cdesc, X0, self.prior_info)
            self.prev = self.pose
            self.pose = s.compose(guess, X)
            entry.update(status=status, inliers=inliers, n_corr=len(corr), n_clipped=len(xyzw))
        merged, added = s.merge(self.map, self.pose, fixed, desc, corr, idx)
        self.frames += 1
        entry.update(pose=self.pose.copy(), map_size=s.map_size(self.map), merged=merged, added=added)
        self.log.append(entry)
        return entry


def same_frame_three_times(stages, B, dataset):
    """..00To00_Tracker_.._MergerDefault (tests/test_trackers.cpp:7-88, :164-256): the robot does not move, the map does not grow"""
    if dataset == "kitti":
        cfg, merger = kitti_setup(no_merges=False)
        fixed, desc = kitti_measurements(B, 0)
    else:
        cfg, merger = icl_setup()
        fixed, desc = icl_measurements(B, 0)
    t = Tracker(stages, cfg, merger)
    return [t.process(fixed, desc) for _ in range(3)]


def kitti_00_to_04(stages, B, no_merges, kind="weighted_mean", **kw):
    cfg, merger = kitti_setup(no_merges, kind)
    t = Tracker(stages, cfg, merger, **kw)
    log = [t.process(*kitti_measurements(B, i)) for i in range(5)]
    error = rp.t2tnq(np.linalg.inv(np.asarray(t.pose, np.float64)) @ rp.kitti_relative(4, 0))
    return log, error


def icl_00_01_50(stages, B, **kw):
    cfg, merger = icl_setup()
    t = Tracker(stages, cfg, merger, **kw)
    log = [t.process(*icl_measurements(B, k)) for k in (0, 1, 50)]
    error = rp.t2tnq(np.linalg.inv(np.asarray(t.pose, np.float64)) @ rp.icl_relative(50, 0))
    return log, error
