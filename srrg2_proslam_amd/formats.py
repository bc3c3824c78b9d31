"""Harness formats of the reference's benchmark tooling: trajectory files and a reader for
the subset of the BOSS `.conf` grammar the hot-path parameters live in.  Host-side plumbing, plain Python.

Trajectories (apps/app_benchmark.cpp:195-262):
  * a run keeps, per local map (keyframe), the stamped poses of the frames tracked in it, LOCAL to that map;
    `unroll_trajectory` composes `keyframe_estimate * local_pose` and orders by timestamp (`std::map::insert`:
    the first pose seen for a timestamp wins) (:195-203);
  * KITTI: one line per pose, the 3x4 matrix row-major, every value `std::fixed << std::setprecision(9)`
    followed by ONE space (also the last one), then a newline (:205-222, :244-250);
  * TUM: `timestamp tx ty tz qx qy qz qw ` in the same number format (:224-241, :252-259).  `t2tqxyzw` is part
    of srrg2_core (not in the reference tree): BUILD-DEFINED as translation + the unit quaternion of the rotation
    block by the usual trace / largest-diagonal branches, (x, y, z, w) order as the name says.
Values are float32 in the reference (`Isometry3f`); they are widened to double for printing, exactly what
`ostream << float` does, so a file written here is byte-identical to one written by the reference for the same
poses.

`.conf` files (configurations/*.conf): a sequence of `"ClassName" { ... }` records; a record body is a JSON
object with `//` comments, `"#id"` numbers the record, `{"#pointer": id}` refers to another record (-1 = none).
"""
import json
import math

import numpy as np


# ---------------------------------------------------------------- trajectories
def unroll_trajectory(local_maps):
    """local_maps: iterable of (keyframe_estimate [4,4], [(timestamp, local_pose [4,4]), ...]) ->
    list of (timestamp, global_pose float32 [4,4]) ordered by timestamp (app_benchmark.cpp:195-203)."""
    unrolled = {}
    for keyframe, stamped in local_maps:
        K = np.asarray(keyframe, np.float32).reshape(4, 4)
        for stamp, local in stamped:
            stamp = float(stamp)
            if stamp not in unrolled:  # std::map::insert keeps the entry already there
                unrolled[stamp] = (K @ np.asarray(local, np.float32).reshape(4, 4)).astype(np.float32)
    return [(s, unrolled[s]) for s in sorted(unrolled)]


def _fixed9(x):
    return "%.9f" % float(x)


def rotation_to_quaternion_xyzw(R):
    """unit quaternion (x, y, z, w) of a rotation matrix, float32 arithmetic (trace / largest-diagonal branches)."""
    m = np.asarray(R, np.float32).reshape(3, 3)
    f = np.float32
    t = m[0, 0] + m[1, 1] + m[2, 2]
    if t > f(0):
        t = np.sqrt(t + f(1))
        w = f(0.5) * t
        t = f(0.5) / t
        x, y, z = (m[2, 1] - m[1, 2]) * t, (m[0, 2] - m[2, 0]) * t, (m[1, 0] - m[0, 1]) * t
    else:
        i = 0
        if m[1, 1] > m[0, 0]:
            i = 1
        if m[2, 2] > m[i, i]:
            i = 2
        j, k = (i + 1) % 3, (i + 2) % 3
        t = np.sqrt(m[i, i] - m[j, j] - m[k, k] + f(1))
        q = [f(0), f(0), f(0)]
        q[i] = f(0.5) * t
        t = f(0.5) / t
        w = (m[k, j] - m[j, k]) * t
        q[j] = (m[j, i] + m[i, j]) * t
        q[k] = (m[k, i] + m[i, k]) * t
        x, y, z = q
    return np.array([x, y, z, w], np.float32)


def kitti_line(pose):
    """app_benchmark.cpp:244-250: twelve values, each followed by a space."""
    T = np.asarray(pose, np.float32).reshape(4, 4)
    return "".join(_fixed9(T[r, c]) + " " for r in range(3) for c in range(4))


def tum_line(stamp, pose):
    """app_benchmark.cpp:252-259: timestamp (double), translation, quaternion xyzw, each followed by a space."""
    T = np.asarray(pose, np.float32).reshape(4, 4)
    v = list(T[:3, 3]) + list(rotation_to_quaternion_xyzw(T[:3, :3]))
    return _fixed9(stamp) + " " + "".join(_fixed9(x) + " " for x in v)


def write_trajectory_kitti(path, stamped_poses):
    """stamped_poses: [(timestamp, pose [4,4])], already unrolled; ordered by timestamp on the way out."""
    with open(path, "w") as f:  # an unwritable path raises, like the reference (:211-214)
        for _, pose in sorted(stamped_poses, key=lambda e: e[0]):
            f.write(kitti_line(pose) + "\n")


def write_trajectory_tum(path, stamped_poses):
    with open(path, "w") as f:
        for stamp, pose in sorted(stamped_poses, key=lambda e: e[0]):
            f.write(tum_line(stamp, pose) + "\n")


def read_trajectory_kitti(path):
    """-> float64 [n, 4, 4] (last row 0 0 0 1)"""
    out = []
    with open(path) as f:
        for line in f:
            v = line.split()
            if not v:
                continue
            if len(v) != 12:
                raise ValueError("read_trajectory_kitti|expected 12 values per line, got %d" % len(v))
            T = np.eye(4)
            T[:3, :4] = np.array(v, np.float64).reshape(3, 4)
            out.append(T)
    return np.array(out).reshape(-1, 4, 4)


def read_trajectory_tum(path):
    """-> (timestamps float64 [n], poses float64 [n, 4, 4])"""
    stamps, poses = [], []
    with open(path) as f:
        for line in f:
            v = line.split()
            if not v or v[0].startswith("#"):
                continue
            if len(v) != 8:
                raise ValueError("read_trajectory_tum|expected 8 values per line, got %d" % len(v))
            ts, tx, ty, tz, qx, qy, qz, qw = (float(x) for x in v)
            n = math.sqrt(qx * qx + qy * qy + qz * qz + qw * qw)
            qx, qy, qz, qw = qx / n, qy / n, qz / n, qw / n
            T = np.eye(4)
            T[:3, :3] = [[1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * qw), 2 * (qx * qz + qy * qw)],
                         [2 * (qx * qy + qz * qw), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * qw)],
                         [2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx * qx + qy * qy)]]
            T[:3, 3] = [tx, ty, tz]
            stamps.append(ts)
            poses.append(T)
    return np.array(stamps), np.array(poses).reshape(-1, 4, 4)


# ---------------------------------------------------------------- .conf reader (subset)
def _strip_line_comments(text):
    out, in_string, i = [], False, 0
    while i < len(text):
        ch = text[i]
        if in_string:
            out.append(ch)
            if ch == "\\" and i + 1 < len(text):
                out.append(text[i + 1])
                i += 1
            elif ch == '"':
                in_string = False
        elif ch == '"':
            in_string = True
            out.append(ch)
        elif ch == "/" and text[i + 1:i + 2] == "/":
            while i < len(text) and text[i] != "\n":
                i += 1
            continue
        else:
            out.append(ch)
        i += 1
    return "".join(out)


class ConfRecord(dict):
    """one `"ClassName" { ... }` record: a dict of its fields plus `.class_name` and `.id`"""

    def __init__(self, class_name, fields):
        super().__init__(fields)
        self.class_name = class_name
        self.id = fields.get("#id")


class Conf:
    """records of a `.conf` file in file order, by id, by `name` field and by class"""

    def __init__(self, records):
        self.records = records
        self.by_id = {r.id: r for r in records if r.id is not None}
        self.by_name = {r["name"]: r for r in records if isinstance(r.get("name"), str) and r.get("name")}

    def of_class(self, class_name):
        return [r for r in self.records if r.class_name == class_name]

    def deref(self, value):
        """`{"#pointer": id}` -> the record (None for -1 / unknown ids); anything else is returned as it is"""
        if isinstance(value, dict) and "#pointer" in value:
            return self.by_id.get(value["#pointer"])
        return value

    def follow(self, record, *fields):
        """record.field1 -> pointer -> .field2 -> ... ; None as soon as a link is missing"""
        cur = record
        for name in fields:
            if cur is None or name not in cur:
                return None
            cur = self.deref(cur[name])
        return cur


def parse_conf(text):
    """-> Conf.  Raises ValueError on text that is not a sequence of `"ClassName" { json }` records."""
    clean = _strip_line_comments(text)
    dec = json.JSONDecoder()
    pos, records = 0, []
    while True:
        while pos < len(clean) and clean[pos].isspace():
            pos += 1
        if pos >= len(clean):
            break
        try:
            class_name, pos = dec.raw_decode(clean, pos)
            while pos < len(clean) and clean[pos].isspace():
                pos += 1
            fields, pos = dec.raw_decode(clean, pos)
        except json.JSONDecodeError as e:
            raise ValueError("parse_conf|not a record at offset %d: %s" % (pos, e)) from None
        if not isinstance(class_name, str) or not isinstance(fields, dict):
            raise ValueError("parse_conf|expected \"ClassName\" { ... } at offset %d" % pos)
        records.append(ConfRecord(class_name, fields))
    return Conf(records)


def read_conf(path):
    with open(path) as f:
        return parse_conf(f.read())


_SEARCH_CLASSES = (("CorrespondenceFinderProjectiveKDTree", 0), ("CorrespondenceFinderProjectiveSquare", 1),
                   ("CorrespondenceFinderProjectiveCircle", 2), ("CorrespondenceFinderProjectiveRhombus", 3))
_SLICE_FACTORS = (("AlignerSliceProcessorProjectiveStereo", 4), ("AlignerSliceProcessorProjectiveDepth", 3),
                  ("AlignerSliceProcessorProjective", 2))  # error dimension of the slice's factor (configs.FACTOR_*)


def _pick(record, keys):
    return {k: record[k] for k in keys if k in record}


def hot_path_params(conf):
    """the hot-path parameter groups of a parsed configuration, in the shape of `configs.py`'s dictionaries,
    following the file's own wiring: stereo adaptor -> epipolar finder; projective slice processor -> finder,
    projector, robustifier; the aligner is the MultiAligner whose `slice_processors` hold that slice.
    Only fields present in the file are returned (the reference fills the rest with its class defaults)."""
    out = {}
    adaptor = next((r for r in conf.records if r.class_name.startswith("RawDataPreprocessorStereoProjective")), None)
    matcher = conf.follow(adaptor, "correspondence_finder") if adaptor is not None else None
    if matcher is not None:
        out["stereo_matcher"] = _pick(matcher, ("maximum_descriptor_distance", "maximum_distance_ratio_to_second_best",
                                                "minimum_matching_ratio", "maximum_disparity_pixels", "epipolar_line_thickness_pixels"))
    tri = next((r for r in conf.records if r.class_name.startswith("TriangulatorRigidStereo")), None)
    if tri is not None:
        out["triangulator"] = _pick(tri, ("minimum_disparity_pixels", "infinity_depth_meters"))
    slice_record, factor = None, None
    for prefix, dim in _SLICE_FACTORS:
        slice_record = next((r for r in conf.records if r.class_name.startswith(prefix)), None)
        if slice_record is not None:
            factor = dim
            break
    if slice_record is None:
        return out
    finder = conf.follow(slice_record, "finder")
    if finder is not None:
        d = _pick(finder, ("maximum_descriptor_distance", "maximum_distance_ratio_to_second_best", "minimum_matching_ratio",
                           "minimum_descriptor_distance", "descriptor_distance_step_size_pixels", "maximum_search_radius_pixels",
                           "minimum_search_radius_pixels", "search_radius_step_size_pixels", "minimum_number_of_iterations",
                           "maximum_estimate_change_norm_for_convergence", "number_of_solver_iterations_per_projection"))
        for prefix, kind in _SEARCH_CLASSES:
            if finder.class_name.startswith(prefix):
                d["search_type"] = kind
        out["projective_finder"] = d
    projector = conf.follow(slice_record, "projector")
    if projector is not None:
        out["projector"] = _pick(projector, ("range_min", "range_max"))
    al = {"factor_type": factor}
    if "diagonal_info_matrix" in slice_record:
        al["diagonal_info"] = tuple(slice_record["diagonal_info_matrix"])
    al.update(_pick(slice_record, ("enable_inverse_depth_weighting", "min_num_correspondences")))
    robustifier = conf.follow(slice_record, "robustifier")
    if robustifier is not None and "chi_threshold" in robustifier:
        al["chi_threshold"] = robustifier["chi_threshold"]
    for r in conf.records:
        if r.class_name.startswith("MultiAligner") and any(conf.deref(p) is slice_record for p in r.get("slice_processors", [])):
            al.update(_pick(r, ("max_iterations", "min_num_inliers", "enable_inlier_only_runs", "keep_only_inlier_correspondences")))
            algorithm = conf.follow(r, "solver", "algorithm")
            if algorithm is not None and "damping" in algorithm:
                al["damping"] = algorithm["damping"]
            break
    out["aligner"] = al
    return out
