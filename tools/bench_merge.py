"""merger throughput: B maps x one frame each (kitti.conf merger: stereo triangulation + pose-based smoother)
usage: python tools/bench_merge.py [B] [N_measured] [N_scene]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from srrg2_proslam_amd import _lib, configs, ops  # noqa: E402


def merger_params(cfg, est_type):
    cam = cfg["camera"]
    p = _lib.MergerParams()
    p.variant, p.enable_binning = ops.MERGER_STEREO_TRIANGULATION, 1
    p.number_of_row_bins, p.number_of_col_bins = 20, 60  # kitti.conf:203-207
    p.canvas_rows, p.canvas_cols = cam["rows"], cam["cols"]
    p.maximum_distance_appearance, p.target_number_of_merges, p.target_merge_ratio = 100.0, 10 ** 6, 0.5
    p.triangulator = ops.triangulator_params(cfg)
    p.fx, p.fy, p.cx, p.cy = cam["fx"], cam["fy"], cam["cx"], cam["cy"]
    e = p.estimator
    e.type, e.measurement_dim = est_type, 4
    e.maximum_distance_geometry_meters_squared = 100.0
    e.minimum_state_element_covariance, e.maximum_covariance_norm_squared = 0.01, 0.25
    e.fx, e.fy, e.cx, e.cy = cam["fx"], cam["fy"], cam["cx"], cam["cy"]
    e.b_x, e.b_y = cam["fx"] * cam["baseline_m"], 0.0
    e.maximum_number_of_iterations, e.convergence_criterion_minimum_chi2_delta = 100, 1e-6
    e.maximum_reprojection_error_pixels_squared, e.minimum_number_of_measurements_for_optimization = 100.0, 3
    for i, v in enumerate([cam["fx"], 0, cam["cx"], 0, cam["fy"], cam["cy"], 0, 0, 1]):
        e.camera_matrix[i] = v
    return p


def run(B, NM, NS, est_type, name, history=6, quiet=False):
    cfg = configs.get("kitti")
    cam = cfg["camera"]
    ctx = ops.Context(0)
    ctx.use_torch_stream()
    K = 8
    maps = ops.MapBatch(0, B, NS + NM, K, 16, NM, NM)
    g = torch.Generator(device="cuda").manual_seed(3)
    # scene: NS landmarks with `history` earlier measurements each, all seen from the identity pose
    z = torch.rand((B, NS), device="cuda", generator=g) * 40 + 6
    x = (torch.rand((B, NS), device="cuda", generator=g) - 0.5) * z
    y = (torch.rand((B, NS), device="cuda", generator=g) - 0.5) * 0.3 * z
    P = torch.stack([x, y, z, torch.zeros_like(z)], dim=-1)
    maps.coords[:, :NS] = P
    maps.state[:, :NS] = P
    maps.covariance[:, :NS] = torch.eye(3, device="cuda").reshape(9)
    maps.n_points.fill_(NS)
    uL = cam["fx"] * x / z + cam["cx"]
    vL = cam["fy"] * y / z + cam["cy"]
    uR = uL - cam["fx"] * cam["baseline_m"] / z
    if est_type == ops.EST_SMOOTHER:
        m = torch.zeros((B, NS, K, 7), device="cuda", dtype=torch.float32)
        for h in range(history):
            m[:, :, h, 0], m[:, :, h, 1], m[:, :, h, 2] = uL, vL, uR
            m[:, :, h, 3], m[:, :, h, 4], m[:, :, h, 5] = x, y, z
        maps.meas[:, :NS] = m.view(torch.int32)
        maps.meas[:, :NS, :, 6] = 0
        maps.n_meas[:, :NS] = history
        maps.n_opt[:, :NS] = history
    eye12 = torch.eye(4, device="cuda")[:3].reshape(12)
    maps.poses[:, :, :12] = eye12
    maps.poses[:, :, 12:] = eye12
    # frame: the first min(NM, NS) measurements re-observe scene points (identity correspondences), the rest are new
    n_c = min(NM, NS)
    meas = torch.zeros((B, NM, 4), device="cuda")
    meas[:, :n_c, 0], meas[:, :n_c, 1], meas[:, :n_c, 2], meas[:, :n_c, 3] = uL[:, :n_c], vL[:, :n_c], uR[:, :n_c], vL[:, :n_c]
    if NM > n_c:
        meas[:, n_c:, 0] = torch.rand((B, NM - n_c), device="cuda", generator=g) * (cam["cols"] - 80) + 60
        meas[:, n_c:, 1] = torch.rand((B, NM - n_c), device="cuda", generator=g) * (cam["rows"] - 1)
        meas[:, n_c:, 2] = meas[:, n_c:, 0] - torch.rand((B, NM - n_c), device="cuda", generator=g) * 50 - 2
        meas[:, n_c:, 3] = meas[:, n_c:, 1]
    maps.measurement.copy_(meas)
    maps.n_measured.fill_(NM)
    idx = torch.arange(n_c, device="cuda", dtype=torch.int32)
    maps.corr[:, :n_c, 0] = idx
    maps.corr[:, :n_c, 1] = idx
    maps.corr[:, :n_c, 2] = torch.full((n_c,), 10.0, device="cuda").view(torch.int32)
    maps.n_corr.fill_(n_c)
    maps.frame.fill_(1)
    p = merger_params(cfg, est_type)
    # the map is consumed by a merge, so every timed launch starts from a fresh copy of the counters
    snap = (maps.n_points.clone(), maps.n_meas.clone(), maps.n_opt.clone(), maps.state.clone(), maps.coords.clone())
    times = []
    for it in range(6):
        maps.n_points.copy_(snap[0]); maps.n_meas.copy_(snap[1]); maps.n_opt.copy_(snap[2]); maps.state.copy_(snap[3]); maps.coords.copy_(snap[4])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.merge_batch(ctx, p, maps)
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1))
    ms = float(np.median(times[1:]))
    r = maps.result.cpu().numpy()
    assert (r[:, 2] >= 0).all(), r[:4]
    merged, added = r[:, 0].mean(), r[:, 1].mean()
    per_lm = 16 + 16 + 36 + 32 + 12 + 28 * (history + 1 if est_type == ops.EST_SMOOTHER else 0)  # row bytes touched per merged landmark
    bytes_ = B * (NM * 48 + n_c * 12 + merged * 2 * per_lm + added * (16 + 16 + 36 + 32 + 12 + 28))
    if not quiet:
        print("%-14s B=%d measured=%d scene=%d merged=%.0f added=%.0f: %.3f ms/launch, %.2f M frames/s, %.0f GB/s of landmark-row traffic" % (
            name, B, NM, NS, merged, added, ms, B / ms / 1e3, bytes_ / ms / 1e6))
    ctx.close()
    del maps
    torch.cuda.empty_cache()
    return {"estimator": name, "maps_per_launch": B, "measurements_per_frame": NM, "scene_points": NS, "history_per_landmark": history,
            "merged_per_frame": float(merged), "added_per_frame": float(added), "ms_per_launch": ms, "frames_per_s": B / (ms * 1e-3),
            "algorithmic_bytes_per_launch": float(bytes_), "gbps": float(bytes_) / ms / 1e6}


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    NM = int(sys.argv[2]) if len(sys.argv) > 2 else 704
    NS = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
    run(B, NM, NS, ops.EST_SMOOTHER, "smoother")
    run(B, NM, NS, ops.EST_WEIGHTED_MEAN, "weighted mean")
    run(B, NM, NS, ops.EST_EKF, "stereo EKF f64")
