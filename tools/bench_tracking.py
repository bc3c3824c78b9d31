#!/usr/bin/env python3
"""Closed tracking loop on one MI355X along the KITTI-00 ground-truth trajectory: B independent
synthetic sequences, every frame goes
   constant-velocity prediction -> stereo matcher (+ adaptor / triangulator) -> scene clipper on the resident local map
   -> projective finder + GN aligner (motion-model prior slice on) -> pose update -> merger (pose-based smoother)
with every buffer chained on the device (SURVEY.md 8a + 8f rows 1 and 2).  The finder objects live across frames like
the reference's (adaptive radius / threshold schedule, retries, track loss), the camera follows the real 4541-pose
KITTI-00 path (tests/golden/ref_kitti_gt.npz), and a fresh local map is seeded from the current frame whenever the
camera has moved 10 m / 0.25 rad from the map's origin (LocalMapSplittingCriterionViewpoint3D, kitti.conf:542-550; local
map management itself is outside SURVEY.md section 8, this is the harness's stand-in for it).
bench.py measures the hot path alone; `bench.py --mode closed-loop` calls run() below.  A number of sequences is
replayed frame by frame on the CPU oracle and compared.

    python tools/bench_tracking.py [--batch 4096] [--frames 60] [--unique 5] [--keypoints 2000] [--check 1]
prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

WARN_RETRIED, WARN_TRACK_LOST = 8, 16


def kitti00_poses(n):
    """camera_k in world, float64 4x4 (test_data/kitti/city/gt.txt through tools/make_ref_fixtures.py)"""
    gt = np.load(os.path.join(ROOT, "tests", "golden", "ref_kitti_gt.npz"))["city"][:n].astype(np.float64)
    T = np.tile(np.eye(4), (len(gt), 1, 1))
    T[:, :3, :4] = gt.reshape(-1, 3, 4)
    return T


def split_schedule(poses, distance=10.0, angle=0.25):
    """frames after which a new local map starts (kitti.conf:542-550: 10 m or 0.25 rad from the map's origin)"""
    origin, split_after = 0, []
    for k in range(1, len(poses)):
        rel = np.linalg.inv(poses[origin]) @ poses[k]
        ang = np.arccos(np.clip((np.trace(rel[:3, :3]) - 1.0) / 2.0, -1.0, 1.0))
        if np.linalg.norm(rel[:3, 3]) > distance or ang > angle:
            split_after.append(k)
            origin = k
    return set(split_after)


def make_sequences(cfg, n_seq, poses, n_kp, seed, per_slice=None):
    """static landmark worlds along the path + the stereo frames they produce.  The world is a corridor (40 m wide, 6 m
    high) filled slice by slice: the slice that enters the view 80 m ahead of pose j, plus the slices already in view at
    pose 0, so the number of visible landmarks is stationary (~0.45 N) and near structure keeps arriving."""
    from srrg2_proslam_amd import synthetic as syn
    cam, z_far = cfg["camera"], cfg["depth"]["max"]
    steps = np.linalg.norm(np.diff(poses[:, :3, 3], axis=0), axis=1)
    step = float(np.median(steps)) if len(steps) else 0.86
    if per_slice is None:
        per_slice = max(int(round(0.0042 * n_kp)), 1)
    n_back = int(np.ceil((z_far - 2.0) / step))
    seqs = []
    for u in range(n_seq):
        rng = np.random.default_rng(seed + u)
        W = []
        for j in range(-n_back, len(poses)):
            P = poses[max(j, 0)]
            z0 = z_far + min(j, 0) * step  # slices behind the first pose's far plane fill its initial view
            pc = np.stack([rng.uniform(-20.0, 20.0, per_slice), rng.uniform(-4.35, 1.65, per_slice), rng.uniform(z0 - step, z0, per_slice)], axis=1)
            W.append(pc @ P[:3, :3].T + P[:3, 3])
        W = np.concatenate(W)
        D = syn.random_descriptors(rng, len(W))
        frames = []
        for P in poses:
            Ti = np.linalg.inv(P)
            pk = W @ Ti[:3, :3].T + Ti[:3, 3]
            near = (pk[:, 2] > 2.0) & (pk[:, 2] < 1.2 * z_far) & (np.abs(pk[:, 0]) < pk[:, 2]) & (np.abs(pk[:, 1]) < 0.4 * pk[:, 2])
            frames.append(syn.stereo_frame(rng, cfg, n_kp, landmarks=pk[near].astype(np.float32), landmark_desc=D[near]))
        seqs.append(frames)
    return seqs


def oracle_chain(cfg, frames, splits, cap, max_meas, prior_info):
    """the same loop on the CPU oracle for one sequence -> (local pose per frame, map size per frame, flags per frame, seconds)"""
    from oracle import binding as ob
    from oracle import binding_mapping as om
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import aligner_params as oap, oracle_stereo_params, oracle_tri_params, pcf_params_from_cfg
    cam = cfg["camera"]
    K = (cam["fx"], cam["fy"], cam["cx"], cam["cy"])
    est = om.estimator_params(om.EST_SMOOTHER, 4, K, max_dist2=100.0, chi2_delta=1e-6)
    p = om.MergerParams()
    p.variant, p.enable_binning = om.MERGER_STEREO_TRIANGULATION, 1
    p.number_of_row_bins, p.number_of_col_bins = 20, 60
    p.canvas_rows, p.canvas_cols = cam["rows"], cam["cols"]
    p.maximum_distance_appearance, p.target_number_of_merges, p.target_merge_ratio = 100.0, 10 ** 6, 0.5
    p.triangulator = oracle_tri_params(ob, cfg)
    p.fx, p.fy, p.cx, p.cy = K
    p.estimator = est
    m = om.Map(cap, max_meas)
    poses = om.pose_table(max_meas + 1)
    I4 = np.eye(4, dtype=np.float32)
    pose, prev = I4.copy(), I4.copy()
    finder = ob.ProjectiveFinder(pcf_params_from_cfg(ob, cfg))  # ONE object for the whole sequence
    out_pose, out_n, out_flags = [], [], []
    slot = 0
    t0 = time.perf_counter()
    for k, fr in enumerate(frames):
        corr, _ = ob.stereo_match(fr["uv_left"], fr["desc_left"], fr["uv_right"], fr["desc_right"], oracle_stereo_params(ob, cfg["stereo_matcher"]))
        fixed, src = ob.stereo_assemble(fr["uv_left"], fr["uv_right"], corr)
        fdesc = fr["desc_left"][src]
        c = np.zeros(0, ob.CORR_DTYPE)
        imap, flags = None, 0
        if k > 0:
            pred = ob.motion_predict(prev, pose)
            xyzw = m.coords[: m.n_points].copy()
            xyzw[:, 3] = ob.info_scale_from_nopt(m.n_opt[: m.n_points])
            cx, cd, gi, _ = ob.scene_clip(pcf_params_from_cfg(ob, cfg).projector, pred, I4, xyzw, m.desc[: m.n_points])
            finder.set_fixed(fixed, fdesc)
            finder.set_moving(cx[:, :3], cd)
            ap = oap(ob, cfg, mean_disparity=ob.mean_disparity(fixed))
            ap.enable_motion_prior = 1 if prior_info > 0 else 0
            for i in range(6):
                ap.motion_prior_info[i] = prior_info
            res, rc = ob.align_frame(finder, ap, fixed, cx[:, :3], cx[:, 3], I4)
            flags = res.warnings
            prev = pose
            pose = ob.se3_mul(pred, ob.se3_inverse(np.array(res.X, np.float32).reshape(4, 4)))
            c = rc.copy()
            c["fixed_idx"], c["moving_idx"] = rc["moving_idx"], rc["fixed_idx"]
            imap = np.concatenate([gi, np.zeros(cap - len(gi), np.int32)])
        rcode, _ = om.merge(p, pose, pose, poses, slot, m, fixed, fdesc, c, imap)
        if rcode != 0:
            raise SystemExit("oracle merge failed with %d at frame %d" % (rcode, k))
        out_pose.append(pose.copy())
        out_flags.append(flags)
        slot += 1
        if k in splits:
            # new local map with this frame as its origin: the last two poses are re-expressed in it, the map is re-seeded
            inv_pose = ob.motion_predict(pose, I4)  # pose^-1
            prev = ob.se3_mul(inv_pose, ob.se3_inverse(ob.motion_predict(prev, I4)))  # pose^-1 * prev
            pose = I4.copy()
            m = om.Map(cap, max_meas)
            poses = om.pose_table(max_meas + 1)
            rcode, _ = om.merge(p, pose, pose, poses, 0, m, fixed, fdesc, np.zeros(0, ob.CORR_DTYPE), None)
            slot = 1
        out_n.append(m.n_points)
    finder.close()
    return out_pose, out_n, out_flags, time.perf_counter() - t0


def unroll(local_poses, splits):
    """global camera poses (float64) from per-frame local poses and the split schedule"""
    origin = np.eye(4)
    out = []
    for k, T in enumerate(local_poses):
        G = origin @ np.asarray(T, np.float64)
        out.append(G)
        if k in splits:
            origin = G
    return out


def run(batch=4096, frames=60, unique=5, keypoints=2000, cap=6144, check=1, prior_info=1.0, trajectory="", max_fixed=1024, device=0, seed_offset=0):
    import torch
    from bench_merge import merger_params
    from srrg2_proslam_amd import configs, ops, synthetic as syn

    cfg = configs.get("kitti")
    cam = cfg["camera"]
    B, N, K = batch, keypoints, frames
    gt = kitti00_poses(K)
    gt = np.linalg.inv(gt[0]) @ gt  # the first camera is the first local map's origin
    splits = split_schedule(gt)
    longest = max(np.diff([0] + sorted(splits) + [K])) + 2
    dev = torch.device("cuda", device)
    idx = torch.arange(B, device=dev) % unique
    seqs = make_sequences(cfg, unique, gt, N, syn.seed_for(1, 0) + 500000 + seed_offset)
    stage = ops.StereoFrames(device, len(seqs), N, epilogue=False)
    inputs = []
    for k in range(K):  # per-frame inputs of every sequence, resident in HBM
        for u, fr_list in enumerate(seqs):
            fr = fr_list[k]
            stage.upload(u, fr["uv_left"], fr["desc_left"], fr["uv_right"], fr["desc_right"])
        inputs.append(tuple(t.index_select(0, idx).contiguous() for t in (stage.left_kp, stage.left_desc, stage.right_kp, stage.right_desc, stage.n_left, stage.n_right)))
    del stage
    sf = ops.StereoFrames(device, B, N, epilogue=True)
    max_meas = int(longest) + 1
    maps = ops.MapBatch(device, B, cap, max_meas, max_meas + 1, N, N)
    maps.measurement, maps.measurement_desc, maps.n_measured = sf.fixed_uvuv, sf.fixed_desc, sf.n_fixed
    clip = ops.ClipScenes(device, B, cap)
    clip.scene_xyzw, clip.scene_desc, clip.n_scene, clip.scene_n_opt = maps.coords, maps.desc, maps.n_points, maps.n_opt
    af = ops.AlignFrames(device, B, N, cap)
    af.fixed, af.fixed_desc, af.n_fixed = sf.fixed_uvuv, sf.fixed_desc, sf.n_fixed
    af.moving, af.moving_desc, af.n_moving = clip.clipped_xyzw, clip.clipped_desc, clip.n_clipped
    af.max_fixed = max_fixed  # stereo matches per frame the aligner is sized for (<= 1024: search / GN pipeline); more is a loud per-frame error
    maps.corr, maps.corr_from_aligner, maps.scene_index_map = af.corr, 1, clip.global_indices
    zero_corr = torch.zeros((B,), dtype=torch.int32, device=dev)
    eye = torch.eye(4, dtype=torch.float32, device=dev).repeat(B, 1, 1).contiguous()
    pose, prev, pred, tmp_a, tmp_b = eye.clone(), eye.clone(), eye.clone(), eye.clone(), eye.clone()
    ctx = ops.Context(device)
    sp, tp = ops.stereo_params(cfg["stereo_matcher"], cam["rows"]), ops.triangulator_params(cfg)
    pp = ops.pcf_params(cfg)
    apar = ops.aligner_params(cfg)
    if prior_info > 0:
        ops.set_motion_prior(apar, (prior_info,) * 6)  # AlignerSliceMotionModel3D (kitti.conf:747-772); mean = identity
    proj = pp.projector
    mp = merger_params(cfg, ops.EST_SMOOTHER)
    I4 = np.eye(4, dtype=np.float32)
    stream = torch.cuda.Stream(device=dev)
    n_log = min(max(check, 1), unique)
    traj = torch.zeros((K, n_log, 16), dtype=torch.float32, device=dev)
    flag_log = torch.zeros((K, B), dtype=torch.int32, device=dev)
    warn_off = ops.AlignResult.warnings.offset

    def merge(slot, with_corr):
        maps.n_corr = af.n_corr if with_corr else zero_corr
        maps.measurement_in_world.copy_(pose, non_blocking=True)
        maps.measurement_in_scene.copy_(pose, non_blocking=True)
        maps.frame.fill_(slot)
        ops.merge_batch(ctx, mp, maps)

    state = {"slot": 0}

    def frame(k, ev=None):
        sf.left_kp, sf.left_desc, sf.right_kp, sf.right_desc, sf.n_left, sf.n_right = inputs[k]
        if ev:
            ev[0].record()
        ops.stereo_match_batch(ctx, sp, sf, tp)
        if ev:
            ev[1].record()
        if k > 0:
            ops.motion_predict_batch(ctx, prev, pose, pred)  # MotionModelConstantVelocity3D
            clip.robot_in_local_map.copy_(pred, non_blocking=True)
            af.X.copy_(eye.view(B, 16), non_blocking=True)   # the local map is clipped at the prediction: guess = identity
            ops.scene_clip_batch(ctx, proj, I4, clip)
            if ev:
                ev[2].record()
            ops.align_batch(ctx, pp, apar, af)               # finder state carried over from the previous frame
            if ev:
                ev[3].record()
            prev.copy_(pose, non_blocking=True)
            ops.pose_compose_batch(ctx, pred, af.X, pose)    # pose = prediction * X^-1
            flag_log[k].copy_(af.result[:, warn_off: warn_off + 4].contiguous().view(torch.int32).view(B), non_blocking=True)
        merge(state["slot"], k > 0)
        state["slot"] += 1
        traj[k].copy_(pose[:n_log].reshape(n_log, 16), non_blocking=True)
        if k in splits:
            # new local map seeded from this frame: prev <- pose^-1 * prev, pose <- identity
            ops.motion_predict_batch(ctx, pose, eye, tmp_a)      # pose^-1
            ops.motion_predict_batch(ctx, prev, eye, tmp_b)      # prev^-1
            ops.pose_compose_batch(ctx, tmp_a, tmp_b, prev)      # pose^-1 * prev
            pose.copy_(eye, non_blocking=True)
            maps.n_points.zero_()
            maps.n_meas.zero_()
            merge(0, False)
            state["slot"] = 1
        if ev:
            ev[4].record()

    with torch.cuda.stream(stream):
        ctx.use_torch_stream()
        frame(0)
        frame(1)  # warm-up of the tracked path (its effects are kept: the timed region continues with frame 2)
        torch.cuda.synchronize()
        events = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(K)]
        t0 = time.perf_counter()
        for k in range(2, K):
            frame(k, events[k])
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
    status = maps.result[:, 2].cpu().numpy()
    if (status < 0).any():
        raise SystemExit("merge reported error %d" % int(status.min()))
    flags = flag_log.cpu().numpy()
    if (flags < 0).any():
        raise SystemExit("aligner reported error %d at frame %d (fixed cloud above max_fixed?)" % (int(flags.min()), int(np.argwhere(flags < 0)[0][0])))
    tracked = K - 2
    ms = lambda a, b: float(np.mean([events[k][a].elapsed_time(events[k][b]) for k in range(2, K)]))  # noqa: E731
    local = traj.cpu().numpy().reshape(K, n_log, 4, 4)
    est = [unroll(local[:, u], splits) for u in range(n_log)]
    drift = [float(np.linalg.norm(est[u][-1][:3, 3] - gt[-1][:3, 3])) for u in range(n_log)]
    path_len = float(np.sum(np.linalg.norm(np.diff(gt[:, :3, 3], axis=0), axis=1)))
    if trajectory:
        from srrg2_proslam_amd import formats
        unrolled = formats.unroll_trajectory([(np.eye(4), [(0.1 * k, T) for k, T in enumerate(est[0])])])
        formats.write_trajectory_kitti(trajectory + "_kitti.txt", unrolled)
        formats.write_trajectory_tum(trajectory + "_tum.txt", unrolled)
    out = {
        "metric": "tracked frames/sec on KITTI-00 stereo (1241x376, ~2k kp); SE(3) vs ref",
        "value": B * tracked / elapsed, "unit": "frames/s", "n_gpus": 1,
        "ms_per_step": elapsed / tracked * 1e3,
        "config": {"workload": "closed loop along the KITTI-00 ground-truth trajectory: %d sequences x %d frames, %d keypoints per image, kitti.conf "
                               "matcher / finder (state carried across frames) / aligner + motion-model prior / merger (stereo triangulation + "
                               "pose-based smoother, 20 x 60 bins), constant-velocity prediction, map capacity %d, fresh local map every 10 m "
                               "(%d maps)" % (B, K, N, cap, len(splits) + 1),
                   "frames_per_step_per_gpu": B, "tracked_frames_timed": tracked},
        "ms_per_stage": {"stereo_match": ms(0, 1), "predict+scene_clip": ms(1, 2), "align": ms(2, 3), "pose_update+merge": ms(3, 4)},
        "map_points_mean": float(maps.n_points.float().mean().item()),
        "aligner_correspondences_mean": float(af.n_corr.float().mean().item()),
        "finder_retries_per_frame": float(((flags & WARN_RETRIED) != 0).mean()),
        "track_losses_per_frame": float(((flags & WARN_TRACK_LOST) != 0).mean()),
        "trajectory_length_m": path_len, "end_point_drift_m": drift, "drift_percent_of_path": [100.0 * d / path_len for d in drift],
    }
    if check > 0:
        worst, worst_t, exact_n, cpu_s, cpu_frames, same_flags = 0.0, 0.0, True, 0.0, 0, True
        for u in range(min(check, len(seqs))):
            op, on, of_, dt = oracle_chain(cfg, seqs[u], splits, cap, max_meas, prior_info)
            cpu_s += dt
            cpu_frames += K
            for k in range(K):
                worst = max(worst, float(np.linalg.norm(local[k, u] - op[k]) / np.linalg.norm(op[k])))
                same_flags = same_flags and int(flags[k, u]) == int(of_[k])
            worst_t = max(worst_t, float(np.linalg.norm(est[u][-1][:3, 3] - unroll(op, splits)[-1][:3, 3])))
            exact_n = exact_n and int(maps.n_points[u].item()) == on[-1]
        out["parity_vs_oracle_chain"] = {"pose_rel_frobenius_max_over_all_frames": worst, "end_point_difference_m": worst_t, "map_size_equal": exact_n,
                                         "finder_flags_equal_every_frame": same_flags, "sequences_checked": min(check, len(seqs)), "frames_each": K}
        out["cpu_baseline"] = {"value": cpu_frames / cpu_s, "unit": "frames/s", "cores": 1, "kind": "port",
                               "sample": "%d frames of the same loop on the oracle, %.1f s" % (cpu_frames, cpu_s)}
    ctx.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--frames", type=int, default=60, help="frames per sequence along the KITTI-00 path (frames 0 and 1 are untimed)")
    ap.add_argument("--unique", type=int, default=5, help="distinct sequences generated on the host and tiled (odd: workgroup b runs on XCD b mod 8)")
    ap.add_argument("--keypoints", type=int, default=2000)
    ap.add_argument("--cap", type=int, default=6144, help="landmark capacity of a local map")
    ap.add_argument("--check", type=int, default=1, help="sequences replayed on the CPU oracle (0 = skip)")
    ap.add_argument("--prior", type=float, default=1.0, help="information of the motion-model prior slice (0 = off)")
    ap.add_argument("--trajectory", default="", help="write sequence 0's estimated trajectory as <prefix>_kitti.txt and <prefix>_tum.txt "
                    "(the reference benchmark's formats, apps/app_benchmark.cpp:205-259; 10 Hz timestamps)")
    ap.add_argument("--max-fixed", type=int, default=1024, help="bound on the stereo matches per frame (above 1024 the one-kernel aligner runs)")
    args = ap.parse_args()
    print(json.dumps(run(args.batch, args.frames, args.unique, args.keypoints, args.cap, args.check, args.prior, args.trajectory, args.max_fixed)))


if __name__ == "__main__":
    main()
