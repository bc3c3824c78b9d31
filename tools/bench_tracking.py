#!/usr/bin/env python3
"""Closed tracking loop on one MI355X: B independent synthetic KITTI-shaped sequences, every frame goes
   stereo matcher (+ adaptor / triangulator) -> scene clipper on the resident local map -> projective finder +
   GN aligner -> pose update -> merger (pose-based smoother + binned additions)
with every buffer chained on the device (SURVEY.md 8a + 8f rows 1 and 2).  bench.py measures the hot path
alone; this tool measures the whole per-frame loop including map maintenance and checks it against the same
chain on the CPU oracle.

    python tools/bench_tracking.py [--batch 7680] [--frames 13] [--unique 8] [--keypoints 2000]
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


def camera_pose(k):
    from srrg2_proslam_amd import synthetic as syn
    T = syn.make_transform((0.03 * k, -0.01 * k, 0.7 * k), (0.002 * k, 0.012 * k, -0.001 * k))
    return np.asarray(T, dtype=np.float32).reshape(4, 4)


def make_sequences(cfg, n_seq, n_frames, n_kp, seed):
    from srrg2_proslam_amd import synthetic as syn
    cam = cfg["camera"]
    seqs = []
    for u in range(n_seq):
        rng = np.random.default_rng(seed + u)
        n_lm = int(round(n_kp * 0.45))
        W = syn.sample_landmarks(rng, cam, cfg["depth"], n_lm)
        D = syn.random_descriptors(rng, n_lm)
        frames = []
        for k in range(n_frames):
            Ti = np.linalg.inv(camera_pose(k).astype(np.float64))
            pk = ((Ti[:3, :3] @ W.T.astype(np.float64)).T + Ti[:3, 3]).astype(np.float32)
            frames.append(syn.stereo_frame(rng, cfg, n_kp, landmarks=pk, landmark_desc=D))
        seqs.append(frames)
    return seqs


def oracle_chain(cfg, frames, cap, max_meas):
    """the same loop on the CPU oracle for one sequence -> (poses per frame, map size per frame, seconds)"""
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
    poses = om.pose_table(len(frames) + 1)
    I4 = np.eye(4, dtype=np.float32)
    pose = I4.copy()
    out_pose, out_n = [], []
    t0 = time.perf_counter()
    for k, fr in enumerate(frames):
        corr, _ = ob.stereo_match(fr["uv_left"], fr["desc_left"], fr["uv_right"], fr["desc_right"], oracle_stereo_params(ob, cfg["stereo_matcher"]))
        fixed, src = ob.stereo_assemble(fr["uv_left"], fr["uv_right"], corr)
        fdesc = fr["desc_left"][src]
        c = np.zeros(0, ob.CORR_DTYPE)
        imap = None
        if k > 0:
            xyzw = m.coords[: m.n_points].copy()
            xyzw[:, 3] = ob.info_scale_from_nopt(m.n_opt[: m.n_points])
            cx, cd, gi, _ = ob.scene_clip(pcf_params_from_cfg(ob, cfg).projector, pose, I4, xyzw, m.desc[: m.n_points])
            of = ob.ProjectiveFinder(pcf_params_from_cfg(ob, cfg))
            of.set_fixed(fixed, fdesc)
            of.set_moving(cx[:, :3], cd)
            res, rc = ob.align_frame(of, oap(ob, cfg, mean_disparity=ob.mean_disparity(fixed)), fixed, cx[:, :3], cx[:, 3], I4)
            of.close()
            pose = ob.se3_mul(pose, ob.se3_inverse(np.array(res.X, np.float32).reshape(4, 4)))
            c = rc.copy()
            c["fixed_idx"], c["moving_idx"] = rc["moving_idx"], rc["fixed_idx"]
            imap = np.concatenate([gi, np.zeros(cap - len(gi), np.int32)])
        rcode, _ = om.merge(p, pose, pose, poses, k, m, fixed, fdesc, c, imap)
        if rcode != 0:
            raise SystemExit("oracle merge failed with %d at frame %d" % (rcode, k))
        out_pose.append(pose.copy())
        out_n.append(m.n_points)
    return out_pose, out_n, time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=7680)
    ap.add_argument("--frames", type=int, default=13, help="frames per sequence (frame 0 seeds the map, untimed)")
    ap.add_argument("--unique", type=int, default=8, help="distinct sequences generated on the host and tiled")
    ap.add_argument("--keypoints", type=int, default=2000)
    ap.add_argument("--cap", type=int, default=3072, help="landmark capacity of a local map")
    ap.add_argument("--check", type=int, default=2, help="sequences replayed on the CPU oracle (0 = skip)")
    ap.add_argument("--from-images", action="store_true",
                    help="start every frame from a rectified 8-bit stereo image pair (feature extraction on the device); "
                         "the scene is a static layered world, the camera steps sideways (use a smaller --batch: the images stay resident)")
    ap.add_argument("--trajectory", default="", help="write sequence 0's estimated trajectory as <prefix>_kitti.txt and <prefix>_tum.txt "
                    "(the reference benchmark's formats, apps/app_benchmark.cpp:205-259; 10 Hz timestamps)")
    args = ap.parse_args()
    import torch
    from bench_merge import merger_params
    from srrg2_proslam_amd import configs, ops, synthetic as syn

    cfg = configs.get("kitti")
    cam = cfg["camera"]
    B, N, K, cap = args.batch, args.keypoints, args.frames, args.cap
    dev = torch.device("cuda", 0)
    idx = torch.arange(B, device=dev) % args.unique
    inputs, images, step_m = [], [], 0.0
    if args.from_images:
        N = 1024  # feature capacity per image (kitti.conf: target 1000)
        img_seqs = [syn.stereo_image_sequence(np.random.default_rng(syn.seed_for(1, 0) + 600000 + u), cfg, K) for u in range(args.unique)]
        step_m = img_seqs[0][1]
        for k in range(K):
            L = torch.from_numpy(np.stack([s[0][k][0] for s in img_seqs])).to(dev).index_select(0, idx).contiguous()
            R = torch.from_numpy(np.stack([s[0][k][1] for s in img_seqs])).to(dev).index_select(0, idx).contiguous()
            images.append((L, R))
        seqs = img_seqs
    else:
        seqs = make_sequences(cfg, args.unique, K, N, syn.seed_for(1, 0) + 500000)
        # per-frame inputs of every sequence, resident in HBM
        stage = ops.StereoFrames(0, len(seqs), N, epilogue=False)
        for k in range(K):
            for u, frames in enumerate(seqs):
                fr = frames[k]
                stage.upload(u, fr["uv_left"], fr["desc_left"], fr["uv_right"], fr["desc_right"])
            inputs.append(tuple(t.index_select(0, idx).contiguous() for t in (stage.left_kp, stage.left_desc, stage.right_kp, stage.right_desc, stage.n_left, stage.n_right)))
        del stage
    sf = ops.StereoFrames(0, B, N, epilogue=True)
    ext_status = torch.zeros((B,), dtype=torch.int32, device=dev)
    ep = ops.extractor_params()
    max_meas = K + 1
    maps = ops.MapBatch(0, B, cap, max_meas, K + 1, N, N)
    maps.measurement, maps.measurement_desc, maps.n_measured = sf.fixed_uvuv, sf.fixed_desc, sf.n_fixed
    clip = ops.ClipScenes(0, B, cap)
    clip.scene_xyzw, clip.scene_desc, clip.n_scene, clip.scene_n_opt = maps.coords, maps.desc, maps.n_points, maps.n_opt
    af = ops.AlignFrames(0, B, N, cap)
    af.fixed, af.fixed_desc, af.n_fixed = sf.fixed_uvuv, sf.fixed_desc, sf.n_fixed
    af.moving, af.moving_desc, af.n_moving = clip.clipped_xyzw, clip.clipped_desc, clip.n_clipped
    af.max_fixed = 896
    maps.corr, maps.corr_from_aligner, maps.scene_index_map = af.corr, 1, clip.global_indices
    zero_corr = torch.zeros((B,), dtype=torch.int32, device=dev)
    state0 = af.state.clone()
    eye = torch.eye(4, dtype=torch.float32, device=dev).repeat(B, 1, 1).contiguous()
    pose = eye.clone()
    ctx = ops.Context(0)
    sp, tp = ops.stereo_params(cfg["stereo_matcher"], cam["rows"]), ops.triangulator_params(cfg)
    pp, apar = ops.pcf_params(cfg), ops.aligner_params(cfg)
    proj = pp.projector
    mp = merger_params(cfg, ops.EST_SMOOTHER)
    I4 = np.eye(4, dtype=np.float32)
    stream = torch.cuda.Stream(device=dev)
    poses_log = []
    traj = torch.zeros((K, 16), dtype=torch.float32, device=dev) if args.trajectory else None

    def frame(k, ev=None):
        if ev:
            ev[5].record()
        if args.from_images:
            ops.extract_features_batch(ctx, ep, images[k][0], sf.left_kp, sf.left_desc, sf.n_left, ext_status)
            ops.extract_features_batch(ctx, ep, images[k][1], sf.right_kp, sf.right_desc, sf.n_right, ext_status)
        else:
            sf.left_kp, sf.left_desc, sf.right_kp, sf.right_desc, sf.n_left, sf.n_right = inputs[k]
        if ev:
            ev[0].record()
        ops.stereo_match_batch(ctx, sp, sf, tp)
        if ev:
            ev[1].record()
        if k > 0:
            clip.robot_in_local_map.copy_(pose, non_blocking=True)
            af.state.copy_(state0, non_blocking=True)
            af.X.copy_(eye.view(B, 16), non_blocking=True)
            af.n_corr.zero_()
            ops.scene_clip_batch(ctx, proj, I4, clip)
            if ev:
                ev[2].record()
            ops.align_batch(ctx, pp, apar, af)
            if ev:
                ev[3].record()
            ops.pose_compose_batch(ctx, clip.robot_in_local_map, af.X, pose)
            maps.n_corr = af.n_corr
        else:
            maps.n_corr = zero_corr
        maps.measurement_in_world.copy_(pose, non_blocking=True)
        maps.measurement_in_scene.copy_(pose, non_blocking=True)
        maps.frame.fill_(k)
        ops.merge_batch(ctx, mp, maps)
        if ev:
            ev[4].record()
        if traj is not None:
            traj[k].copy_(pose[0].reshape(16), non_blocking=True)  # device-to-device, 64 bytes

    with torch.cuda.stream(stream):
        ctx.use_torch_stream()
        frame(0)
        frame(1)  # warm-up of the tracked path (its effects are kept: the timed region continues with frame 2)
        poses_log.append(pose[: len(seqs)].cpu().numpy().copy())
        torch.cuda.synchronize()
        events = [[torch.cuda.Event(enable_timing=True) for _ in range(6)] for _ in range(K)]
        t0 = time.perf_counter()
        for k in range(2, K):
            frame(k, events[k])
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        final_pose = pose[: len(seqs)].cpu().numpy().copy()
    if traj is not None:
        from srrg2_proslam_amd import formats
        # one local map per sequence: its keyframe estimate is the identity, the tracked poses are local to it
        unrolled = formats.unroll_trajectory([(np.eye(4), [(0.1 * k, T) for k, T in enumerate(traj.cpu().numpy().reshape(K, 4, 4))])])
        formats.write_trajectory_kitti(args.trajectory + "_kitti.txt", unrolled)
        formats.write_trajectory_tum(args.trajectory + "_tum.txt", unrolled)
    status = maps.result[:, 2].cpu().numpy()
    if (status < 0).any():
        raise SystemExit("merge reported error %d" % int(status.min()))
    tracked = K - 2
    ms = lambda a, b: float(np.mean([events[k][a].elapsed_time(events[k][b]) for k in range(2, K)]))  # noqa: E731
    truth = camera_pose(K - 1)
    if args.from_images:
        truth = np.eye(4, dtype=np.float32)
        truth[0, 3] = (K - 1) * step_m
    err = float(np.max(np.linalg.norm(final_pose[:, :3, 3] - truth[:3, 3], axis=1)))
    out = {
        "metric": "tracked frames/sec, closed loop (matcher -> clipper -> finder/aligner -> pose update -> merger) on KITTI-shaped synthetic stereo",
        "value": B * tracked / elapsed, "unit": "frames/s", "n_gpus": 1,
        "frames_per_step": B, "tracked_frames_timed": tracked, "ms_per_frame_step": elapsed / tracked * 1e3,
        "ms_per_stage": {"feature_extraction": ms(5, 0), "stereo_match": ms(0, 1), "scene_clip": ms(1, 2), "align": ms(2, 3),
                         "pose_update+merge": ms(3, 4)},
        "from_images": bool(args.from_images),
        "map_points_mean": float(maps.n_points.float().mean().item()),
        "merged_per_frame_mean": float(maps.result[:, 0].float().mean().item()),
        "added_per_frame_mean": float(maps.result[:, 1].float().mean().item()),
        "aligner_correspondences_mean": float(af.n_corr.float().mean().item()),
        "trajectory_error_max_m": err, "trajectory_length_m": float(np.linalg.norm(truth[:3, 3])),
        "config": {"workload": "%d sequences x %d frames, %d keypoints per image, kitti.conf matcher / finder / aligner / merger (stereo "
                               "triangulation + pose-based smoother, 20 x 60 bins), map capacity %d" % (B, K, N, cap)},
    }
    if args.check > 0 and not args.from_images:
        worst, exact_n, cpu_s, cpu_frames = 0.0, True, 0.0, 0
        for u in range(min(args.check, len(seqs))):
            op, on, dt = oracle_chain(cfg, seqs[u], cap, max_meas)
            cpu_s += dt
            cpu_frames += K
            worst = max(worst, float(np.linalg.norm(final_pose[u] - op[-1]) / np.linalg.norm(op[-1])))
            exact_n = exact_n and int(maps.n_points[u].item()) == on[-1]
        out["parity_vs_oracle_chain"] = {"pose_rel_frobenius_max": worst, "map_size_equal": exact_n, "sequences_checked": min(args.check, len(seqs))}
        out["cpu_baseline"] = {"value": cpu_frames / cpu_s, "unit": "frames/s", "cores": 1, "kind": "port",
                               "sample": "%d frames of the same loop on the oracle, %.1f s" % (cpu_frames, cpu_s)}
    print(json.dumps(out))
    ctx.close()


if __name__ == "__main__":
    main()
