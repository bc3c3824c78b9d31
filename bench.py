#!/usr/bin/env python3
"""Throughput bench of the per-frame tracking hot path on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch of synthetic KITTI-00-shaped input:
for each of B independent sequences (per GPU) one tracked frame =
    stereo epipolar match (2 x N keypoints, 256-bit descriptors)
      -> stereo adaptor assembly + rectified triangulation (fused epilogue)
      -> projective correspondence finder + reprojection-error GN aligner (100 iterations, kitti.conf)
All inputs are resident in HBM before the timed region.  Sequences are independent, so ranks
shard them with no data-path collective (weak scaling: per-GPU work is fixed).

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line (contract in the task statement) with these extra objects:
  roofline          the time-dominant kernel (the aligner's Gauss-Newton kernel, VALU-bound): achieved algorithmic
                    fp32 TFLOP/s against the 157.3 TFLOP/s vector peak, live HIP-event time of its launches
  roofline_search   the projective search kernel, roofline_matcher the stereo matcher kernel (the one BASELINE.json's
                    north star prices: achieved algorithmic HBM GB/s against 8 TB/s)
  steady_state      the same step with the finder objects carried over from frame to frame (the reference's adaptive
                    radius / threshold schedule live) instead of fresh finders
  cpu_baseline      the single-threaded CPU restatement (oracle, "port") timed on a bounded sample
                    of the same frames on this box's host cores (rank 0, N=1 only)
`--mode closed-loop` runs the whole per-frame loop (matcher -> clipper -> finder / aligner -> pose update -> merger)
along the KITTI-00 ground-truth trajectory instead (tools/bench_tracking.py) and prints its line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_VECTOR_TFLOPS = 157.3  # MI355X_MICROARCH.md: peak FP32 (vector)
FLOP_PER_CORRESPONDENCE = 250.0  # one linearised correspondence of one GN iteration


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=18432,
                    help="independent sequences (frames per step) per GPU; 18432 = 72 per CU, a multiple of the 1 / 3 / 8 workgroups "
                         "per CU the matcher / search / GN kernels keep resident (no partial last wave); the five search + GN "
                         "rounds of a step each end in a 4-byte readback, which larger batches amortise")
    ap.add_argument("--keypoints", type=int, default=2000, help="keypoints per image (KITTI config: ~2000)")
    ap.add_argument("--moving", type=int, default=2000, help="local-map points per frame")
    ap.add_argument("--max-fixed", type=int, default=896, help="LDS sizing bound on stereo matches per frame")
    ap.add_argument("--unique", type=int, default=32, help="distinct synthetic frames generated on the host and tiled")
    ap.add_argument("--cpu-frames", type=int, default=1024, help="frames of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mode", choices=["frame", "closed-loop"], default="frame",
                    help="frame: the section-8(a) hot path on B independent frames per step (the headline); closed-loop: the "
                         "stateful per-frame loop incl. clipper and merger along the KITTI-00 trajectory (tools/bench_tracking.py)")
    ap.add_argument("--frames", type=int, default=60, help="closed-loop mode: frames per sequence")
    ap.add_argument("--all-iterations", action="store_true", help="disable the exact fixed-point early exit of the GN loop")
    ap.add_argument("--cpu-all-cores", type=int, default=-1,
                    help="worker processes of the all-core CPU figure (one independent sequence per core, SURVEY 8d); "
                         "-1 = min(host cores, 256), 0 = skip")
    ap.add_argument("--cpu-worker", type=int, default=0, help=argparse.SUPPRESS)  # internal: frames to time in a CPU-only child
    return ap.parse_args()


def make_unique_frames(cfg, n_unique, n_kp, n_moving, seed_base):
    from srrg2_proslam_amd import synthetic as syn
    frames = []
    for i in range(n_unique):
        rng = np.random.default_rng(seed_base + i)
        fr = syn.stereo_frame(rng, cfg, n_kp, visible_fraction=0.36)  # M ~ 0.35 N stereo matches
        T = syn.default_motion(rng, cfg)
        mp = syn.local_map(rng, cfg, fr, T, n_moving=n_moving, tracked_fraction=0.75)
        X0 = syn.perturb(rng, T, 0.05, 0.003)  # stand-in for the constant-velocity prediction error
        frames.append({"fr": fr, "T": T.astype(np.float32), "mp": mp, "X0": X0})
    return frames


def cpu_baseline(cfg, frames, n_frames):
    """single-threaded oracle on the same frames: stereo match + assemble + triangulate + align"""
    from oracle import binding as ob
    ob.lib()
    m = cfg["stereo_matcher"]
    sp = ob.StereoParams(m["maximum_descriptor_distance"], m["maximum_distance_ratio_to_second_best"],
                         m["minimum_matching_ratio"], m["maximum_disparity_pixels"], m["epipolar_line_thickness_pixels"])
    cam, tri, f = cfg["camera"], cfg["triangulator"], cfg["projective_finder"]
    tp = ob.TriangulatorParams(cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["fx"] * cam["baseline_m"],
                               tri["minimum_disparity_pixels"], tri["infinity_depth_meters"])
    proj = ob.Projector(cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["cols"], cam["rows"],
                        cfg["projector"]["range_min"], cfg["projector"]["range_max"])
    pp = ob.PcfParams(f["maximum_descriptor_distance"], f["maximum_distance_ratio_to_second_best"], f["minimum_matching_ratio"],
                      f["minimum_descriptor_distance"], f["descriptor_distance_step_size_pixels"],
                      f["maximum_search_radius_pixels"], f["minimum_search_radius_pixels"], f["search_radius_step_size_pixels"],
                      f["minimum_number_of_iterations"], f["maximum_estimate_change_norm_for_convergence"],
                      f["number_of_solver_iterations_per_projection"], f["search_type"], proj)
    al = cfg["aligner"]
    ap = ob.AlignerParams()
    ap.factor_type = al["factor_type"]
    ap.fx, ap.fy, ap.cx, ap.cy = cam["fx"], cam["fy"], cam["cx"], cam["cy"]
    ap.image_cols, ap.image_rows = cam["cols"], cam["rows"]
    ap.baseline_left_in_right_px[0] = -cam["fx"] * cam["baseline_m"]
    for i in range(3):
        ap.diagonal_info[i] = al["diagonal_info"][i]
    ap.chi_threshold, ap.enable_inverse_depth_weighting = al["chi_threshold"], al["enable_inverse_depth_weighting"]
    ap.damping, ap.max_iterations = al["damping"], al["max_iterations"]
    ap.min_num_inliers, ap.min_num_correspondences = al["min_num_inliers"], al["min_num_correspondences"]
    scales = [ob.info_scale_from_nopt(fr["mp"]["n_opt"]) for fr in frames]
    poses = []
    t0 = time.perf_counter()
    for k in range(n_frames):
        d = frames[k % len(frames)]
        fr, mp = d["fr"], d["mp"]
        corr, _ = ob.stereo_match(fr["uv_left"], fr["desc_left"], fr["uv_right"], fr["desc_right"], sp)
        uvuv, src = ob.stereo_assemble(fr["uv_left"], fr["uv_right"], corr)
        ob.triangulate(uvuv, tp)
        finder = ob.ProjectiveFinder(pp)
        finder.set_fixed(uvuv, fr["desc_left"][src])
        finder.set_moving(mp["xyz"], mp["desc"])
        ap.mean_disparity = ob.mean_disparity(uvuv)
        res, c = ob.align_frame(finder, ap, uvuv, mp["xyz"], scales[k % len(frames)], d["X0"])
        finder.close()
        if k < len(frames):
            poses.append((np.array(res.X, np.float32), c))
    dt = time.perf_counter() - t0
    return n_frames / dt, dt, poses



def cpu_worker(args):
    """child process of the all-core CPU leg: never touches the GPU; prints the frames/s of its own sequence"""
    from srrg2_proslam_amd import configs, synthetic as syn
    cfg = configs.get("kitti")
    frames = make_unique_frames(cfg, 4, args.keypoints, args.moving, syn.seed_for(1, 0) + 7000 + os.getpid() % 1000)
    fps, dt, _ = cpu_baseline(cfg, frames, args.cpu_worker)
    print(json.dumps({"fps": fps, "seconds": dt}))


def cpu_all_cores(args, n_workers, frames_each):
    """the oracle on every host core at once, one independent sequence per process"""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-worker", str(frames_each), "--keypoints", str(args.keypoints),
           "--moving", str(args.moving)]
    env = dict(os.environ)
    env["HIP_VISIBLE_DEVICES"] = ""
    env["OMP_NUM_THREADS"] = "1"
    t0 = time.perf_counter()
    procs = [subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env) for _ in range(n_workers)]
    total, ok = 0.0, 0
    for pr in procs:
        try:
            out, _ = pr.communicate(timeout=240)
            total += json.loads(out.decode().strip().splitlines()[-1])["fps"]
            ok += 1
        except Exception:
            pr.kill()
    return total, ok, time.perf_counter() - t0


def profile_evidence(frames_per_launch, keypoints):
    """HBM bytes per bench step and VALU-busy fractions of the kernels from the COMMITTED rocprofv3 passes of this same command
    (profiles/rNN/rocprof_summary.json, written by tools/profile_round.sh: FETCH_SIZE and WRITE_SIZE in separate passes).
    Per /opt/skills/guides/MI355X_MICROARCH.md the counters are KiB and FETCH_SIZE reports half of wide coalesced reads on
    gfx950, so it is doubled.  Evidence from a file, not a live measurement: every figure carries its source.  Kernels are
    matched by the exact prefix of their demangled name."""
    import glob
    prefixes = {"matcher": "prs::v5::stereo_match5_kernel<", "search": "prs::align_kernel<", "gn": "prs::gn_kernel<"}
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "rocprof_summary.json")), reverse=True):
        try:
            summ = json.load(open(path))
            if int(summ.get("frames_per_launch", 0)) != int(frames_per_launch) or int(summ.get("keypoints_per_image", 2000)) != int(keypoints):
                continue

            def pick(table, prefix):
                hits = [(k, v) for k, v in table.items() if k.replace("void ", "").startswith(prefix)]
                if len(hits) != 1:
                    raise KeyError(prefix)
                return hits[0]

            names = {k: pick(summ["pmc_fetch"], pre)[0] for k, pre in prefixes.items()}
            assert len(set(names.values())) == 3, names  # three distinct kernels
            steps = pick(summ["pmc_fetch"], prefixes["matcher"])[1]["FETCH_SIZE"]["launches"]
            out = {"source": os.path.relpath(path, ROOT), "unit": "bytes per bench step (2 * FETCH_SIZE + WRITE_SIZE, separate PMC passes)"}
            for k, pre in prefixes.items():
                f = pick(summ["pmc_fetch"], pre)[1]["FETCH_SIZE"]
                w = pick(summ["pmc_write"], pre)[1]["WRITE_SIZE"]
                out[k] = (2.0 * f["mean"] + w["mean"]) * 1024.0 * (f["launches"] / steps)
            return out
        except (OSError, KeyError, ValueError, AssertionError):
            continue
    return None


def closed_loop(args):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_tracking
    out = bench_tracking.run(batch=min(args.batch, 4096), frames=args.frames, keypoints=args.keypoints, check=1 if not args.no_cpu_baseline else 0)
    out.update({"steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "u64 popcount (Hamming) + f32 (projection, Jacobians, 6x6 normal equations) + f64 (landmark filters)", "data": "synthetic"})
    print(json.dumps(out))


def main():
    args = parse()
    if args.cpu_worker > 0:
        cpu_worker(args)
        return
    if args.mode == "closed-loop":
        closed_loop(args)
        return
    import torch

    from srrg2_proslam_amd import sharding
    rank, world, local_rank = sharding.rank_world()
    # test hook for 1-GPU boxes: all ranks share device 0 and the rendezvous runs on gloo, so that the
    # multi-rank control flow (barriers, MAX over ranks, rank-0 reporting) can be exercised without 8 GPUs
    shared = os.environ.get("PRS_BENCH_SHARE_GPU", "0") == "1"
    if shared:
        local_rank = 0
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))
    torch.cuda.set_device(local_rank)
    sharding.init_distributed("gloo" if shared else "nccl", local_rank)
    red_dev = "cpu" if shared else torch.device("cuda", local_rank)

    from srrg2_proslam_amd import _lib, configs, ops, synthetic as syn

    cfg = configs.get("kitti")
    B, N, NM = args.batch, args.keypoints, args.moving
    dev = torch.device("cuda", local_rank)

    # ---- synthetic KITTI-00-shaped inputs: distinct seeds per rank (independent sequences) -------
    uniq = make_unique_frames(cfg, args.unique, N, NM, syn.seed_for(1, 0) + 100000 * rank)
    sframes = ops.StereoFrames(local_rank, B, N, epilogue=True)
    aframes = ops.AlignFrames(local_rank, B, N, NM)
    stage = ops.StereoFrames(local_rank, len(uniq), N, epilogue=False)
    astage = ops.AlignFrames(local_rank, len(uniq), 1, NM)
    for u, d in enumerate(uniq):
        fr, mp = d["fr"], d["mp"]
        stage.upload(u, fr["uv_left"], fr["desc_left"], fr["uv_right"], fr["desc_right"])
        astage.upload(u, np.zeros((0, 4), np.float32), np.zeros((0, 32), np.uint8), mp["xyz"],
                      ops.info_scale_from_nopt(mp["n_opt"]), mp["desc"], d["X0"])
    idx = (torch.arange(B, device=dev) % len(uniq))
    for dst, src in ((sframes.left_kp, stage.left_kp), (sframes.right_kp, stage.right_kp),
                     (sframes.left_desc, stage.left_desc), (sframes.right_desc, stage.right_desc),
                     (sframes.n_left, stage.n_left), (sframes.n_right, stage.n_right),
                     (aframes.moving, astage.moving), (aframes.moving_desc, astage.moving_desc),
                     (aframes.n_moving, astage.n_moving)):
        dst.copy_(src.index_select(0, idx))
    X0_all = astage.X.index_select(0, idx).clone()
    state0 = aframes.state.clone()
    del stage, astage
    # the aligner consumes the matcher's device-resident outputs directly
    aframes.fixed, aframes.fixed_desc, aframes.n_fixed = sframes.fixed_uvuv, sframes.fixed_desc, sframes.n_fixed
    aframes.max_fixed = args.max_fixed  # stereo matches per frame are ~0.35 N; a frame above the bound fails loudly

    ctx = ops.Context(local_rank)
    stream = torch.cuda.Stream(device=dev)
    sp = ops.stereo_params(cfg["stereo_matcher"], cfg["camera"]["rows"])
    tp = ops.triangulator_params(cfg)
    pp = ops.pcf_params(cfg)
    ap = ops.aligner_params(cfg, stop_at_fixed_point=0 if args.all_iterations else 1)

    def step(ev=None, fresh_finders=True):
        # the motion-model guess for every frame of the batch; fresh finder objects (heaviest case: maximum search radius)
        # or the finders of the previous step (the reference's objects live across frames)
        if fresh_finders:
            aframes.state.copy_(state0, non_blocking=True)
        aframes.X.copy_(X0_all, non_blocking=True)
        aframes.n_corr.zero_()
        if ev:
            ev[0].record()
        ops.stereo_match_batch(ctx, sp, sframes, tp)
        if ev:
            ev[1].record()
        ops.align_batch(ctx, pp, ap, aframes)
        if ev:
            ev[2].record()

    def barrier():
        torch.cuda.synchronize()
        sharding.barrier()

    with torch.cuda.stream(stream):
        ctx.use_torch_stream()
        for _ in range(args.warmup):
            step()
        events = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
        ctx.enable_timing(True)  # HIP events around every launch of the aligner's two kernels, on this stream
        barrier()
        t0 = time.perf_counter()
        for k in range(args.steps):
            step(events[k])
        barrier()
        elapsed_local = time.perf_counter() - t0
        timing = ctx.align_timing()
        ctx.enable_timing(False)
        fps, elapsed = sharding.aggregate_throughput(B * args.steps, elapsed_local, red_dev)
        # results of the timed configuration, before the steady-state leg overwrites them
        n_match = sframes.n_matches.float().mean().item()
        n_fixed = sframes.n_fixed.float().mean().item()
        n_corr = aframes.n_corr.float().mean().item()
        results = aframes.result.cpu().numpy()
        Xg = aframes.X.cpu().numpy()
        corr_g = [aframes.corr_of(u) for u in range(min(len(uniq), B))]
        # ---- steady state: the same frames again with the finders carried over (radius / threshold schedule adapted) ----
        steady = None
        if rank == 0 and world == 1:
            for _ in range(6):  # radius 50 -> 10 px in steps of 10, threshold 25 -> 50
                step(fresh_finders=False)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(max(args.steps // 2, 1)):
                step(fresh_finders=False)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
            st = aframes.state_of(0)
            steady = {"value": B * max(args.steps // 2, 1) / dt, "unit": "frames/s", "ms_per_step": dt / max(args.steps // 2, 1) * 1e3,
                      "search_radius_pixels": int(st.search_radius_pixels), "descriptor_distance": float(st.descriptor_distance),
                      "aligner_correspondences_per_frame": aframes.n_corr.float().mean().item(),
                      "note": "finder objects live across steps like the reference's (correspondence_finder_projective_base_impl.cpp:277-287): "
                              "after six tracked frames the search radius has shrunk to its minimum and the descriptor threshold has grown"}

    # ---- per-kernel time (HIP events on the launch stream) + algorithmic bytes / flops ------------------
    ms_match = float(np.mean([e[0].elapsed_time(e[1]) for e in events]))
    ms_align = float(np.mean([e[1].elapsed_time(e[2]) for e in events]))
    ms_search = timing["search_ms"] / args.steps
    ms_gn = timing["gn_ms"] / args.steps
    res = [_lib.AlignResult.from_buffer_copy(results[b].tobytes()) for b in range(min(B, 256))]
    warn_min = min(_lib.AlignResult.from_buffer_copy(results[b].tobytes()).warnings for b in range(B))
    if warn_min < 0:
        raise SystemExit("align kernel reported error %d (raise --max-fixed?)" % warn_min)
    status_ok = float(np.mean([r.status for r in res]))
    it_exec = float(np.mean([r.iterations_executed for r in res]))
    # matcher 40 (N_L + N_R) + 12 M, triangulator 16 M + 13 M: the launch runs both (fused epilogue)
    bytes_match = 40.0 * (2 * N) + 12.0 * n_match + 29.0 * n_fixed
    bytes_search = 44.0 * NM + 40.0 * n_fixed + 64 + 12.0 * n_corr  # projective finder, per recompute
    searches = timing["search_launches"] / args.steps
    gbps_match = B * bytes_match / (ms_match * 1e-3) / 1e9
    gbps_search = B * bytes_search * searches / (ms_search * 1e-3) / 1e9 if ms_search > 0 else 0.0
    flops_gn = FLOP_PER_CORRESPONDENCE * n_corr * it_exec * B  # per step
    tflops_gn = flops_gn / (ms_gn * 1e-3) / 1e12 if ms_gn > 0 else 0.0
    evidence = profile_evidence(B, N)

    out = {
        "metric": "tracked frames/sec on KITTI-00 stereo (1241x376, ~2k kp); SE(3) vs ref",
        "value": fps,
        "unit": "frames/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u64 popcount (Hamming) + f32 (projection, Jacobians, 6x6 normal equations)",
        "data": "synthetic",
        "config": {
            "workload": "KITTI 00 stereo shaped (1241x376, FAST+ORB-like ~%d kp/image, 256-bit descriptors), "
                        "kitti.conf parameters: epipolar matcher + triangulator + projective circle finder + "
                        "stereo GN aligner, %d iterations/frame; fresh finder objects every step (maximum search radius)" % (N, cfg["aligner"]["max_iterations"]),
            "frames_per_step_per_gpu": B,
            "keypoints_per_image": N,
            "local_map_points": NM,
            "stereo_matches_per_frame": n_match,
            "aligner_correspondences_per_frame": n_corr,
            "parallelism": "independent sequences sharded one set per GPU, no collectives",
            "gn_iterations_executed_mean": it_exec,
            "aligner_success_fraction": status_ok,
        },
        "roofline": {
            "kernel": "gn_kernel<128, SLOTS, stereo> (reprojection-error Gauss-Newton rounds: factor linearisation, ordered H / b sums, "
                      "6x6 solve; the time-dominant kernel, all its launches of one step)",
            "bound": "valu",
            "achieved": tflops_gn,
            "peak": FP32_VECTOR_TFLOPS,
            "unit": "TFLOP/s",
            "frac": tflops_gn / FP32_VECTOR_TFLOPS,
            "traffic": evidence["gn"] if evidence else None,
            "traffic_unit": evidence["unit"] if evidence else None,
            "traffic_source": evidence["source"] if evidence else None,
            "ms_per_step": ms_gn,
            "launches_per_step": timing["gn_launches"] / args.steps,
            "algorithmic_flop_per_step": flops_gn,
            "flop_per_correspondence_iteration": FLOP_PER_CORRESPONDENCE,
            "note": "fp32 vector arithmetic, no MFMA: 6x6 normal equations are not a dense contraction; the kernel is bound by "
                    "instruction issue (VALU-busy fraction in profiles/), frac prices only the algorithmic flops",
        },
        "roofline_search": {
            "kernel": "align_kernel<512, split, circle> (projective search: projection, cell-grid circle search, Hamming, candidate filter)",
            "bound": "hbm", "achieved": gbps_search, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": gbps_search / HBM_PEAK_GBPS,
            "traffic": evidence["search"] if evidence else None, "traffic_source": evidence["source"] if evidence else None,
            "ms_per_step": ms_search, "launches_per_step": searches, "algorithmic_bytes_per_frame_and_launch": bytes_search, "frames_per_launch": B,
            "traffic_over_algorithmic": (evidence["search"] / (B * bytes_search * searches)) if evidence and searches > 0 else None,
        },
        "roofline_matcher": {
            "kernel": "stereo_match5_kernel<%d> (the kernel BASELINE.json north_star prices; matcher + fused adaptor / triangulator epilogue)" % (1 if N <= 1024 else 2),
            "bound": "hbm",
            "achieved": gbps_match,
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": gbps_match / HBM_PEAK_GBPS,
            "traffic": evidence["matcher"] if evidence else None,
            "traffic_unit": evidence["unit"] if evidence else None,
            "traffic_source": evidence["source"] if evidence else None,
            "ms_per_launch": ms_match,
            "algorithmic_bytes_per_frame": bytes_match,
            "frames_per_launch": B,
            "traffic_over_algorithmic": (evidence["matcher"] / (B * bytes_match)) if evidence else None,
        },
        "kernel_time_share": {"stereo_match5_kernel": ms_match / (ms_match + ms_align), "align_kernel (search)": ms_search / (ms_match + ms_align),
                              "gn_kernel": ms_gn / (ms_match + ms_align)},
    }
    if steady:
        out["steady_state"] = steady

    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.cpu_frames > 0:
        cpu_fps, cpu_s, poses = cpu_baseline(cfg, uniq, args.cpu_frames)
        # the same run doubles as an end-to-end check of the device pipeline on the bench inputs
        worst = 0.0
        exact_corr = True
        for u in range(min(len(uniq), B)):
            Xr, c = poses[u]
            worst = max(worst, float(np.linalg.norm(Xg[u] - Xr) / np.linalg.norm(Xr)))
            gc = corr_g[u]
            exact_corr = exact_corr and len(gc) == len(c) and bool(np.array_equal(gc["fixed_idx"], c["fixed_idx"])) and bool(np.array_equal(gc["moving_idx"], c["moving_idx"]))
        out["cpu_baseline"] = {
            "value": cpu_fps,
            "unit": "frames/s",
            "cores": 1,
            "kind": "port",
            "sample": "%d frames of the same synthetic workload (stereo match + assemble + triangulate + 100-iteration align), "
                      "single-threaded oracle (-O2, no fast-math), %.1f s; host has %d cores" % (args.cpu_frames, cpu_s, os.cpu_count() or 0),
        }
        out["parity_on_bench_inputs"] = {"pose_rel_frobenius_max": worst, "correspondences_bit_exact": exact_corr,
                                         "frames_checked": min(len(uniq), B)}
        n_workers = args.cpu_all_cores if args.cpu_all_cores >= 0 else min(os.cpu_count() or 1, 256)
        if n_workers > 0:
            total, ok, wall = cpu_all_cores(args, n_workers, 48)
            out["cpu_baseline_all_cores"] = {
                "value": total, "unit": "frames/s", "cores": ok, "kind": "port",
                "sample": "%d worker processes (one independent sequence each, 48 frames per worker) of the same oracle, "
                          "sum of the per-worker rates, %.1f s wall" % (ok, wall),
            }
    if rank == 0:
        print(json.dumps(out))
    sharding.shutdown()


if __name__ == "__main__":
    main()
