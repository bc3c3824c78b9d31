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

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline      achieved algorithmic HBM GB/s of the stereo matcher kernel (the kernel the
                north star prices) + the same for the time-dominant aligner kernel
  cpu_baseline  the single-threaded CPU restatement (oracle, "port") timed on a bounded sample
                of the same frames on this box's host cores (rank 0, N=1 only)
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

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


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
    ap.add_argument("--binned-matcher", action="store_true", help="use the second-generation (row, column-block) binned matcher kernel")
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


def valu_busy_fractions():
    """VALU-busy fraction of the aligner kernels from the committed PMC passes (profiles/rNN/align_pmc.json:
    SQ_ACTIVE_INST_VALU / (8 x SQ_BUSY_CYCLES)); None when the file is absent.  Evidence, not a live measurement."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01", "align_pmc.json")
    try:
        with open(path) as f:
            frac = json.load(f)["valu_busy_fraction"]
        return {k.split("prs::")[-1]: round(v, 3) for k, v in frac.items() if isinstance(v, float) and ("align_kernel" in k or "gn_kernel<128, 7, 4>" in k)}
    except (OSError, KeyError, ValueError):
        return None


def pmc_traffic_bytes(kernel_substrings, frames_per_launch, keypoints):
    """HBM bytes per bench step of the named kernels (= per launch for the matcher) from the committed rocprofv3 PMC passes of this same
    command (profiles/rNN/rocprof_summary.json, written by tools/profile_round.sh: FETCH_SIZE and
    WRITE_SIZE collected in separate passes).  Per /opt/skills/guides/MI355X_MICROARCH.md the
    counters are KiB and FETCH_SIZE reports half of wide coalesced reads on gfx950, so it is doubled.
    Returns (bytes, source) or (None, None) when no summary matches this configuration."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "rocprof_summary.json")), reverse=True):
        try:
            summ = json.load(open(path))
            if int(summ.get("frames_per_launch", 0)) != int(frames_per_launch) or int(summ.get("keypoints_per_image", 2000)) != int(keypoints):
                continue
            total = 0.0
            steps = [v["FETCH_SIZE"]["launches"] for k, v in summ["pmc_fetch"].items() if "stereo_match" in k][0]
            for sub in kernel_substrings:
                f = [v["FETCH_SIZE"] for k, v in summ["pmc_fetch"].items() if sub in k]
                w = [v["WRITE_SIZE"] for k, v in summ["pmc_write"].items() if sub in k]
                if not f or not w:
                    raise KeyError(sub)
                # launches of this kernel per bench step (the matcher is launched once per step)
                total += (2.0 * f[0]["mean"] + w[0]["mean"]) * 1024.0 * (f[0]["launches"] / steps)
            return total, os.path.relpath(path, ROOT)
        except (OSError, KeyError, ValueError):
            continue
    return None, None

def main():
    args = parse()
    if args.cpu_worker > 0:
        cpu_worker(args)
        return
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # test hook for 1-GPU boxes: all ranks share device 0 and the rendezvous runs on gloo, so that the
    # multi-rank control flow (barriers, MAX over ranks, rank-0 reporting) can be exercised without 8 GPUs
    shared = os.environ.get("PRS_BENCH_SHARE_GPU", "0") == "1"
    if shared:
        local_rank = 0
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if shared:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

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
    sp = ops.stereo_params(cfg["stereo_matcher"], cfg["camera"]["rows"], cfg["camera"]["cols"] if args.binned_matcher else 0)
    tp = ops.triangulator_params(cfg)
    pp = ops.pcf_params(cfg)
    ap = ops.aligner_params(cfg, stop_at_fixed_point=0 if args.all_iterations else 1)

    def step(ev=None):
        # fresh finder objects + motion-model guess for every frame of the batch
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
        if world > 1:
            dist.barrier()

    with torch.cuda.stream(stream):
        ctx.use_torch_stream()
        for _ in range(args.warmup):
            step()
        events = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
        barrier()
        t0 = time.perf_counter()
        for k in range(args.steps):
            step(events[k])
        barrier()
        elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if shared else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- per-kernel time (HIP events on the launch stream) + algorithmic bytes ------------------
    ms_match = float(np.mean([e[0].elapsed_time(e[1]) for e in events]))
    ms_align = float(np.mean([e[1].elapsed_time(e[2]) for e in events]))
    n_match = sframes.n_matches.float().mean().item()
    n_fixed = sframes.n_fixed.float().mean().item()
    n_corr = aframes.n_corr.float().mean().item()
    results = aframes.result.cpu().numpy()
    res0 = _lib.AlignResult.from_buffer_copy(results[0].tobytes())
    warn_min = min(_lib.AlignResult.from_buffer_copy(results[b].tobytes()).warnings for b in range(B))
    if warn_min < 0:
        raise SystemExit("align kernel reported error %d (raise --max-fixed?)" % warn_min)
    status_ok = float(np.mean([_lib.AlignResult.from_buffer_copy(results[b].tobytes()).status for b in range(min(B, 256))]))
    it_exec = float(np.mean([_lib.AlignResult.from_buffer_copy(results[b].tobytes()).iterations_executed for b in range(min(B, 256))]))
    # matcher 40 (N_L + N_R) + 12 M, triangulator 16 M + 13 M: the launch runs both (fused epilogue)
    bytes_match = 40.0 * (2 * N) + 12.0 * n_match + 29.0 * n_fixed
    bytes_align = 48.0 * NM + 48.0 * n_fixed + 64 + 12.0 * n_corr + 64  # single pass: map + fixed cloud + pose in, corr + pose out
    gbps_match = B * bytes_match / (ms_match * 1e-3) / 1e9
    gbps_align = B * bytes_align / (ms_align * 1e-3) / 1e9
    fps = world * B * args.steps / elapsed
    traffic_match, traffic_src = pmc_traffic_bytes(["stereo_match"], B, N)
    # all search + GN rounds of one step
    traffic_align, _ = pmc_traffic_bytes(["align_kernel", "gn_kernel"], B, N)

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
            "workload": "KITTI 00 stereo shaped (1241x376, FAST+BRIEF-like ~%d kp/image, 256-bit descriptors), "
                        "kitti.conf parameters: epipolar matcher + triangulator + projective circle finder + "
                        "stereo GN aligner, %d iterations/frame" % (N, cfg["aligner"]["max_iterations"]),
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
            "kernel": "stereo_match5_kernel<%d> (the kernel BASELINE.json north_star prices; matcher + fused adaptor / triangulator epilogue)" % (1 if N <= 1024 else 2),
            "bound": "hbm",
            "achieved": gbps_match,
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": gbps_match / HBM_PEAK_GBPS,
            "traffic": traffic_match,
            "traffic_unit": "bytes per launch (2*FETCH_SIZE + WRITE_SIZE, separate PMC passes)",
            "traffic_source": traffic_src,
            "ms_per_launch": ms_match,
            "algorithmic_bytes_per_frame": bytes_match,
            "frames_per_launch": B,
        },
        "roofline_align": {
            "kernel": "align_kernel<512> (projective search) + gn_kernel, all rounds of one step "
                      "(time-dominant; VALU-bound, not HBM-bound: see valu_busy_frac)",
            "valu_busy_frac": valu_busy_fractions(),
            "bound": "hbm",
            "achieved": gbps_align,
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": gbps_align / HBM_PEAK_GBPS,
            "traffic": traffic_align,
            "ms_per_launch": ms_align,
            "algorithmic_bytes_per_frame": bytes_align,
            "frames_per_launch": B,
        },
        "kernel_time_share": {"stereo_match5_kernel": ms_match / (ms_match + ms_align), "align_kernel": ms_align / (ms_match + ms_align)},
    }

    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.cpu_frames > 0:
        cpu_fps, cpu_s, poses = cpu_baseline(cfg, uniq, args.cpu_frames)
        # the same run doubles as an end-to-end check of the device pipeline on the bench inputs
        Xg = aframes.X.cpu().numpy()
        worst = 0.0
        exact_corr = True
        for u in range(min(len(uniq), B)):
            Xr, c = poses[u]
            worst = max(worst, float(np.linalg.norm(Xg[u] - Xr) / np.linalg.norm(Xr)))
            gc = aframes.corr_of(u)
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
    if world > 1:
        dist.destroy_process_group()
    del res0


if __name__ == "__main__":
    main()
