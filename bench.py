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
    python bench.py --gpus N ...                     (no launcher: starts N ranks itself, one per device, RCCL rendezvous on 127.0.0.1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line (contract in the task statement).  Besides the contract's fields:
  roofline          the time-dominant kernel of the step (a copy of one of the three below)
  roofline_search / roofline_gn / roofline_matcher
                    projective search kernel and stereo matcher kernel (the one BASELINE.json's north star prices): achieved
                    algorithmic HBM GB/s against 8 TB/s; Gauss-Newton kernel: achieved algorithmic fp32 TFLOP/s against the
                    157.3 TFLOP/s vector peak.  Times are HIP events on the launch stream, taken in a separate UNTIMED pass after
                    the headline loop (per launch of every round), `traffic` comes from the committed PMC passes of the same command
  steady_state      the same step with the finder objects carried over from frame to frame (the reference's adaptive
                    radius / threshold schedule live) instead of fresh finders
  other_configs     (+ kitti_real: the reference's own KITTI stereo pairs through the device extractor: REAL ORB descriptors, tiled)
                    BASELINE.json configs 3 and 4 and the reference's own keypoint budget: EuRoC-shaped stereo (752x480), TUM-shaped
                    RGB-D (640x480, depth-projective path), KITTI at 1000 keypoints per image; a few steps each, with a parity check
  closed_loop       the stateful per-frame loop (matcher -> clipper -> finder / aligner + motion prior -> pose update -> merger) along
                    the KITTI-00 trajectory, checked frame by frame against the same loop on the CPU checker
  from_images       pixels to poses on the device: extractor (both images) -> matcher -> finder + aligner on the reference's KITTI images, tiled
  latency_b1        ONE sequence, one frame at a time through the C++ plugin adapters (AoS clouds, gather timed) and through the bare
                    C-ABI with host pointers (PCIe, launches, synchronisation included), the CPU checker on the same frames beside it
  cpu_baseline      the single-threaded CPU restatement ("port") timed on a bounded sample of the same frames on this box's host
                    cores (rank 0, N=1 only); cpu_baseline_all_cores: one independent sequence per core
`--mode closed-loop` prints the closed-loop line alone; under torch.distributed.run it shards KITTI sequences 00-07 (BASELINE.json
config 5) over the ranks, longest first.
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
GPU_CLOCK_HZ = 2.4e9        # MI355X_MICROARCH.md: peak engine clock (s_memtime totals of one-workgroup kernels here: ~2.3e9)
N_CU = 256


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=55296,
                    help="independent sequences (frames per step) per GPU; 55296 = 216 per CU, a multiple of the 1 / 3 / 8 workgroups per CU the "
                         "matcher / search / GN kernels keep resident (no partial last wave); ~28 GB of the 288 GB.  Rounds 1-3 used 18432: the "
                         "fifth search / GN round of a step (the ~15 % of the frames whose finder latches one projection later) is a latency-bound "
                         "0.6 ms whatever the batch, so a three times larger batch is 4.6 % faster per frame (1.194 -> 1.250 M frames/s)")
    ap.add_argument("--keypoints", type=int, default=2000, help="keypoints per image (KITTI config: ~2000)")
    ap.add_argument("--moving", type=int, default=2000, help="local-map points per frame")
    ap.add_argument("--max-fixed", type=int, default=896, help="LDS sizing bound on stereo matches per frame")
    ap.add_argument("--unique", type=int, default=251,
                    help="distinct synthetic frames generated on the host and tiled over the batch (frame b = unique frame b mod this).  A prime: "
                         "workgroup b runs on XCD b mod 8 and with eight resident frames per CU, so a count that shares a factor with 8 or 256 "
                         "pins every unique frame to one XCD / CU slot (measured in round 4 with 8 unique frames: the frames that need a fifth "
                         "search round all sat on three XCDs, the other five idled, the GN time of the leg doubled)")
    ap.add_argument("--cpu-frames", type=int, default=1024, help="frames of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-steady-state", action="store_true", help="skip the steady-state leg (profiles of the headline workload alone)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the EuRoC / TUM / KITTI-1000 legs and the closed loop")
    ap.add_argument("--timing-steps", type=int, default=4, help="steps of the untimed per-kernel timing pass")
    ap.add_argument("--mode", choices=["frame", "closed-loop"], default="frame",
                    help="frame: the section-8(a) hot path on B independent frames per step (the headline); closed-loop: the "
                         "stateful per-frame loop incl. clipper and merger along the KITTI-00 trajectory (tools/bench_tracking.py)")
    ap.add_argument("--frames", type=int, default=60, help="closed-loop mode: frames per sequence (single GPU)")
    ap.add_argument("--frames-scale", type=float, default=0.02,
                    help="closed-loop mode under torch.distributed: fraction of every KITTI sequence's length that is run")
    ap.add_argument("--all-iterations", action="store_true", help="disable the exact fixed-point early exit of the GN loop")
    ap.add_argument("--cpu-all-cores", type=int, default=-1,
                    help="worker processes of the all-core CPU figure (one independent sequence per core, SURVEY 8d); "
                         "-1 = min(host cores, 256), 0 = skip")
    ap.add_argument("--cpu-worker", type=int, default=0, help=argparse.SUPPRESS)  # internal: frames to time in a CPU-only child
    ap.add_argument("--dry-run", action="store_true",
                    help="control-flow check without a device (CPU tests of the N > 1 path): ranks rendezvous over gloo, a step is a fixed sleep, "
                         "the line carries value 0 and data 'dry-run'")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------------------------
# synthetic frames of one configuration + the same frames on the CPU checker
# ---------------------------------------------------------------------------------------------------------------------
def make_unique_frames(cfg, n_unique, n_kp, n_moving, seed_base):
    from srrg2_proslam_amd import synthetic as syn
    frames = []
    stereo = cfg["aligner"]["factor_type"] == 4
    for i in range(n_unique):
        rng = np.random.default_rng(seed_base + i)
        if stereo:
            fr = syn.stereo_frame(rng, cfg, n_kp, visible_fraction=0.36)  # M ~ 0.35 N stereo matches
        else:
            fr = syn.rgbd_frame(rng, cfg, n_kp)  # (u, v, d) + descriptor: what the RGB-D adaptor hands over (out of scope, SURVEY #9)
        T = syn.default_motion(rng, cfg)
        mp = syn.local_map(rng, cfg, fr, T, n_moving=n_moving, tracked_fraction=0.75)
        X0 = syn.perturb(rng, T, 0.05, 0.003)  # stand-in for the constant-velocity prediction error
        frames.append({"fr": fr, "T": T.astype(np.float32), "mp": mp, "X0": X0})
    return frames


def oracle_params(cfg):
    from oracle import binding as ob
    cam, f, al = cfg["camera"], cfg["projective_finder"], cfg["aligner"]
    proj = ob.Projector(cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["cols"], cam["rows"],
                        cfg["projector"]["range_min"], cfg["projector"]["range_max"])
    pp = ob.PcfParams(f["maximum_descriptor_distance"], f["maximum_distance_ratio_to_second_best"], f["minimum_matching_ratio"],
                      f["minimum_descriptor_distance"], f["descriptor_distance_step_size_pixels"],
                      f["maximum_search_radius_pixels"], f["minimum_search_radius_pixels"], f["search_radius_step_size_pixels"],
                      f["minimum_number_of_iterations"], f["maximum_estimate_change_norm_for_convergence"],
                      f["number_of_solver_iterations_per_projection"], f["search_type"], proj)
    ap = ob.AlignerParams()
    ap.factor_type = al["factor_type"]
    ap.fx, ap.fy, ap.cx, ap.cy = cam["fx"], cam["fy"], cam["cx"], cam["cy"]
    ap.image_cols, ap.image_rows = cam["cols"], cam["rows"]
    ap.baseline_left_in_right_px[0] = -cam["fx"] * cam.get("baseline_m", 0.0)
    for i in range(3):
        ap.diagonal_info[i] = al["diagonal_info"][i]
    ap.chi_threshold, ap.enable_inverse_depth_weighting = al["chi_threshold"], al["enable_inverse_depth_weighting"]
    ap.damping, ap.max_iterations = al["damping"], al["max_iterations"]
    ap.min_num_inliers, ap.min_num_correspondences = al["min_num_inliers"], al["min_num_correspondences"]
    ap.enable_inlier_only_runs = int(al.get("enable_inlier_only_runs", 0))
    ap.keep_only_inlier_correspondences = int(al.get("keep_only_inlier_correspondences", 0))
    sp = tp = None
    if cfg.get("stereo_matcher"):
        m, tri = cfg["stereo_matcher"], cfg["triangulator"]
        sp = ob.StereoParams(m["maximum_descriptor_distance"], m["maximum_distance_ratio_to_second_best"],
                             m["minimum_matching_ratio"], m["maximum_disparity_pixels"], m["epipolar_line_thickness_pixels"])
        tp = ob.TriangulatorParams(cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["fx"] * cam["baseline_m"],
                                   tri["minimum_disparity_pixels"], tri["infinity_depth_meters"])
    return sp, tp, pp, ap


def cpu_baseline(cfg, frames, n_frames):
    """single-threaded CPU checker on the same frames: (stereo match + assemble + triangulate +) align -> (fps, seconds, poses)"""
    from oracle import binding as ob
    ob.lib()
    sp, tp, pp, ap = oracle_params(cfg)
    scales = [ob.info_scale_from_nopt(fr["mp"]["n_opt"]) for fr in frames]
    poses = []
    t0 = time.perf_counter()
    for k in range(n_frames):
        d = frames[k % len(frames)]
        fr, mp = d["fr"], d["mp"]
        if sp is not None:
            corr, _ = ob.stereo_match(fr["uv_left"], fr["desc_left"], fr["uv_right"], fr["desc_right"], sp)
            fixed, src = ob.stereo_assemble(fr["uv_left"], fr["uv_right"], corr)
            ob.triangulate(fixed, tp)
            fdesc = fr["desc_left"][src]
            ap.mean_disparity = ob.mean_disparity(fixed)
        else:
            fixed, fdesc = fr["fixed"], fr["desc_fixed"]
        finder = ob.ProjectiveFinder(pp)
        finder.set_fixed(fixed, fdesc)
        finder.set_moving(mp["xyz"], mp["desc"])
        res, c = ob.align_frame(finder, ap, fixed, mp["xyz"], scales[k % len(frames)], d["X0"])
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
    t_start = time.time()
    fps, dt, _ = cpu_baseline(cfg, frames, args.cpu_worker)
    print(json.dumps({"fps": fps, "seconds": dt, "frames": args.cpu_worker, "t_start": t_start, "t_end": time.time()}))


def usable_cores():
    """cores this process may actually use: the scheduler affinity capped by the cgroup CPU quota (a container that shows 256 cores
    can be limited to a handful; oversubscribing the quota only measures the throttle)"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                if q > 0:
                    quota = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().split()[0])
            break
        except (OSError, ValueError, IndexError):
            continue
    if quota:
        n = max(1, min(n, int(quota + 0.999)))
    return n, quota


def cpu_all_cores(args, n_workers, frames_each):
    """the CPU checker on every host core at once, one independent sequence per process"""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-worker", str(frames_each), "--keypoints", str(args.keypoints),
           "--moving", str(args.moving)]
    env = dict(os.environ)
    env["HIP_VISIBLE_DEVICES"] = ""
    env["OMP_NUM_THREADS"] = "1"
    t0 = time.perf_counter()
    procs = [subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env) for _ in range(n_workers)]
    frames, ok, starts, ends = 0, 0, [], []
    for pr in procs:
        try:
            out, _ = pr.communicate(timeout=240)
            r = json.loads(out.decode().strip().splitlines()[-1])
            frames += r["frames"]
            starts.append(r["t_start"])
            ends.append(r["t_end"])
            ok += 1
        except Exception:
            pr.kill()
    # wall clock of the region in which the workers compute (first start to last end: the start-up of the interpreters is outside,
    # the contention for memory bandwidth and the stragglers are inside)
    span = (max(ends) - min(starts)) if ok else 0.0
    return (frames / span if span > 0 else 0.0), ok, span, time.perf_counter() - t0


def profile_evidence(frames_per_launch, keypoints):
    """HBM bytes per bench step of the kernels from the COMMITTED rocprofv3 passes of this same command
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
            # wave-instructions per bench step by class (SQ_INSTS_*: the issue roofline of the matcher)
            try:
                out["insts"] = {k: {c.replace("SQ_INSTS_", "").lower(): v["mean"] * (v["launches"] / steps)
                                    for c, v in pick(summ["pmc_sq2"], pre)[1].items()} for k, pre in prefixes.items()}
            except (KeyError, TypeError):
                out["insts"] = None
            return out
        except (OSError, KeyError, ValueError, AssertionError):
            continue
    return None


# ---------------------------------------------------------------------------------------------------------------------
# B independent frames of one configuration resident in HBM; step() = one pass of the hot path over all of them
# ---------------------------------------------------------------------------------------------------------------------
class FrameWorkload:
    def __init__(self, cfg, device_index, batch, keypoints, moving, max_fixed, unique, seed, all_iterations=False, frames=None):
        import torch
        from srrg2_proslam_amd import ops
        self.cfg, self.B, self.N, self.NM = cfg, batch, keypoints, moving
        self.stereo = cfg["aligner"]["factor_type"] == 4
        self.dev = torch.device("cuda", device_index)
        # `frames`: ready-made frames (the real-image leg) instead of `unique` synthetic ones; keypoints / moving are then the strides
        self.uniq = frames if frames is not None else make_unique_frames(cfg, unique, keypoints, moving, seed)
        idx = (torch.arange(batch, device=self.dev) % len(self.uniq))
        self.aframes = ops.AlignFrames(device_index, batch, keypoints, moving)
        astage = ops.AlignFrames(device_index, len(self.uniq), 1 if self.stereo else keypoints, moving)
        if self.stereo:
            self.sframes = ops.StereoFrames(device_index, batch, keypoints, epilogue=True)
            stage = ops.StereoFrames(device_index, len(self.uniq), keypoints, epilogue=False)
        for u, d in enumerate(self.uniq):
            fr, mp = d["fr"], d["mp"]
            if self.stereo:
                stage.upload(u, fr["uv_left"], fr["desc_left"], fr["uv_right"], fr["desc_right"])
                astage.upload(u, np.zeros((0, 4), np.float32), np.zeros((0, 32), np.uint8), mp["xyz"],
                              ops.info_scale_from_nopt(mp["n_opt"]), mp["desc"], d["X0"])
            else:
                astage.upload(u, fr["fixed"], fr["desc_fixed"], mp["xyz"], ops.info_scale_from_nopt(mp["n_opt"]), mp["desc"], d["X0"])
        pairs = [(self.aframes.moving, astage.moving), (self.aframes.moving_desc, astage.moving_desc), (self.aframes.n_moving, astage.n_moving)]
        if self.stereo:
            sf = self.sframes
            pairs += [(sf.left_kp, stage.left_kp), (sf.right_kp, stage.right_kp), (sf.left_desc, stage.left_desc),
                      (sf.right_desc, stage.right_desc), (sf.n_left, stage.n_left), (sf.n_right, stage.n_right)]
        else:
            pairs += [(self.aframes.fixed, astage.fixed), (self.aframes.fixed_desc, astage.fixed_desc), (self.aframes.n_fixed, astage.n_fixed)]
        for dst, src in pairs:
            dst.copy_(src.index_select(0, idx))
        self.X0_all = astage.X.index_select(0, idx).clone()
        self.state0 = self.aframes.state.clone()
        if self.stereo:  # the aligner consumes the matcher's device-resident outputs directly
            self.aframes.fixed, self.aframes.fixed_desc, self.aframes.n_fixed = self.sframes.fixed_uvuv, self.sframes.fixed_desc, self.sframes.n_fixed
        self.aframes.max_fixed = max_fixed  # a frame above the bound fails loudly
        self.ctx = ops.Context(device_index)
        self.sp = ops.stereo_params(cfg["stereo_matcher"], cfg["camera"]["rows"]) if self.stereo else None
        self.tp = ops.triangulator_params(cfg) if self.stereo else None
        self.pp = ops.pcf_params(cfg)
        self.ap = ops.aligner_params(cfg, stop_at_fixed_point=0 if all_iterations else 1)
        self.ops = ops

    def step(self, ev=None, fresh_finders=True):
        # the motion-model guess for every frame of the batch; fresh finder objects (heaviest case: maximum search radius)
        # or the finders of the previous step (the reference's objects live across frames)
        a = self.aframes
        if fresh_finders:
            a.state.copy_(self.state0, non_blocking=True)
        a.X.copy_(self.X0_all, non_blocking=True)
        a.n_corr.zero_()
        if ev:
            ev[0].record()
        if self.stereo:
            self.ops.stereo_match_batch(self.ctx, self.sp, self.sframes, self.tp)
        if ev:
            ev[1].record()
        self.ops.align_batch(self.ctx, self.pp, self.ap, a)
        if ev:
            ev[2].record()

    def kernel_times(self, n_steps, fresh_finders=True):
        """untimed pass: HIP events on the launch stream around the matcher launch and around every launch of the aligner's two
        kernels -> ms per step of each kernel and ms of every round's launches"""
        import torch
        events = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(n_steps)]
        self.ctx.enable_timing(True)
        for k in range(n_steps):
            self.step(events[k], fresh_finders)
        torch.cuda.synchronize()
        t = self.ctx.align_timing()
        s_round, g_round, nb = self.ctx.align_round_timing()
        self.ctx.enable_timing(False)
        nb = max(nb, 1)
        rounds = max(i + 1 for i in range(16) if s_round[i] > 0 or g_round[i] > 0 or i == 0)
        return {"matcher_ms": float(np.mean([e[0].elapsed_time(e[1]) for e in events])) if self.stereo else 0.0,
                "align_ms": float(np.mean([e[1].elapsed_time(e[2]) for e in events])),
                "search_ms": t["search_ms"] / n_steps, "gn_ms": t["gn_ms"] / n_steps,
                "search_launches": t["search_launches"] / n_steps, "gn_launches": t["gn_launches"] / n_steps,
                "search_ms_by_round": [s_round[i] / nb for i in range(rounds)], "gn_ms_by_round": [g_round[i] / nb for i in range(rounds)]}

    def snapshot(self):
        a = self.aframes
        n = min(len(self.uniq), self.B)
        from srrg2_proslam_amd import _lib
        off = _lib.PcfState.num_recomputes.offset
        recomputes = a.state[:, off: off + 4].contiguous().cpu().numpy().view(np.int32).reshape(-1).astype(np.float64).mean()
        out = {"n_corr": a.n_corr.float().mean().item(), "results": a.result.cpu().numpy(), "X": a.X[:n].cpu().numpy(),
               "corr": [a.corr_of(u) for u in range(n)], "n_fixed": a.n_fixed.float().mean().item(),
               "searches_per_frame": float(recomputes)}  # projective searches a frame of this step went through (finder state starts at 0)
        if self.stereo:
            out["n_match"] = self.sframes.n_matches.float().mean().item()
        return out

    def check(self, snap):
        """errors the kernels report per frame are loud; -> (success fraction, mean executed iterations)"""
        from srrg2_proslam_amd import _lib
        res = [_lib.AlignResult.from_buffer_copy(snap["results"][b].tobytes()) for b in range(self.B)]
        worst = min(r.warnings for r in res)
        if worst < 0:
            raise SystemExit("align kernel reported error %d (raise --max-fixed?)" % worst)
        head = res[: min(self.B, 256)]
        return float(np.mean([r.status for r in head])), float(np.mean([r.iterations_executed for r in head]))

    def parity(self, snap, poses):
        worst, exact = 0.0, True
        for u, (Xr, c) in enumerate(poses[: len(snap["corr"])]):
            worst = max(worst, float(np.linalg.norm(snap["X"][u] - Xr) / np.linalg.norm(Xr)))
            gc = snap["corr"][u]
            exact = exact and len(gc) == len(c) and bool(np.array_equal(gc["fixed_idx"], c["fixed_idx"])) and bool(np.array_equal(gc["moving_idx"], c["moving_idx"]))
        return {"pose_rel_frobenius_max": worst, "correspondences_bit_exact": exact, "frames_checked": min(len(poses), len(snap["corr"]))}

    def close(self):
        self.ctx.close()


def small_config_leg(name, cfg, keypoints, moving, max_fixed, batch, device_index, seed, steps=3):
    """one of the other BASELINE configurations: a few steps, per-kernel times, parity of four frames against the CPU checker"""
    import torch
    w = FrameWorkload(cfg, device_index, batch, keypoints, moving, max_fixed, 13, seed)  # a prime: see --unique
    stream = torch.cuda.Stream(device=w.dev)
    with torch.cuda.stream(stream):
        w.ctx.use_torch_stream()
        w.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            w.step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        snap = w.snapshot()
        ok, it_exec = w.check(snap)
        kt = w.kernel_times(2)
    _, _, poses = cpu_baseline(cfg, w.uniq[:4], 4)
    cam = cfg["camera"]
    out = {"workload": "%s: %dx%d, %d keypoints per image, %d local-map points, %s, %s.conf finder / aligner parameters" % (
               name, cam["cols"], cam["rows"], keypoints, moving,
               "stereo matcher + triangulator + projective finder + stereo GN aligner" if w.stereo else
               "depth-projective path: (u, v, d) measurements from the RGB-D adaptor (host-side input prep, out of scope) -> projective finder + depth GN aligner "
               "with inlier-only runs", cfg["name"]),
           "value": batch * steps / dt, "unit": "frames/s", "ms_per_step": dt / steps * 1e3, "frames_per_step": batch, "steps": steps,
           "ms_per_kernel": {"stereo_match5_kernel": kt["matcher_ms"], "align_kernel (search)": kt["search_ms"], "gn_kernel": kt["gn_ms"]},
           "search_ms_by_round": kt["search_ms_by_round"], "gn_ms_by_round": kt["gn_ms_by_round"],
           "fixed_points_per_frame": snap["n_fixed"], "aligner_correspondences_per_frame": snap["n_corr"],
           "aligner_success_fraction": ok, "gn_iterations_executed_mean": it_exec, "parity": w.parity(snap, poses)}
    w.close()
    del w
    torch.cuda.empty_cache()
    return out


# ---------------------------------------------------------------------------------------------------------------------
# the rows SURVEY 8 marks "next" (f1-f4), each with its own roofline object in the driver-run line
# ---------------------------------------------------------------------------------------------------------------------
I8_MFMA_TOPS = 5000.0  # MI355X_MICROARCH.md: I8 MFMA = 2 x the dense BF16 rate (~2.5 PF)


def _safe_cpu(fn):
    """a row's CPU baseline (the checker in oracle/, timed on this box's host: a reported figure, never the product path); a failure
    to produce it must not take the row's GPU figures with it"""
    try:
        return fn()
    except Exception as exc:  # noqa: BLE001
        return {"error": "%s: %s" % (type(exc).__name__, exc)}


def _cpu_port_merge(NM=704, NS=2000, history=6, repeats=3):
    """the CPU checker's MergerProjective_::compute + pose-based smoother on one map of the f1 shape (tools/bench_merge.py: NS landmarks
    with `history` earlier measurements seen from the identity pose, a frame of NM measurements that re-observe them)"""
    from oracle import binding as ob
    from oracle import binding_mapping as om
    from srrg2_proslam_amd import configs
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from tests.test_oracle_mapping import merger_params as oracle_merger_params
    ob.lib()
    cfg = configs.get("kitti")
    cam = cfg["camera"]
    K = (cam["fx"], cam["fy"], cam["cx"], cam["cy"])
    est = om.estimator_params(om.EST_SMOOTHER, 4, K, baseline_px=(cam["fx"] * cam["baseline_m"], 0.0), max_dist2=100.0, min_cov=0.01, max_cov_norm2=0.25,
                              max_iterations=100, chi2_delta=1e-6, max_reprojection2=100.0, min_measurements=3)
    p = oracle_merger_params(cfg, om.MERGER_STEREO_TRIANGULATION, est, enable_binning=1, row_bins=20, col_bins=60, max_appearance=100.0, target_merges=10 ** 6)
    rng = np.random.default_rng(3)
    z = (rng.random(NS) * 40 + 6).astype(np.float32)
    x = ((rng.random(NS) - 0.5) * z).astype(np.float32)
    y = ((rng.random(NS) - 0.5) * 0.3 * z).astype(np.float32)
    uL, vL = cam["fx"] * x / z + cam["cx"], cam["fy"] * y / z + cam["cy"]
    uR = uL - cam["fx"] * cam["baseline_m"] / z
    base = om.Map(NS + NM, 8)
    base.n_points = NS
    base.coords[:NS, 0], base.coords[:NS, 1], base.coords[:NS, 2] = x, y, z
    base.state[:NS] = base.coords[:NS]
    base.covariance[:NS] = np.eye(3, dtype=np.float32).reshape(9)
    for h in range(history):
        base.meas["point_in_image"][:NS, h, 0], base.meas["point_in_image"][:NS, h, 1], base.meas["point_in_image"][:NS, h, 2] = uL, vL, uR
        base.meas["point_in_camera"][:NS, h, 0], base.meas["point_in_camera"][:NS, h, 1], base.meas["point_in_camera"][:NS, h, 2] = x, y, z
    base.n_meas[:NS] = history
    base.n_opt[:NS] = history
    poses = om.pose_table(16)
    for f in range(16):
        om.set_pose(poses, f, np.eye(4, dtype=np.float32))
    n_c = min(NM, NS)
    meas = np.stack([uL[:n_c], vL[:n_c], uR[:n_c], vL[:n_c]], axis=1).astype(np.float32)
    desc = np.zeros((n_c, 32), np.uint8)
    corr = np.zeros(n_c, dtype=ob.CORR_DTYPE)
    corr["fixed_idx"], corr["moving_idx"], corr["response"] = np.arange(n_c), np.arange(n_c), 10.0
    I4 = np.eye(4, dtype=np.float32)
    best, merged = None, 0
    for _ in range(repeats):
        m = base.copy()
        t0 = time.perf_counter()
        rc, res = om.merge(p, I4, I4, poses, 1, m, meas, desc, corr)
        dt = time.perf_counter() - t0
        if rc < 0:
            raise RuntimeError("orc_merge error %d" % rc)
        best, merged = (dt if best is None or dt < best else best), int(res.n_merged)
    return {"value": 1.0 / best, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "one frame of %d measurements into a %d-point map (%d landmarks merged, %d earlier measurements each), best of %d runs: %.2f ms" % (
                NM, NS, merged, history, repeats, 1e3 * best)}


def _cpu_port_clip(n=16000, repeats=5):
    """the CPU checker's SceneClipperProjective3D::compute on one local map of the f2 shape (points in a box around the frustum)"""
    from oracle import binding as ob
    from srrg2_proslam_amd import configs
    cam = configs.get("kitti")["camera"]
    proj = ob.Projector(cam["fx"], cam["fy"], cam["cx"], cam["cy"], int(cam["cols"]), int(cam["rows"]), 0.1, 1000.0)
    rng = np.random.default_rng(1)
    xyzw = rng.random((n, 4), dtype=np.float32)
    xyzw[:, 0] = (xyzw[:, 0] - 0.5) * 60
    xyzw[:, 1] = (xyzw[:, 1] - 0.5) * 16
    xyzw[:, 2] = xyzw[:, 2] * 50 + 2
    desc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    I4 = np.eye(4, dtype=np.float32)
    best = None
    for _ in range(repeats):
        t0 = time.perf_counter()
        kept = ob.scene_clip(proj, I4, I4, xyzw, desc)[0]
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    return {"value": 1.0 / best, "unit": "local maps/s", "cores": 1, "kind": "port",
            "sample": "one local map of %d points with descriptors (%d kept), best of %d runs: %.3f ms" % (n, len(kept), repeats, 1e3 * best)}


def _cpu_port_features(repeats=3):
    """the CPU checker's extractor (FAST-9 + NMS, 3 x 3 binned selection in libstdc++'s order, ORB-256) on the reference's real KITTI images"""
    from oracle import binding_features as of
    z = np.load(os.path.join(ROOT, "tests", "golden", "ref_kitti.npz"))
    images = [im for im in z["city_left"]][:4]
    p = of.extractor_params(selection_order=of.SELECT_LIBSTDCXX)
    best = None
    for _ in range(repeats):
        t0 = time.perf_counter()
        n = [len(of.extract_features(p, im)[0]) for im in images]
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    return {"value": len(images) / best, "unit": "images/s", "cores": 1, "kind": "port",
            "sample": "%d of the reference's KITTI images (1241 x 376, %.0f features each), best of %d runs: %.1f ms per image" % (
                len(images), float(np.mean(n)), repeats, 1e3 * best / len(images))}


def f_rows_leg():
    """f1 merger + landmark estimators, f2 scene clipper, f3 feature extraction, f4 brute-force matcher: one launch shape each (the
    shapes of tools/bench_{merge,clip,features,bruteforce}.py), inputs resident in HBM, time = HIP events around the operator's
    launches on the launch stream, algorithmic bytes / operations as defined beside each figure."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_bruteforce
    import bench_clip
    import bench_features
    import bench_merge
    from srrg2_proslam_amd import ops
    out = {}

    def hbm(achieved_gbps, extra):
        d = {"bound": "hbm", "achieved": achieved_gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved_gbps / HBM_PEAK_GBPS, "traffic": None}
        d.update(extra)
        return d

    try:
        rows = [bench_merge.run(2048, 704, 2000, est, name, quiet=True) for est, name in
                ((ops.EST_SMOOTHER, "pose-based smoother (kitti.conf)"), (ops.EST_WEIGHTED_MEAN, "weighted mean"), (ops.EST_EKF, "stereo EKF (f64)"))]
        out["roofline_f1"] = hbm(rows[0]["gbps"], {
            "kernel": "merge_kernel + smoother_kernel + tail (MergerCorrespondenceProjective + LandmarkEstimatorPoseBasedSmoother, kitti.conf; one frame of 704 "
                      "measurements into a 2000-point map per launch and map)",
            "ms_per_launch": rows[0]["ms_per_launch"], "frames_per_s": rows[0]["frames_per_s"], "algorithmic_bytes_per_launch": rows[0]["algorithmic_bytes_per_launch"],
            "algorithmic_bytes": "48 B per measurement + 12 B per correspondence + 2 x (112 B + 28 B per kept measurement) per merged landmark + 140 B per added one",
            "estimators": rows, "cpu_baseline": _safe_cpu(_cpu_port_merge),
            "note": "not bandwidth-bound: the smoother runs up to 100 Gauss-Newton iterations per merged landmark on its measurement history "
                    "(one lane per landmark, serial), the EKF a 3 x 3 f64 update; frac prices the landmark rows a merge touches"})
    except (SystemExit, RuntimeError, AssertionError) as exc:
        out["roofline_f1"] = {"error": str(exc)}
    try:
        c = bench_clip.run(2048, 2000, True, quiet=True)
        c8 = bench_clip.run(2048, 16000, True, quiet=True)
        out["roofline_f2"] = hbm(c8["gbps"], {
            "kernel": "scene_clip_kernel (SceneClipperProjective3D::compute: transform, pinhole projection, frustum test, ordered compaction with descriptors)",
            "ms_per_launch": c8["ms_per_launch"], "algorithmic_bytes_per_launch": c8["algorithmic_bytes_per_launch"],
            "algorithmic_bytes": "48 B per scene point read (xyzw + 256-bit row) + 52 B per kept point written (+ global index) + 64 B per scene",
            "shapes": [c, c8], "cpu_baseline": _safe_cpu(_cpu_port_clip),
            "note": "2048 local maps of 16000 points per launch (the headline figure) and of 2000 points (a KITTI local map; launch-latency share larger)"})
    except (SystemExit, RuntimeError, AssertionError) as exc:
        out["roofline_f2"] = {"error": str(exc)}
    try:
        f = bench_features.run(4096, "kitti", "libstdcxx", quiet=True)
        out["roofline_f3"] = hbm(f["gbps"], {
            "kernel": "fast_blur_kernel + raster_order_kernel + select_describe_kernel + describe_kernel (FAST-9 + NMS, 3 x 3 binned selection in the reference's "
                      "std::sort order, 7x7 Gaussian on v_mfma_i32_16x16x32_i8, ORB-256)",
            "ms_per_launch": f["ms_per_launch"], "images_per_s": f["images_per_s"], "algorithmic_bytes_per_launch": f["algorithmic_bytes_per_launch"],
            "algorithmic_bytes": "the 8-bit image once (1241 x 376) + 44 B per kept feature (keypoint, descriptor, counters)",
            "detail": f, "cpu_baseline": _safe_cpu(_cpu_port_features),
            "note": "bound by vector issue, not by HBM: the tile kernel retires one vector instruction per cycle and CU (profiles/r05/features_pmc.txt); "
                    "frac prices the pixels and the features only"})
    except (SystemExit, RuntimeError, AssertionError, OSError) as exc:
        out["roofline_f3"] = {"error": str(exc)}
    try:
        from srrg2_proslam_amd import ops as _ops
        # the default mode (PRS_BF_DENSE_MATRIX_WHEN_FULL): these batches fill the chip and take the fused matrix-core shape
        b = bench_bruteforce.run(1024, 2000, 50.0, quiet=True)
        b1 = bench_bruteforce.run(1024, 1000, 50.0, quiet=True)
        p = bench_bruteforce.run(1024, 2000, 50.0, quiet=True, dense=_ops.BF_DENSE_POPCOUNT)
        p1 = bench_bruteforce.run(1024, 1000, 50.0, quiet=True, dense=_ops.BF_DENSE_POPCOUNT)
        real = {}
        for name, mode in (("default", None), ("popcount", _ops.BF_DENSE_POPCOUNT)):
            real[name] = [bench_bruteforce.run_real(1024, 50.0, quiet=True, target=1000, capacity=65536, dense=mode),
                          bench_bruteforce.run_real(256, 50.0, quiet=True, target=2000, capacity=262144, dense=mode)]
        real["speedup_over_popcount"] = [real["popcount"][i]["ms_per_launch"] / real["default"][i]["ms_per_launch"] for i in range(2)]
        real["one_cloud_pair"] = [bench_bruteforce.run_real(1, 50.0, quiet=True, target=1000, capacity=65536),
                                  bench_bruteforce.run_real(1, 50.0, quiet=True, target=2000, capacity=262144)]
        real["cpu_baseline"] = _safe_cpu(bench_bruteforce.cpu_port_real)
        tops = b["pairs_per_s"] * 512.0 / 1e12
        out["roofline_f4"] = {
            "kernel": "bruteforce dense phase + registration (CorrespondenceFinderDescriptorBasedBruteforce::compute: all N_f x N_m Hamming distances, pools by "
                      "distance, uniqueness, Lowe's ratio on both sides)",
            "bound": "mfma", "achieved": tops, "peak": I8_MFMA_TOPS, "unit": "TOP/s", "frac": tops / I8_MFMA_TOPS, "traffic": None,
            "mode": "the default dense phase (fused shape, distances from v_mfma_i32_16x16x64_i8) on uniform random rows with one true partner per point",
            "operations_per_descriptor_pair": 512, "descriptor_pairs_per_s": b["pairs_per_s"], "ms_per_launch": b["ms_per_launch"],
            "shapes": [b, b1],
            "popcount_kernels": {"shapes": [p, p1], "speedup_of_default": [p["ms_per_launch"] / b["ms_per_launch"], p1["ms_per_launch"] / b1["ms_per_launch"]]},
            "real_descriptors": real,
            "note": "hamming(a, b) = pop(a) + pop(b) - 2 a.b: the binary dot product of 256-bit rows is 256 multiply-adds = 512 operations per pair, "
                    "priced against the dense I8 MFMA rate (2 x BF16); includes the registration phases of every cloud pair.  Uniform random rows are "
                    "128 +- 8 bits apart (one candidate per fixed point); real descriptors (KITTI stereo pairs, our extractor) 125 +- 33 bits, 1.6 % of "
                    "the pairs within 50 bits -- there the matcher's time is candidate handling and registration, not scoring: real_descriptors holds "
                    "the same modes on real rows, and one cloud pair at a time (the popcount kernels spread over several workgroups)"}
    except (SystemExit, RuntimeError, AssertionError) as exc:
        out["roofline_f4"] = {"error": str(exc)}
    return out


# ---------------------------------------------------------------------------------------------------------------------
# a13_alt: the headline step under the OTHER readings of the external srrg2_solver arithmetic
# ---------------------------------------------------------------------------------------------------------------------
A13_VARIANTS = (
    ("shipped", {}, {}),
    ("damping_identity", {"damping_form": 1}, {"damping_form": 1}),                                # H + lambda I
    ("kernel_tau_over_chi", {"kernel_weight_form": 1}, {"kernel_form": 1}),                        # Omega * tau / chi
    ("translation_weight_clamp", {"translation_weight_form": 1}, {"idw_form": 1}),                 # clamp(d / mean, 0.01, 1)
    ("identity_and_tau_over_chi", {"damping_form": 1, "kernel_weight_form": 1}, {"damping_form": 1, "kernel_form": 1}),
)


def a13_alt_leg(cfg, frames, keypoints, moving, max_fixed, batch, device_index, steps=3, check=4):
    """What the headline costs under each reading of the un-vendored solver arithmetic that prs_aligner_params exposes
    (kernel_weight_form, damping_form, translation_weight_form; include/proslam_hip.h): the same frames, the same step, `batch`
    frames per step; frames/s, searches per frame, per-kernel ms, iterations, and parity of `check` frames against the CPU checker
    switched to the SAME reading (orc_set_variant).  The damping form sets how far a step moves the pose, hence how many projective
    searches and Gauss-Newton rounds a frame needs: a maintainer who holds srrg2_solver reads the price of either reading here."""
    import copy
    import torch
    from oracle import binding as ob
    rows = {}
    for name, dev_forms, orc_forms in A13_VARIANTS:
        vcfg = copy.deepcopy(cfg)
        vcfg["aligner"].update(dev_forms)
        w = FrameWorkload(vcfg, device_index, batch, keypoints, moving, max_fixed, len(frames), 0, frames=frames)
        stream = torch.cuda.Stream(device=w.dev)
        with torch.cuda.stream(stream):
            w.ctx.use_torch_stream()
            w.step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                w.step()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            snap = w.snapshot()
            ok, it_exec = w.check(snap)
            kt = w.kernel_times(2)
        try:
            ob.set_variant(**orc_forms)
            _, _, poses = cpu_baseline(vcfg, frames[:check], check)
        finally:
            ob.set_variant()  # back to the shipped definition
        par = w.parity(snap, poses)
        Xs = snap["X"][:check]
        rows[name] = {"forms": dev_forms or {"kernel_weight_form": 0, "damping_form": 0, "translation_weight_form": 0},
                      "value": batch * steps / dt, "unit": "frames/s", "ms_per_step": dt / steps * 1e3,
                      "searches_per_frame": snap["searches_per_frame"], "gn_iterations_executed_mean": it_exec,
                      "ms_per_kernel": {"stereo_match5_kernel": kt["matcher_ms"], "align_kernel (search)": kt["search_ms"], "gn_kernel": kt["gn_ms"]},
                      "search_ms_by_round": kt["search_ms_by_round"], "gn_ms_by_round": kt["gn_ms_by_round"],
                      "aligner_correspondences_per_frame": snap["n_corr"], "aligner_success_fraction": ok,
                      "parity_vs_checker_same_reading": par,
                      "_X": Xs}
        w.close()
        del w
        torch.cuda.empty_cache()
    # how far the poses of the readings lie apart (relative Frobenius against the shipped family's pose of the same frame)
    X0 = rows["shipped"]["_X"]
    for name, r in rows.items():
        X = r.pop("_X")
        r["pose_rel_frobenius_vs_shipped_max"] = float(max(np.linalg.norm(X[i] - X0[i]) / np.linalg.norm(X0[i]) for i in range(len(X0))))
    return {"frames_per_step": batch, "steps": steps, "readings": rows,
            "note": "row a13 of SURVEY 8 is external to the reference tree (srrg2_solver): the shipped family is the one that meets all 17 pose "
                    "bounds of the reference's own tests (tests/test_sweep_a13.py); every other reading is a run-time field of prs_aligner_params, "
                    "bit-exact against the checker switched the same way (tests/test_a13_forms_gpu.py); the fast Gauss-Newton instantiations carry "
                    "the shipped forms at compile time, the others run the generic one"}


# ---------------------------------------------------------------------------------------------------------------------
# tolerance_exit: the opt-in early exit (prs_aligner_params.step_norm_exit) -- LESS work than the reference, never the headline
# ---------------------------------------------------------------------------------------------------------------------
def tolerance_exit_leg(cfg, frames, keypoints, moving, max_fixed, batch, device_index, bound=1e-5, steps=3, check=8):
    """The same step with step_norm_exit = `bound`: a frame leaves the Gauss-Newton loop once its finder has latched and |dx| < bound.
    The reference runs every iteration (kitti.conf:1006-1009 gives MultiAligner3DQR no termination criterion), so this is labelled
    as doing less work; correspondences are checked bit-exact and poses <= 1e-4 against the 100-iteration CPU checker."""
    import copy
    import torch
    vcfg = copy.deepcopy(cfg)
    vcfg["aligner"]["step_norm_exit"] = bound
    w = FrameWorkload(vcfg, device_index, batch, keypoints, moving, max_fixed, len(frames), 0, frames=frames)
    stream = torch.cuda.Stream(device=w.dev)
    with torch.cuda.stream(stream):
        w.ctx.use_torch_stream()
        w.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            w.step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        snap = w.snapshot()
        ok, it_exec = w.check(snap)
        kt = w.kernel_times(2)
    _, _, poses = cpu_baseline(cfg, frames[:check], check)  # the checker runs ALL iterations
    par = w.parity(snap, poses)
    w.close()
    del w
    torch.cuda.empty_cache()
    return {"does_less_work_than_the_reference": True, "step_norm_exit": bound, "frames_per_step": batch, "value": batch * steps / dt, "unit": "frames/s",
            "ms_per_step": dt / steps * 1e3, "gn_iterations_executed_mean": it_exec,
            "ms_per_kernel": {"stereo_match5_kernel": kt["matcher_ms"], "align_kernel (search)": kt["search_ms"], "gn_kernel": kt["gn_ms"]},
            "aligner_success_fraction": ok, "parity_vs_100_iteration_checker": par,
            "note": "opt-in (0 = off is the default and the headline): the correspondence vector is the full run's, the pose is within the stated "
                    "1e-4 of it; reported beside the exact number, not instead of it"}


# ---------------------------------------------------------------------------------------------------------------------
# small batches: what a caller with few sequences gets from the device-resident batch API (between latency_b1 and the headline)
# ---------------------------------------------------------------------------------------------------------------------
def small_batch_curve(cfg, frames, keypoints, moving, max_fixed, device_index, batches=(1, 2, 4, 8, 16, 64, 256, 2048), steps=8):
    """frames per second and ms per step of the headline step (matcher + epilogue, search / GN rounds) at B = 1 .. 2048 frames per
    step, inputs resident in HBM, device pointers (prs_stereo_match_batch + prs_align_batch_run), fresh finders every step"""
    import torch
    curve = []
    for b in batches:
        w = FrameWorkload(cfg, device_index, b, keypoints, moving, max_fixed, len(frames), 0, frames=frames)
        stream = torch.cuda.Stream(device=w.dev)
        with torch.cuda.stream(stream):
            w.ctx.use_torch_stream()
            w.step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                w.step()
                torch.cuda.synchronize()  # a caller with one frame per sequence needs the pose before the next frame
            dt = (time.perf_counter() - t0) / steps
            ok, _ = w.check(w.snapshot())
        curve.append({"frames_per_step": b, "ms_per_step": dt * 1e3, "frames_per_s": b / dt, "aligner_success_fraction": ok})
        w.close()
        del w
        torch.cuda.empty_cache()
    return {"curve": curve,
            "note": "device-resident batch API, one synchronisation per step; up to two frames per CU the Gauss-Newton rounds run the instantiation "
                    "that owns the whole register file (align.hip, `lone`); the serial chain of a frame (100 damped iterations) sets ms_per_step "
                    "until the batch fills the chip"}


# ---------------------------------------------------------------------------------------------------------------------
# host-fed sequences: keypoints + descriptors of every frame arrive from pinned host memory ("host memory bandwidth /
# PCIe for input upload" is what the ranks of a node share)
# ---------------------------------------------------------------------------------------------------------------------
def streamed_leg(cfg, frames, keypoints, moving, max_fixed, batch, device_index, steps=6):
    """B frames per step whose stereo inputs (2 x 8 B keypoints + 2 x 32 B descriptor rows per keypoint = 160 kB per frame at 2000
    keypoints) are copied from pinned host memory on a copy stream into one of two device slots while the previous step computes
    on the other; the local map stays resident (it is the tracker's own output).  -> streamed rate, the resident rate of the same
    batch, the copy-only rate and how much of the shorter of the two the overlap hides"""
    import torch
    w = FrameWorkload(cfg, device_index, batch, keypoints, moving, max_fixed, len(frames), 0, frames=frames)
    sf = w.sframes
    names = ("left_kp", "right_kp", "left_desc", "right_desc")
    slots = [[getattr(sf, n) for n in names], [torch.empty_like(getattr(sf, n)) for n in names]]
    host = [getattr(sf, n).cpu().pin_memory() for n in names]  # ONE pinned image of the batch's inputs (both slots are fed from it)
    bytes_step = float(sum(t.numel() * t.element_size() for t in host))
    compute = torch.cuda.Stream(device=w.dev)
    copier = torch.cuda.Stream(device=w.dev)
    copied = [torch.cuda.Event() for _ in range(2)]
    consumed = [torch.cuda.Event() for _ in range(2)]

    def enqueue_copy(k):
        with torch.cuda.stream(copier):
            if k >= 2:
                copier.wait_event(consumed[k % 2])  # the step that read this slot has finished with it
            for dst, src in zip(slots[k % 2], host):
                dst.copy_(src, non_blocking=True)
            copied[k % 2].record(copier)

    def use_slot(k):
        for n, t in zip(names, slots[k % 2]):
            setattr(sf, n, t)

    with torch.cuda.stream(compute):
        w.ctx.use_torch_stream()
        # resident: the same batch with its inputs already in HBM
        w.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            w.step()
        torch.cuda.synchronize()
        t_resident = (time.perf_counter() - t0) / steps
        # copy only
        enqueue_copy(0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            enqueue_copy(k % 2)
        torch.cuda.synchronize()
        t_copy = (time.perf_counter() - t0) / steps
        # streamed: the copy of step k + 1 runs under the compute of step k
        copied = [torch.cuda.Event() for _ in range(2)]
        consumed = [torch.cuda.Event() for _ in range(2)]
        enqueue_copy(0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            enqueue_copy(k + 1)
            compute.wait_event(copied[k % 2])
            use_slot(k)
            w.step()
            consumed[k % 2].record(compute)
        torch.cuda.synchronize()
        t_streamed = (time.perf_counter() - t0) / steps
        snap = w.snapshot()
        ok, _ = w.check(snap)
    hidden = (t_copy + t_resident - t_streamed) / min(t_copy, t_resident)
    out = {"workload": "the headline frames, %d per step, stereo inputs (%.0f kB per frame) streamed from pinned host memory into two device slots; "
                       "local map resident" % (batch, bytes_step / batch / 1e3),
           "value": batch / t_streamed, "unit": "frames/s", "ms_per_step": t_streamed * 1e3, "frames_per_step": batch, "steps": steps,
           "resident_frames_per_s": batch / t_resident, "ms_per_step_resident": t_resident * 1e3,
           "h2d_gbps_streamed": bytes_step / t_streamed / 1e9, "h2d_gbps_copy_only": bytes_step / t_copy / 1e9, "ms_per_step_copy_only": t_copy * 1e3,
           "overlap_fraction": max(0.0, min(1.0, hidden)), "bound": "h2d" if t_copy > t_resident else "compute",
           "aligner_success_fraction": ok,
           "note": "PCIe-inclusive rate of a host-fed set of sequences on ONE GPU; at N > 1 the ranks share the host's memory and PCIe root "
                   "complexes (SURVEY 8e): compare value x N with the N-GPU line.  `value` of the headline keeps inputs resident in HBM."}
    w.close()
    del w
    torch.cuda.empty_cache()
    return out


# ---------------------------------------------------------------------------------------------------------------------
# real descriptors: the reference's own KITTI stereo pairs (tests/golden/ref_kitti.npz), extracted on the device
# ---------------------------------------------------------------------------------------------------------------------
def kitti_real_frames(device_index, cfg, target):
    """the 14 KITTI images the reference ships (city 00-04, highway 274 / 275) through the device extractor (FAST 15, 3 x 3 bins,
    ORB-256, `target` keypoints per image, libstdc++ selection order) and the host-pointer matcher / triangulator -> five tracked frames
    (01 <- 00, 02 <- 01, 03 <- 02, 04 <- 03, 275 <- 274): fixed = the frame's stereo measurements, moving = the previous frame's
    triangulated points in ITS camera frame, guess = identity like the reference's aligner tests (tests/test_aligners.cpp:1239)"""
    from srrg2_proslam_amd import ops
    z = np.load(os.path.join(ROOT, "tests", "golden", "ref_kitti.npz"))
    ctx = ops.Context(device_index)
    ep = ops.extractor_params(threshold=15, target=target, vertical=3, horizontal=3, selection_order=ops.SELECT_LIBSTDCXX)
    sp, tp = ops.stereo_params(cfg["stereo_matcher"], cfg["camera"]["rows"]), ops.triangulator_params(cfg)
    per_pair = []
    for key, n in (("city", 5), ("highway", 2)):
        for i in range(n):
            uvl, _, dl = ops.extract_features(ctx, ep, z[key + "_left"][i], capacity=4096)
            uvr, _, dr = ops.extract_features(ctx, ep, z[key + "_right"][i], capacity=4096)
            corr, _ = ops.stereo_match(ctx, sp, uvl, dl, uvr, dr)
            L, R = uvl[corr["fixed_idx"]], uvr[corr["moving_idx"]]
            keep = (L[:, 0] - R[:, 0] >= 0) & (L[:, 1] - R[:, 1] >= 0)  # the stereo adaptor's test (raw_data_preprocessor_stereo_projective.cpp:120-128)
            fixed = np.concatenate([L[keep], R[keep]], axis=1).astype(np.float32)
            xyz, valid = ops.triangulate(ctx, tp, fixed)
            valid = np.asarray(valid).astype(bool)
            per_pair.append(dict(uvl=uvl, dl=dl, uvr=uvr, dr=dr, xyz=np.asarray(xyz)[valid], desc=dl[corr["fixed_idx"]][keep][valid], key=key))
    ctx.close()
    frames = []
    for a, b in ((0, 1), (1, 2), (2, 3), (3, 4), (5, 6)):
        prev, cur = per_pair[a], per_pair[b]
        key, idx = ("city", b) if b < 5 else ("highway", b - 5)
        frames.append({"fr": {"uv_left": cur["uvl"], "desc_left": cur["dl"], "uv_right": cur["uvr"], "desc_right": cur["dr"]},
                       "mp": {"xyz": prev["xyz"].astype(np.float32), "desc": prev["desc"], "n_opt": np.zeros(len(prev["xyz"]), np.uint32)},
                       "X0": np.eye(4, dtype=np.float32), "T": np.eye(4, dtype=np.float32),
                       "images": (z[key + "_left"][idx], z[key + "_right"][idx])})
    return frames


def from_images_leg(device_index, cfg, frames, poses, batch, target, steps=3):
    """pixels -> poses on the device: FAST + binned selection + ORB of BOTH images of every stereo pair, epipolar matcher + triangulator,
    projective finder + GN aligner, everything on device-resident buffers (the extractor writes the matcher's input arrays in place).
    `frames` / `poses`: the five real KITTI frames of kitti_real_frames and what the CPU checker makes of their host-extracted features."""
    import torch
    from srrg2_proslam_amd import ops
    stride = 2048
    w = FrameWorkload(cfg, device_index, batch, stride, stride, 1024, len(frames), 0, frames=frames)
    dev = w.dev
    idx = torch.arange(batch, device=dev) % len(frames)
    img_l = torch.from_numpy(np.stack([f["images"][0] for f in frames])).to(dev)[idx].contiguous()
    img_r = torch.from_numpy(np.stack([f["images"][1] for f in frames])).to(dev)[idx].contiguous()
    ep = ops.extractor_params(threshold=15, target=target, vertical=3, horizontal=3, selection_order=ops.SELECT_LIBSTDCXX)
    st_l = torch.zeros((batch,), dtype=torch.int32, device=dev)
    st_r = torch.zeros((batch,), dtype=torch.int32, device=dev)
    sf = w.sframes
    want_nl, want_nr = sf.n_left.clone(), sf.n_right.clone()  # counts of the host-extracted features (uploaded by the constructor)

    def step(ev=None):
        if ev:
            ev[3].record()
        ops.extract_features_batch(w.ctx, ep, img_l, sf.left_kp, sf.left_desc, sf.n_left, st_l)
        ops.extract_features_batch(w.ctx, ep, img_r, sf.right_kp, sf.right_desc, sf.n_right, st_r)
        w.step(ev)

    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        w.ctx.use_torch_stream()
        step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        step(ev)
        torch.cuda.synchronize()
        snap = w.snapshot()
        ok, _ = w.check(snap)
    if int(st_l.min().item()) < 0 or int(st_r.min().item()) < 0:
        raise SystemExit("extractor reported error %d" % min(int(st_l.min().item()), int(st_r.min().item())))
    same_counts = bool(torch.equal(sf.n_left, want_nl)) and bool(torch.equal(sf.n_right, want_nr))
    out = {"workload": "pixels to poses: %d stereo pairs (the reference's KITTI images, tiled) per step: FAST 15 + 3 x 3 binned selection (libstdc++ order) + ORB-256 of "
                       "both images (target %d keypoints), epipolar matcher + triangulator, projective circle finder against the previous frame's points, "
                       "stereo GN aligner; all on device buffers" % (batch, target),
           "value": batch * steps / dt, "unit": "stereo frames/s", "images_per_second": 2 * batch * steps / dt, "ms_per_step": dt / steps * 1e3,
           "frames_per_step": batch,
           "ms_per_stage": {"extract (2 images per frame)": ev[3].elapsed_time(ev[0]), "stereo_match": ev[0].elapsed_time(ev[1]), "align": ev[1].elapsed_time(ev[2])},
           "aligner_success_fraction": ok,
           "parity": dict(w.parity(snap, poses), feature_counts_equal_host_extraction=same_counts)}
    w.close()
    del w, img_l, img_r
    torch.cuda.empty_cache()
    return out


def kitti_real_leg(device_index, cfg, batch, target=2000, steps=3):
    """matcher + projective search + GN on REAL ORB descriptors (correlated rows, real keypoint distribution), tiled to `batch`; the CPU
    checker on the same five frames beside it"""
    import torch
    frames = kitti_real_frames(device_index, cfg, target)
    stride = 64 * ((max(max(len(f["fr"]["uv_left"]), len(f["fr"]["uv_right"]), len(f["mp"]["xyz"])) for f in frames) + 63) // 64)
    w = FrameWorkload(cfg, device_index, batch, stride, stride, 1024, len(frames), 0, frames=frames)
    stream = torch.cuda.Stream(device=w.dev)
    with torch.cuda.stream(stream):
        w.ctx.use_torch_stream()
        w.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            w.step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        snap = w.snapshot()
        ok, it_exec = w.check(snap)
        kt = w.kernel_times(2)
    _, cpu_s, poses = cpu_baseline(cfg, frames, len(frames))
    out = {"workload": "the reference's own KITTI stereo pairs (tests/golden/ref_kitti.npz: city 00-04, highway 274 / 275): device extractor (FAST 15, 3 x 3 bins, "
                       "%d keypoints per image, ORB-256) -> epipolar matcher + triangulator -> projective circle finder against the previous frame's "
                       "triangulated points + stereo GN aligner from an identity guess, kitti.conf parameters; 5 tracked frames tiled to %d" % (target, batch),
           "keypoints_per_image": [int(len(f["fr"]["uv_left"])) for f in frames], "local_map_points": [int(len(f["mp"]["xyz"])) for f in frames],
           "value": batch * steps / dt, "unit": "frames/s", "ms_per_step": dt / steps * 1e3, "frames_per_step": batch,
           "ms_per_kernel": {"stereo_match5_kernel": kt["matcher_ms"], "align_kernel (search)": kt["search_ms"], "gn_kernel": kt["gn_ms"]},
           "search_ms_by_round": kt["search_ms_by_round"], "gn_ms_by_round": kt["gn_ms_by_round"],
           "stereo_matches_per_frame": snap.get("n_match"), "fixed_points_per_frame": snap["n_fixed"], "aligner_correspondences_per_frame": snap["n_corr"],
           "aligner_success_fraction": ok, "gn_iterations_executed_mean": it_exec, "parity": w.parity(snap, poses),
           "cpu_checker_fps": len(frames) / cpu_s}
    w.close()
    del w
    torch.cuda.empty_cache()
    try:
        out["from_images"] = from_images_leg(device_index, cfg, frames, poses, min(batch, 2048), target)
    except SystemExit as exc:
        out["from_images"] = {"error": str(exc)}
    return out


# ---------------------------------------------------------------------------------------------------------------------
# latency of the drop-in path: one sequence, one frame at a time, host pointers (tools/latency_b1.cpp)
# ---------------------------------------------------------------------------------------------------------------------
def write_latency_frames(path, par, seq):
    """the frames file tools/latency_b1.cpp reads: header, parameter block, then per frame the flat arrays"""
    import struct
    N, NM = seq[0]["fr"]["uv_left"].shape[0], seq[0]["mp"]["xyz"].shape[0]
    with open(path, "wb") as fh:
        fh.write(struct.pack("<4i", len(seq), N, NM, len(par)))
        fh.write(np.asarray(par, np.float32).tobytes())
        for d in seq:
            fr, mp = d["fr"], d["mp"]
            for a, dt in ((fr["uv_left"], np.float32), (fr["desc_left"], np.uint8), (fr["uv_right"], np.float32), (fr["desc_right"], np.uint8),
                          (mp["xyz"], np.float32), (mp["desc"], np.uint8), (mp["n_opt"], np.uint32), (d["X0"], np.float32)):
                fh.write(np.ascontiguousarray(a, dt).tobytes())


def latency_params(cfg):
    cam, m, tri, f, al = cfg["camera"], cfg["stereo_matcher"], cfg["triangulator"], cfg["projective_finder"], cfg["aligner"]
    return [cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["cols"], cam["rows"], cam["baseline_m"], cfg["projector"]["range_min"], cfg["projector"]["range_max"],
            m["maximum_descriptor_distance"], m["maximum_distance_ratio_to_second_best"], m["minimum_matching_ratio"], m["maximum_disparity_pixels"],
            m["epipolar_line_thickness_pixels"], tri["minimum_disparity_pixels"], tri["infinity_depth_meters"],
            f["maximum_descriptor_distance"], f["maximum_distance_ratio_to_second_best"], f["minimum_matching_ratio"], f["minimum_descriptor_distance"],
            f["descriptor_distance_step_size_pixels"], f["maximum_search_radius_pixels"], f["minimum_search_radius_pixels"], f["search_radius_step_size_pixels"],
            f["minimum_number_of_iterations"], f["maximum_estimate_change_norm_for_convergence"], f["number_of_solver_iterations_per_projection"],
            al["diagonal_info"][0], al["diagonal_info"][1], al["diagonal_info"][2], al["chi_threshold"], al["enable_inverse_depth_weighting"], al["damping"],
            al["max_iterations"], al["min_num_inliers"], al["min_num_correspondences"], -cam["fx"] * cam["baseline_m"]]


def latency_b1(cfg, frames, n_frames=64):
    """one frame at a time through the C++ adapters of plugin/ (AoS clouds, gather included) and through the bare C-ABI (flat arrays),
    PCIe and launch latency included; the CPU checker on the same frames beside it, poses compared"""
    import struct
    import subprocess
    import tempfile
    from oracle import binding as ob
    exe = os.path.join(ROOT, "tools", "bin", "latency_b1")
    if not os.path.exists(exe):
        return {"error": "tools/bin/latency_b1 is not built (__graft_entry__.build())"}
    par = latency_params(cfg)
    seq = [frames[k % len(frames)] for k in range(n_frames)]
    warm = 4
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "frames.bin")
        write_latency_frames(path, par, seq)
        run = subprocess.run([exe, path, str(warm)], capture_output=True, text=True, timeout=600)
        if run.returncode != 0:
            return {"error": "latency_b1 exit %d: %s" % (run.returncode, run.stderr[-400:])}
        out = json.loads([l for l in run.stdout.splitlines() if l.startswith("{")][-1])
        poses = np.fromfile(path + ".poses", np.float32).reshape(2, n_frames, 4, 4)
        # the same frames with the opt-in early exit (prs_aligner_params.step_norm_exit = 1e-5): LESS work than the reference, reported beside
        run_x = subprocess.run([exe, path, str(warm), "1e-5"], capture_output=True, text=True, timeout=600)
        out_x, poses_x = None, None
        if run_x.returncode == 0:
            out_x = json.loads([l for l in run_x.stdout.splitlines() if l.startswith("{")][-1])
            poses_x = np.fromfile(path + ".poses", np.float32).reshape(2, n_frames, 4, 4)
    # the CPU checker, same frames, same order, ONE finder object across them (incl. the warm-up frames: its state carries over)
    sp, tp, pp, ap = oracle_params(cfg)
    finder = ob.ProjectiveFinder(pp)
    worst, worst_x, times = [0.0, 0.0], [0.0, 0.0], []
    for k in range(-warm, n_frames):
        d = seq[(k + warm) % n_frames] if k < 0 else seq[k]
        fr, mp = d["fr"], d["mp"]
        t0 = time.perf_counter()
        corr, _ = ob.stereo_match(fr["uv_left"], fr["desc_left"], fr["uv_right"], fr["desc_right"], sp)
        fixed, src = ob.stereo_assemble(fr["uv_left"], fr["uv_right"], corr)
        ob.triangulate(fixed, tp)
        ap.mean_disparity = ob.mean_disparity(fixed)
        finder.set_fixed(fixed, fr["desc_left"][src])
        finder.set_moving(mp["xyz"], mp["desc"])
        res, _ = ob.align_frame(finder, ap, fixed, mp["xyz"], ob.info_scale_from_nopt(mp["n_opt"]), d["X0"])
        dt = time.perf_counter() - t0
        if k >= 0:
            times.append(dt * 1e3)
            X = np.array(res.X, np.float32).reshape(4, 4)
            for flavour in range(2):
                worst[flavour] = max(worst[flavour], float(np.linalg.norm(poses[flavour, k] - X) / np.linalg.norm(X)))
                if poses_x is not None:
                    worst_x[flavour] = max(worst_x[flavour], float(np.linalg.norm(poses_x[flavour, k] - X) / np.linalg.norm(X)))
    finder.close()
    out["cpu_checker"] = {"ms_per_frame_mean": float(np.mean(times)), "fps": 1e3 / float(np.mean(times)), "cores": 1, "kind": "port"}
    out["pose_rel_frobenius_max_vs_cpu_checker"] = {"adapters": worst[0], "c_abi": worst[1]}
    if out_x is not None:
        out["with_step_norm_exit"] = {"does_less_work_than_the_reference": True, "step_norm_exit": 1e-5,
                                      "adapters_ms_per_frame_median": out_x["adapters"]["ms_per_frame_median"], "c_abi_ms_per_frame_median": out_x["c_abi"]["ms_per_frame_median"],
                                      "pose_rel_frobenius_max_vs_100_iteration_checker": {"adapters": worst_x[0], "c_abi": worst_x[1]},
                                      "note": "opt-in early exit of the Gauss-Newton loop (off by default): what one frame at a time costs when the frozen tail is cut"}
    out["note"] = ("one sequence, one frame at a time, host pointers, PCIe + launch + synchronisation inside every call; finder object carried across "
                   "frames; `adapters` = plugin/proslam_hip_plugin.hpp on array-of-structs clouds (gather / scatter timed), `c_abi` = flat arrays; the "
                   "reference times frames the same way (apps/app_benchmark.cpp:345-353)")
    return out


# ---------------------------------------------------------------------------------------------------------------------
# closed loop (tools/bench_tracking.py); distributed = BASELINE.json config 5
# ---------------------------------------------------------------------------------------------------------------------
def closed_loop(args):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_tracking
    from srrg2_proslam_amd import sharding
    from srrg2_proslam_amd.sharding import KITTI_SEQUENCE_FRAMES
    rank, world, local_rank = sharding.rank_world()
    extra = {"steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
             "dtype": "u64 popcount (Hamming) + f32 (projection, Jacobians, 6x6 normal equations) + f64 (landmark filters)", "data": "synthetic"}
    if world <= 1:
        out = bench_tracking.run(batch=min(args.batch, 4096), frames=args.frames, keypoints=args.keypoints, check=1 if not args.no_cpu_baseline else 0)
        out.update(extra)
        print(json.dumps(out))
        return
    # config 5: KITTI sequences 00-07, longest first to the least loaded rank (a sequence is a serial chain: it stays on one GPU);
    # every sequence runs as `batch` independent replicas (own landmarks and noise) for frames-scale of its length
    import torch
    shared = os.environ.get("PRS_BENCH_SHARE_GPU", "0") == "1"
    dev = 0 if shared else local_rank
    torch.cuda.set_device(dev)
    sharding.init_distributed("gloo" if shared else "nccl", dev)
    red_dev = "cpu" if shared else torch.device("cuda", dev)
    batch = min(args.batch, 1024)
    parity = []

    def run_sequence(s, n_frames):
        # the first sequence of every rank is replayed on the CPU checker (pose of every frame bit for bit), the others run unchecked
        check = 1 if not parity and not args.no_cpu_baseline else 0
        o = bench_tracking.run(batch=batch, frames=n_frames, keypoints=args.keypoints, check=check, device=dev, seed_offset=1000 * s)
        parity.append({"sequence": s, "checked": bool(check), "parity_vs_oracle_chain": o.get("parity_vs_oracle_chain") if check else "not checked"})
        n = o["config"]["tracked_frames_timed"] * batch
        return n, n / o["value"]

    torch.cuda.synchronize()
    fps, slowest, parts = sharding.run_sequences_over_ranks(KITTI_SEQUENCE_FRAMES, args.frames_scale, run_sequence, red_dev)
    mine = [p[0] for p in parts]
    if rank == 0:
        out = {"metric": "tracked frames/sec on KITTI-00 stereo (1241x376, ~2k kp); SE(3) vs ref", "value": fps, "unit": "frames/s", "n_gpus": world,
               "ms_per_step": None,
               "config": {"workload": "closed loop (BASELINE.json config 5), %d keypoints per image, KITTI sequences 00-07 (%s frames, the first %.1f %% of each: "
                                      "%s frames) sharded over %d ranks longest-first, %d replicas per sequence; trajectories follow KITTI 00 (the only "
                                      "ground truth shipped besides 01)" % (args.keypoints, list(KITTI_SEQUENCE_FRAMES), 100.0 * args.frames_scale,
                                                                           [max(int(round(n * args.frames_scale)), 4) for n in KITTI_SEQUENCE_FRAMES], world, batch),
                          "keypoints_per_image": args.keypoints, "frames_scale": args.frames_scale, "replicas_per_sequence": batch,
                          "sequences_of_rank_0": mine, "rank_0_parts": parts, "slowest_rank_seconds": slowest, "parity_of_rank_0": parity},
               "ranks": {"backend": "gloo (PRS_BENCH_SHARE_GPU=1: all ranks on device 0)" if shared else "nccl (RCCL)",
                         "backend_world_size": sharding.backend_world_size()}}
        out.update(extra)
        print(json.dumps(out))
    sharding.shutdown()


def visible_gpu_count():
    """GPUs this process would see, WITHOUT a HIP call (the parent of the ranks must not initialise the runtime): the KFD topology
    (one node per agent, `simd_count` > 0 = a GPU), narrowed by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES"""
    import glob
    n = 0
    for props in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            fields = dict(line.split()[:2] for line in open(props) if len(line.split()) >= 2)
            n += int(fields.get("simd_count", "0")) > 0
        except OSError:
            continue
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            listed = len([x for x in v.split(",") if x.strip() != ""])
            n = min(n, listed) if n else listed
    if n == 0:  # no readable topology (some containers): ask a CHILD process, so that still no GPU call happens in this one
        import subprocess
        try:
            n = int(subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True,
                                   timeout=300).stdout.strip().splitlines()[-1])
        except (ValueError, IndexError, OSError, subprocess.SubprocessError):
            n = 0
    return n


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: this process (which makes NO GPU call: the devices are counted from the KFD
    topology in sysfs, visible_gpu_count) starts N fresh children of itself, one per device, with the environment
    torch.distributed.run would give them, waits, and exits with the worst child's code.  Rank 0 prints the line.  Fewer visible
    devices than N is an error, never a 1-GPU run."""
    import socket
    import subprocess
    n = args.gpus
    shared = os.environ.get("PRS_BENCH_SHARE_GPU", "0") == "1"
    if not (shared or args.dry_run):
        have = visible_gpu_count()
        if have < n:
            raise SystemExit("bench.py --gpus %d: only %d device(s) visible (PRS_BENCH_SHARE_GPU=1 puts all ranks on device 0)" % (n, have))
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n), "MASTER_ADDR": "127.0.0.1",
                    "MASTER_PORT": str(port), "PRS_BENCH_SPAWNED": "1"})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    codes = []
    for pr in procs:
        try:
            codes.append(pr.wait(timeout=3600))
        except subprocess.TimeoutExpired:
            pr.kill()
            codes.append(124)
    if any(codes):
        raise SystemExit("bench.py --gpus %d: rank exit codes %s" % (n, codes))


def dry_run(args):
    """the N-rank control flow of main() (rendezvous, barriers, SUM / MAX over ranks, rank-0 line) with a sleep as the step"""
    from srrg2_proslam_amd import sharding
    rank, world, _ = sharding.rank_world()
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))
    sharding.init_distributed("gloo")
    sharding.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.01 * (1 + rank))
    sharding.barrier()
    local = time.perf_counter() - t0
    fps, elapsed = sharding.aggregate_throughput(args.batch * args.steps, local)
    per_rank = sharding.gather_over_ranks(args.batch * args.steps / local)
    if rank == 0:
        print(json.dumps({"metric": "tracked frames/sec on KITTI-00 stereo (1241x376, ~2k kp); SE(3) vs ref", "value": 0.0, "unit": "frames/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": "none", "data": "dry-run (no device work: control flow of the N-rank path only)",
                          "config": {"workload": "dry run", "frames_per_step_per_gpu": args.batch},
                          "ranks": {"backend": "gloo", "backend_world_size": sharding.backend_world_size(), "fps_per_rank": per_rank,
                                    "stand_in_fps": fps}}))
    sharding.shutdown()


def main():
    args = parse()
    if args.cpu_worker > 0:
        cpu_worker(args)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)
        return
    if args.dry_run:
        dry_run(args)
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

    from srrg2_proslam_amd import configs, synthetic as syn

    cfg = configs.get("kitti")
    B, N, NM = args.batch, args.keypoints, args.moving
    # synthetic KITTI-00-shaped inputs: distinct seeds per rank (independent sequences)
    w = FrameWorkload(cfg, local_rank, B, N, NM, args.max_fixed, args.unique, syn.seed_for(1, 0) + 100000 * rank, args.all_iterations)
    stream = torch.cuda.Stream(device=w.dev)

    def barrier():
        torch.cuda.synchronize()
        sharding.barrier()

    with torch.cuda.stream(stream):
        w.ctx.use_torch_stream()
        for _ in range(args.warmup):
            w.step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            w.step()
        barrier()
        elapsed_local = time.perf_counter() - t0
        fps, elapsed = sharding.aggregate_throughput(B * args.steps, elapsed_local, red_dev)
        fps_per_rank = sharding.gather_over_ranks(B * args.steps / elapsed_local, red_dev)
        snap = w.snapshot()  # results of the timed configuration
        status_ok, it_exec = w.check(snap)
        kt = w.kernel_times(max(args.timing_steps, 1))  # per-kernel times: a separate pass, outside the timed region
        # ---- steady state: the same frames again with the finders carried over (radius / threshold schedule adapted) ----
        steady = None
        if rank == 0 and world == 1 and not args.no_steady_state:
            for _ in range(6):  # radius 50 -> 10 px in steps of 10, threshold 25 -> 50
                w.step(fresh_finders=False)
            torch.cuda.synchronize()
            n_st = max(args.steps // 2, 1)
            t1 = time.perf_counter()
            for _ in range(n_st):
                w.step(fresh_finders=False)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
            st = w.aframes.state_of(0)
            skt = w.kernel_times(2, fresh_finders=False)
            steady = {"value": B * n_st / dt, "unit": "frames/s", "ms_per_step": dt / n_st * 1e3,
                      "search_radius_pixels": int(st.search_radius_pixels), "descriptor_distance": float(st.descriptor_distance),
                      "aligner_correspondences_per_frame": w.aframes.n_corr.float().mean().item(),
                      "ms_per_kernel": {"stereo_match5_kernel": skt["matcher_ms"], "align_kernel (search)": skt["search_ms"], "gn_kernel": skt["gn_ms"]},
                      "note": "finder objects live across steps like the reference's (correspondence_finder_projective_base_impl.cpp:277-287): "
                              "after six tracked frames the search radius has shrunk to its minimum and the descriptor threshold has grown"}

    # ---- algorithmic bytes / flops per kernel -----------------------------------------------------------------
    n_match, n_fixed, n_corr = snap["n_match"], snap["n_fixed"], snap["n_corr"]
    ms_match, ms_search, ms_gn = kt["matcher_ms"], kt["search_ms"], kt["gn_ms"]
    # matcher 40 (N_L + N_R) + 12 M, triangulator 16 M + 13 M: the launch runs both (fused epilogue)
    bytes_match = 40.0 * (2 * N) + 12.0 * n_match + 29.0 * n_fixed
    bytes_search = 44.0 * NM + 40.0 * n_fixed + 64 + 12.0 * n_corr  # projective finder, per recompute
    # searches a frame goes through per step, counted by the finder objects themselves (state.num_recomputes of the fresh finders):
    # a launch in which a handful of frames search is priced by those frames, not as a whole launch
    searches = snap["searches_per_frame"]
    gbps_match = B * bytes_match / (ms_match * 1e-3) / 1e9
    gbps_search = B * bytes_search * searches / (ms_search * 1e-3) / 1e9 if ms_search > 0 else 0.0
    flops_gn = FLOP_PER_CORRESPONDENCE * n_corr * it_exec * B  # per step
    tflops_gn = flops_gn / (ms_gn * 1e-3) / 1e12 if ms_gn > 0 else 0.0
    evidence = profile_evidence(B, N)
    roof_search = {
        "kernel": "align_kernel<512, split, circle> (projective search: projection, cell-grid circle search, Hamming, candidate filter; all its launches of one step)",
        "bound": "hbm", "achieved": gbps_search, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": gbps_search / HBM_PEAK_GBPS,
        "traffic": evidence["search"] if evidence else None, "traffic_unit": evidence["unit"] if evidence else None,
        "traffic_source": evidence["source"] if evidence else None,
        "ms_per_step": ms_search, "ms_by_round": kt["search_ms_by_round"], "searches_per_frame_and_step": searches,
        "algorithmic_bytes_per_frame_and_launch": bytes_search, "frames_per_launch": B,
        "traffic_over_algorithmic": (evidence["search"] / (B * bytes_search * searches)) if evidence and searches > 0 else None,
        "counter_gbps": (evidence["search"] / (ms_search * 1e-3) / 1e9) if evidence and ms_search > 0 else None,
        "note": "not bandwidth-bound: the kernel retires 0.84 vector instructions per cycle and CU (34 k per frame-search, profiles/r04/align_pmc.txt) "
                "and 77 % of a launch is the candidate scan, so only the instruction count of a scan trip counts.  Round 6 (17.1 -> 14.0 ms same-box, "
                "profiles/r06/search_scan_r06.txt): fixed descriptor rows in LDS in lattice order, a trip compares two 96-bit half rows with the irrelevance "
                "bound (one LDS round trip), entry + pattern test + full distance for the survivors only, which wait in four LDS slots per thread; no operand "
                "rows written (the Gauss-Newton kernel gathers through the correspondence vector).  frac prices the algorithmic bytes, counter_gbps the "
                "bytes the PMC passes saw",
    }
    roof_gn = {
        "kernel": "gn_kernel<SLOTS, stereo> (reprojection-error Gauss-Newton rounds: factor linearisation, fixed-shape H / b reduction, "
                  "6x6 solve; all its launches of one step)",
        "bound": "valu", "achieved": tflops_gn, "peak": FP32_VECTOR_TFLOPS, "unit": "TFLOP/s", "frac": tflops_gn / FP32_VECTOR_TFLOPS,
        "traffic": evidence["gn"] if evidence else None, "traffic_unit": evidence["unit"] if evidence else None,
        "traffic_source": evidence["source"] if evidence else None,
        "ms_per_step": ms_gn, "ms_by_round": kt["gn_ms_by_round"], "algorithmic_flop_per_step": flops_gn,
        "flop_per_correspondence_iteration": FLOP_PER_CORRESPONDENCE,
        "counter_gbps": (evidence["gn"] / (ms_gn * 1e-3) / 1e9) if evidence and ms_gn > 0 else None,
        "note": "fp32 vector arithmetic, no MFMA: 6x6 normal equations are not a dense contraction; the kernel is bound by vector issue "
                "(SQ_INSTS_VALU per launch / (launch cycles x CUs) = 1.07 vector instructions per cycle and CU, profiles/r05/rocprof_summary.json "
                "pmc_sq2; round 5 took the IEEE reciprocals of the iteration to v_rcp_f32 + one Newton step where all 2^32 operands show it equal, "
                "tests/test_reciprocal_gpu.py; round 6: operand rows gathered through the correspondence vector, no spills in the headline "
                "instantiation), frac prices only the algorithmic flops",
    }
    roof_match = {
        "kernel": "stereo_match5_kernel<%d> (the kernel BASELINE.json north_star prices; matcher + fused adaptor / triangulator epilogue)" % (1 if N <= 1024 else 2),
        "bound": "hbm", "achieved": gbps_match, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": gbps_match / HBM_PEAK_GBPS,
        "traffic": evidence["matcher"] if evidence else None, "traffic_unit": evidence["unit"] if evidence else None,
        "traffic_source": evidence["source"] if evidence else None,
        "ms_per_launch": ms_match, "algorithmic_bytes_per_frame": bytes_match, "frames_per_launch": B,
        "traffic_over_algorithmic": (evidence["matcher"] / (B * bytes_match)) if evidence else None,
        "counter_gbps": (evidence["matcher"] / (ms_match * 1e-3) / 1e9) if evidence and ms_match > 0 else None,
    }
    # the matcher against its ISSUE roofline (review r04, item 3): wave-instructions per frame from the PMC passes x the issue cost of
    # their class on gfx950 (profiles/r04/valu_issue_rates.txt: a SIMD retires a full-rate vector instruction every 2.4 cycles and one
    # of the half-rate class -- popcount, funnel shifts, min / max, packed and three-operand integer: what this kernel is made of --
    # every 4.3, with >= 2 waves per SIMD), four SIMDs per CU, one frame per CU at a time
    if evidence and evidence.get("insts") and evidence["insts"].get("matcher") and ms_match > 0:
        ins = evidence["insts"]["matcher"]
        valu_frame = ins.get("valu", 0.0) / B
        cyc_frame = ms_match * 1e-3 * GPU_CLOCK_HZ * N_CU / B
        lo, hi = valu_frame / 4 * 2.4, valu_frame / 4 * 4.3
        roof_match["issue_roofline"] = {
            "vector_instructions_per_frame": valu_frame, "scalar_instructions_per_frame": ins.get("salu", 0.0) / B, "lds_instructions_per_frame": ins.get("lds", 0.0) / B,
            "issue_bound_cycles_per_frame": [lo, hi], "measured_cycles_per_frame": cyc_frame, "achieved_over_issue_bound": [lo / cyc_frame, hi / cyc_frame],
            "assumed_clock_hz": GPU_CLOCK_HZ, "compute_units": N_CU, "source": evidence["source"],
            "note": "all-full-rate / all-half-rate bracket; the kernel keeps one 1024-thread workgroup per CU (151 KB LDS), so the rest of a frame's "
                    "cycles are its own barriers and single-wave scans (56 % of the wave cycles parked, profiles/r04/matcher_pmc.txt)"}
    dominant = max((ms_search, "search", roof_search), (ms_gn, "gn", roof_gn), (ms_match, "matcher", roof_match))
    total_k = ms_match + ms_search + ms_gn

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
        "roofline": dict(dominant[2], dominant_of="the three kernels of the step (%s)" % dominant[1]),
        "roofline_search": roof_search,
        "roofline_gn": roof_gn,
        "roofline_matcher": roof_match,
        # SURVEY 8(d) single-pass definition of a tracked frame (inputs read once, outputs written once, everything else on chip):
        # 40 (N_L + N_R) + 29 M + 48 N_m + 64 + 12 N_c + 64 bytes, against the whole step's time
        "roofline_whole_step": (lambda bytes_frame: {
            "bound": "hbm", "achieved": B * world * bytes_frame / (elapsed / args.steps) / 1e9, "peak": HBM_PEAK_GBPS * world, "unit": "GB/s",
            "frac": B * bytes_frame / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBPS, "algorithmic_bytes_per_frame": bytes_frame,
            "note": "the step is bound by vector issue (GN, search) and by the serial 100-iteration chain, not by HBM: this is the figure SURVEY 8(d) "
                    "prices the whole frame with"})(40.0 * (2 * N) + 29.0 * n_match + 48.0 * NM + 64 + 12.0 * n_corr + 64),
        "kernel_time_share": {"stereo_match5_kernel": ms_match / total_k, "align_kernel (search)": ms_search / total_k, "gn_kernel": ms_gn / total_k},
        "kernel_timing": "HIP events on the launch stream in a separate pass of %d steps after the timed loop (the timed loop carries no events)" % max(args.timing_steps, 1),
    }
    if world > 1:
        out["ranks"] = {"backend": "gloo (PRS_BENCH_SHARE_GPU=1: all ranks on device 0)" if shared else "nccl (RCCL)",
                        "backend_world_size": sharding.backend_world_size(), "fps_per_rank": fps_per_rank,
                        "launcher": "bench.py itself" if os.environ.get("PRS_BENCH_SPAWNED") == "1" else "external (torch.distributed.run)"}
    if steady:
        out["steady_state"] = steady

    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.cpu_frames > 0:
        cpu_fps, cpu_s, poses = cpu_baseline(cfg, w.uniq, args.cpu_frames)
        out["cpu_baseline"] = {
            "value": cpu_fps,
            "unit": "frames/s",
            "cores": 1,
            "kind": "port",
            "sample": "%d frames of the same synthetic workload (stereo match + assemble + triangulate + 100-iteration align), "
                      "single-threaded CPU restatement (-O2, no fast-math), %.1f s; host has %d cores" % (args.cpu_frames, cpu_s, os.cpu_count() or 0),
        }
        # the same run doubles as an end-to-end check of the device pipeline on the bench inputs
        out["parity_on_bench_inputs"] = w.parity(snap, poses)
        uniq_frames = w.uniq
        cores, quota = usable_cores()
        n_workers = args.cpu_all_cores if args.cpu_all_cores >= 0 else min(cores, 256)
        if n_workers > 0:
            total, ok, span, wall = cpu_all_cores(args, n_workers, 200)
            out["cpu_baseline_all_cores"] = {
                "value": total, "unit": "frames/s", "cores": ok, "kind": "port",
                "sample": "%d worker processes (one independent sequence each, 200 frames per worker) of the same CPU restatement: all frames / "
                          "wall clock from the first worker's start to the last worker's end (%.1f s; %.1f s incl. interpreter start-up); the host shows "
                          "%d cores, the cgroup CPU quota is %s: = %.1f x the one-core rate" % (ok, span, wall, os.cpu_count() or 0,
                                                                                              ("%.1f cores" % quota) if quota else "none", total / max(cpu_fps, 1e-9)),
            }
    else:
        uniq_frames = None
    w.close()
    del w
    torch.cuda.empty_cache()
    if uniq_frames is not None:
        out["latency_b1"] = latency_b1(cfg, uniq_frames)

    # ---- the other BASELINE configurations and the stateful loop, outside the headline timing (rank 0, single GPU) ----
    if rank == 0 and world == 1 and not args.no_other_configs:
        small = min(B, 6144)  # 24 frames per CU: whole resident sets of the GN kernel (8 frames per CU) and of the search kernel (3)
        others = {}
        for name, cname, kp, mv, mf, cidx in (("euroc", "euroc", 1000, 1000, 512, 2), ("tum", "tum", 1000, 1000, 1024, 3), ("kitti_n1000", "kitti", 1000, 1000, 512, 1)):
            try:
                others[name] = small_config_leg(name, configs.get(cname), kp, mv, mf, small, local_rank, syn.seed_for(cidx, 0) + 31)
            except SystemExit as exc:  # a loud per-frame error of a side leg must not take the headline line with it
                others[name] = {"error": str(exc)}
        out.update(f_rows_leg())
        if not args.no_cpu_baseline:
            try:
                out["a13_alt"] = a13_alt_leg(cfg, uniq_frames[:13] if uniq_frames is not None else make_unique_frames(cfg, 13, N, NM, syn.seed_for(1, 0)),
                                             N, NM, args.max_fixed, small, local_rank)
            except (SystemExit, RuntimeError) as exc:
                out["a13_alt"] = {"error": str(exc)}
        if not args.no_cpu_baseline:
            try:
                out["tolerance_exit"] = tolerance_exit_leg(cfg, uniq_frames[:13] if uniq_frames is not None else make_unique_frames(cfg, 13, N, NM, syn.seed_for(1, 0)),
                                                           N, NM, args.max_fixed, small, local_rank)
            except (SystemExit, RuntimeError) as exc:
                out["tolerance_exit"] = {"error": str(exc)}
        try:
            out["small_batch"] = small_batch_curve(cfg, uniq_frames if uniq_frames is not None else make_unique_frames(cfg, 13, N, NM, syn.seed_for(1, 0)),
                                                   N, NM, args.max_fixed, local_rank)
        except (SystemExit, RuntimeError) as exc:
            out["small_batch"] = {"error": str(exc)}
        try:
            out["streamed"] = streamed_leg(cfg, uniq_frames if uniq_frames is not None else make_unique_frames(cfg, 13, N, NM, syn.seed_for(1, 0)),
                                           N, NM, args.max_fixed, small, local_rank)
        except (SystemExit, RuntimeError) as exc:
            out["streamed"] = {"error": str(exc)}
        try:
            others["kitti_real"] = kitti_real_leg(local_rank, cfg, small)
            if "from_images" in others["kitti_real"]:
                out["from_images"] = others["kitti_real"].pop("from_images")
        except (SystemExit, OSError, KeyError) as exc:
            others["kitti_real"] = {"error": str(exc)}
        out["other_configs"] = others
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_tracking
        # 4096 sequences x 62 frames; all five distinct sequences replayed on the CPU checker (310 frames: parity of every pose + CPU rate)
        cl = bench_tracking.run(batch=4096, frames=62, keypoints=N, check=0 if args.no_cpu_baseline else 5, device=local_rank)
        out["closed_loop"] = {k: cl[k] for k in ("value", "unit", "ms_per_step", "config", "ms_per_stage", "map_points_mean", "aligner_correspondences_mean",
                                                 "finder_retries_per_frame", "track_losses_per_frame", "drift_percent_of_path", "parity_vs_oracle_chain",
                                                 "cpu_baseline") if k in cl}
    if rank == 0:
        print(json.dumps(out))
    sharding.shutdown()


if __name__ == "__main__":
    main()
