"""brute-force matcher throughput: B cloud pairs of N x N descriptors resident in HBM
usage: python tools/bench_bruteforce.py [B] [N] [max_dist]                       uniform random rows, one true partner per point
       python tools/bench_bruteforce.py real [B] [max_dist] [target] [capacity]  descriptors of real KITTI stereo pairs
the dense phase is the context's default (popcount kernels) unless PRS_BF_MFMA=1 / auto is set; run(..., dense=ops.BF_DENSE_*) picks one"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from srrg2_proslam_amd import ops  # noqa: E402


def run(B, N, max_dist=50.0, iters=5, quiet=False, dense=None):
    ctx = ops.Context(0)
    ctx.use_torch_stream()
    if dense is not None:
        ctx.set_bruteforce_dense_phase(dense)
    clouds = ops.BruteforceClouds(0, B, N, N)
    g = torch.Generator(device="cuda").manual_seed(1)
    fixed = torch.randint(0, 256, (B, N, 32), device="cuda", dtype=torch.uint8, generator=g)
    # moving = permuted fixed with ~4 % of the bits flipped: every point has one true partner at distance ~10
    perm = torch.argsort(torch.rand((B, N), device="cuda", generator=g), dim=1)
    moving = torch.gather(fixed, 1, perm[..., None].expand(-1, -1, 32))
    noise = torch.zeros_like(moving)
    for bit in range(8):
        noise |= ((torch.rand((B, N, 32), device="cuda", generator=g) < 0.04).to(torch.uint8) << bit)
    clouds.fixed_desc.copy_(fixed)
    clouds.moving_desc.copy_(moving ^ noise)
    clouds.n_fixed.fill_(N)
    clouds.n_moving.fill_(N)
    p = ops.bruteforce_params(max_dist, 0.9)
    ops.bruteforce_match_batch(ctx, p, clouds)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.bruteforce_match_batch(ctx, p, clouds)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    pairs = float(B) * N * N
    # 2 VALU per 32-bit word (v_xor + accumulating v_bcnt), 8 words, + 1 compare = 17 lane-ops per pair;
    # VALU peak: 256 CUs x 4 SIMD x 16 lanes x 2.4 GHz
    valu_peak = 256 * 4 * 16 * 2.4e9
    matches, ok = clouds.n_matches.float().mean().item(), int((clouds.status >= 0).all().item())
    if not quiet:
        print("B=%d N=%d max_dist=%.0f matches/pair=%.0f ok=%d: %.3f ms/launch, %.2f G pairs/s, %.1f us/pair-of-clouds, "
              "%.1f%% of the 17-op/pair VALU bound (popcount kernels), %.1f%% of the dense I8 MFMA rate at 512 op/pair" % (
                  B, N, max_dist, matches, ok, ms, pairs / ms / 1e6, ms * 1e3 / B, 100 * (pairs * 17 / (ms * 1e-3)) / valu_peak,
                  100 * pairs * 512 / (ms * 1e-3) / 5e15))
    ctx.close()
    del clouds
    torch.cuda.empty_cache()
    return {"cloud_pairs_per_launch": B, "points_per_cloud": N, "maximum_descriptor_distance": max_dist, "matches_per_cloud_pair": matches,
            "all_status_ok": bool(ok), "ms_per_launch": ms, "descriptor_pairs_per_launch": pairs, "pairs_per_s": pairs / (ms * 1e-3)}


def run_real(B, max_dist=50.0, iters=5, quiet=False, target=1000, capacity=0, dense=None):
    """B cloud pairs of REAL descriptors: the left / right keypoints our extractor finds in the seven stereo pairs of KITTI images the
    reference's tests hold (tests/golden/ref_kitti.npz; kitti.conf extractor settings, ~1000 points per image), replicated.
    target: keypoints the extractor aims at per image; capacity: candidates a cloud pair may hold (0 = the library default, 16 per point)"""
    from srrg2_proslam_amd import configs  # noqa: F401
    z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ref_kitti.npz"))
    left = [im for im in z["city_left"]] + [im for im in z["highway_left"]]
    right = [im for im in z["city_right"]] + [im for im in z["highway_right"]]
    dev = torch.device("cuda", 0)
    ctx = ops.Context(0)
    ctx.use_torch_stream()
    if dense is not None:
        ctx.set_bruteforce_dense_phase(dense)
    img = torch.from_numpy(np.stack(left + right)).to(dev)
    n_img, stride = img.shape[0], 1024 if target <= 1000 else 2048
    kp = torch.zeros((n_img, stride, 2), dtype=torch.float32, device=dev)
    desc = torch.zeros((n_img, stride, 32), dtype=torch.uint8, device=dev)
    n = torch.zeros((n_img,), dtype=torch.int32, device=dev)
    st = torch.zeros((n_img,), dtype=torch.int32, device=dev)
    ops.extract_features_batch(ctx, ops.extractor_params(target=target, selection_order=ops.SELECT_LIBSTDCXX), img, kp, desc, n, st)
    torch.cuda.synchronize()
    pairs_n = len(left)
    clouds = ops.BruteforceClouds(0, B, stride, stride, candidate_capacity=capacity)
    sel = torch.arange(B, device=dev) % pairs_n
    clouds.fixed_desc.copy_(desc[sel])
    clouds.moving_desc.copy_(desc[sel + pairs_n])
    clouds.n_fixed.copy_(n[sel])
    clouds.n_moving.copy_(n[sel + pairs_n])
    p = ops.bruteforce_params(max_dist, 0.9)
    ops.bruteforce_match_batch(ctx, p, clouds)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.bruteforce_match_batch(ctx, p, clouds)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    pairs = float((clouds.n_fixed.double() * clouds.n_moving.double()).sum().item())
    matches, ok = clouds.n_matches.float().mean().item(), float((clouds.status >= 0).float().mean().item())
    if not quiet:
        print("REAL descriptors: B=%d, %.0f x %.0f points per pair on average, max_dist=%.0f matches/pair=%.0f, %.0f %% of the pairs within capacity: "
              "%.3f ms/launch, %.2f G pairs/s" % (B, clouds.n_fixed.float().mean().item(), clouds.n_moving.float().mean().item(), max_dist, matches, 100 * ok,
                                                 ms, pairs / ms / 1e6))
    clouds_nf, clouds_nm = clouds.n_fixed.float().mean().item(), clouds.n_moving.float().mean().item()
    ctx.close()
    del clouds
    torch.cuda.empty_cache()
    return {"cloud_pairs_per_launch": B, "descriptors": "real (KITTI stereo pairs, our extractor)", "maximum_descriptor_distance": max_dist,
            "points_per_cloud": [clouds_nf, clouds_nm],
            "matches_per_cloud_pair": matches, "cloud_pairs_within_capacity": ok, "ms_per_launch": ms, "descriptor_pairs_per_launch": pairs,
            "pairs_per_s": pairs / (ms * 1e-3)}


def cpu_port_real(max_dist=50.0, target=1000, repeats=3):
    """the CPU checker (oracle/, a single-threaded restatement of the reference's compute()) timed on the seven real stereo pairs'
    descriptors, and the device's matches for the same pairs compared with its output (bench.py's cpu_baseline of the f4 row)"""
    import time
    from oracle import binding as ob
    from tests import helpers as hp
    z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ref_kitti.npz"))
    left = [im for im in z["city_left"]] + [im for im in z["highway_left"]]
    right = [im for im in z["city_right"]] + [im for im in z["highway_right"]]
    dev = torch.device("cuda", 0)
    ctx = ops.Context(0)
    ctx.use_torch_stream()
    img = torch.from_numpy(np.stack(left + right)).to(dev)
    n_img, stride = img.shape[0], 1024 if target <= 1000 else 2048
    kp = torch.zeros((n_img, stride, 2), dtype=torch.float32, device=dev)
    desc = torch.zeros((n_img, stride, 32), dtype=torch.uint8, device=dev)
    n = torch.zeros((n_img,), dtype=torch.int32, device=dev)
    st = torch.zeros((n_img,), dtype=torch.int32, device=dev)
    ops.extract_features_batch(ctx, ops.extractor_params(target=target, selection_order=ops.SELECT_LIBSTDCXX), img, kp, desc, n, st)
    torch.cuda.synchronize()
    pairs_n = len(left)
    clouds = ops.BruteforceClouds(0, pairs_n, stride, stride, candidate_capacity=stride * 64)
    clouds.fixed_desc.copy_(desc[:pairs_n])
    clouds.moving_desc.copy_(desc[pairs_n:])
    clouds.n_fixed.copy_(n[:pairs_n])
    clouds.n_moving.copy_(n[pairs_n:])
    ops.bruteforce_match_batch(ctx, ops.bruteforce_params(max_dist, 0.9), clouds)
    ctx.synchronize()
    hd, hn = desc.cpu().numpy(), n.cpu().numpy()
    equal, scored, t_best = True, 0.0, None
    for _ in range(repeats):
        t0 = time.perf_counter()
        refs = [ob.bruteforce_match(hd[k, : hn[k]], hd[k + pairs_n, : hn[k + pairs_n]], max_dist, 0.9) for k in range(pairs_n)]
        dt = time.perf_counter() - t0
        t_best = dt if t_best is None or dt < t_best else t_best
    for k in range(pairs_n):
        scored += float(hn[k]) * float(hn[k + pairs_n])
        equal = equal and hp.corr_equal(refs[k][0], clouds.matches_of(k)) and refs[k][1] == int(clouds.status[k].item())
    ctx.close()
    return {"value": scored / t_best, "unit": "descriptor pairs/s", "cores": 1, "kind": "port",
            "sample": "the seven real stereo pairs (%.0f x %.0f points on average), best of %d runs: %.1f ms per cloud pair" % (
                float(hn[:pairs_n].mean()), float(hn[pairs_n:].mean()), repeats, 1e3 * t_best / pairs_n),
            "ms_per_cloud_pair": 1e3 * t_best / pairs_n, "device_matches_equal_checker": bool(equal)}


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "cpu":
        print(cpu_port_real())
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "real":
        run_real(int(sys.argv[2]) if len(sys.argv) > 2 else 1024, float(sys.argv[3]) if len(sys.argv) > 3 else 50.0,
                 target=int(sys.argv[4]) if len(sys.argv) > 4 else 1000, capacity=int(sys.argv[5]) if len(sys.argv) > 5 else 0)
        sys.exit(0)
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    md = float(sys.argv[3]) if len(sys.argv) > 3 else 50.0
    run(B, N, md)
    run(B, N // 2, md)
    run(8, N, md)
