"""brute-force matcher throughput: B cloud pairs of N x N descriptors resident in HBM
usage: python tools/bench_bruteforce.py [B] [N] [max_dist]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from srrg2_proslam_amd import ops  # noqa: E402


def run(B, N, max_dist=50.0, iters=5, quiet=False):
    ctx = ops.Context(0)
    ctx.use_torch_stream()
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


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    md = float(sys.argv[3]) if len(sys.argv) > 3 else 50.0
    run(B, N, md)
    run(B, N // 2, md)
    run(8, N, md)
