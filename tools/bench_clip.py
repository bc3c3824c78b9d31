"""scene clipper throughput: B local maps of N points resident in HBM -> achieved algorithmic GB/s
usage: python tools/bench_clip.py [B] [N]   (prints one line per launch shape)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from srrg2_proslam_amd import _lib, configs, ops  # noqa: E402


def run(B, N, with_desc=True, iters=20, quiet=False):
    ctx = ops.Context(0)
    ctx.use_torch_stream()
    cam = configs.get("kitti")["camera"]
    proj = _lib.Projector(cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["cols"], cam["rows"], 0.1, 1000.0)
    scenes = ops.ClipScenes(0, B, N, with_desc=with_desc)
    g = torch.Generator(device="cuda").manual_seed(1)
    # points in a box around the frustum: roughly half survive
    xyz = torch.rand((B, N, 4), device="cuda", generator=g)
    xyz[..., 0] = (xyz[..., 0] - 0.5) * 60
    xyz[..., 1] = (xyz[..., 1] - 0.5) * 16
    xyz[..., 2] = xyz[..., 2] * 50 + 2
    scenes.scene_xyzw.copy_(xyz)
    if with_desc:
        scenes.scene_desc.copy_(torch.randint(0, 256, (B, N, 32), device="cuda", dtype=torch.uint8, generator=g))
    scenes.n_scene.fill_(N)
    I4 = np.eye(4, dtype=np.float32)
    for _ in range(3):
        ops.scene_clip_batch(ctx, proj, I4, scenes)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.scene_clip_batch(ctx, proj, I4, scenes)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    kept = scenes.n_clipped.float().mean().item()
    per_in, per_out = (48 if with_desc else 16), (52 if with_desc else 20)
    bytes_ = B * (N * per_in + kept * per_out + 64)
    if not quiet:
        print("B=%d N=%d desc=%d kept=%.0f (%.0f%%): %.3f ms/launch, %.1f GB/s algorithmic (%.1f%% of 8 TB/s)" % (
            B, N, with_desc, kept, 100 * kept / N, ms, bytes_ / ms / 1e6, 100 * bytes_ / ms / 1e6 / 8000))
    ctx.close()
    del scenes
    torch.cuda.empty_cache()
    return {"scenes_per_launch": B, "points_per_scene": N, "kept_per_scene": kept, "ms_per_launch": ms, "algorithmic_bytes_per_launch": bytes_,
            "bytes_per_point_in": per_in, "bytes_per_point_out": per_out, "gbps": bytes_ / ms / 1e6}


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    run(B, N, True)
    run(B, 8 * N, True)
    run(4, 4_000_000, True)   # few long scenes: tile-parallel shape
    run(B, N, False)
