"""stereo matcher (+ fused adaptor / triangulator epilogue) alone: B frames resident in HBM ->
ms per launch, algorithmic GB/s, fraction of the 8 TB/s HBM peak, and a digest of every output so
that two kernel variants can be compared bit for bit on the bench inputs.
usage: python tools/bench_matcher.py [--batch B] [--keypoints N] [--no-epilogue] [--iters K]"""
import argparse
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from srrg2_proslam_amd import configs, ops, synthetic as syn  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=18432)
    ap.add_argument("--keypoints", type=int, default=2000)
    ap.add_argument("--unique", type=int, default=32)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--no-epilogue", action="store_true")
    ap.add_argument("--thickness", type=int, default=-1, help="override epipolar_line_thickness_pixels")
    args = ap.parse_args()
    cfg = configs.get("kitti")
    B, N = args.batch, args.keypoints
    epi = not args.no_epilogue
    frames = ops.StereoFrames(0, B, N, epilogue=epi)
    stage = ops.StereoFrames(0, args.unique, N, epilogue=False)
    for u in range(args.unique):
        rng = np.random.default_rng(syn.seed_for(1, 0) + u)
        fr = syn.stereo_frame(rng, cfg, N, visible_fraction=0.36)
        stage.upload(u, fr["uv_left"], fr["desc_left"], fr["uv_right"], fr["desc_right"])
    idx = torch.arange(B, device="cuda") % args.unique
    for name in ("left_kp", "right_kp", "left_desc", "right_desc", "n_left", "n_right"):
        getattr(frames, name).copy_(getattr(stage, name).index_select(0, idx))
    ctx = ops.Context(0)
    ctx.use_torch_stream()
    sm = dict(cfg["stereo_matcher"])
    if args.thickness >= 0:
        sm["epipolar_line_thickness_pixels"] = args.thickness
    sp = ops.stereo_params(sm, cfg["camera"]["rows"], 0)
    tp = ops.triangulator_params(cfg) if epi else None
    for _ in range(3):
        ops.stereo_match_batch(ctx, sp, frames, tp)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        ops.stereo_match_batch(ctx, sp, frames, tp)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.iters
    nm = frames.n_matches.cpu().numpy()
    h = hashlib.sha256()
    h.update(nm.tobytes())
    h.update(frames.status.cpu().numpy().tobytes())
    check = min(B, 2 * args.unique)
    m = frames.matches[:check].cpu().numpy()
    for b in range(check):
        h.update(m[b, :nm[b]].tobytes())
    n_fixed = 0.0
    if epi:
        nf = frames.n_fixed.cpu().numpy()
        n_fixed = float(nf.mean())
        h.update(nf.tobytes())
        uv, xyz, fd = frames.fixed_uvuv[:check].cpu().numpy(), frames.fixed_xyz[:check].cpu().numpy(), frames.fixed_desc[:check].cpu().numpy()
        for b in range(check):
            h.update(uv[b, :nf[b]].tobytes())
            h.update(xyz[b, :nf[b]].tobytes())
            h.update(fd[b, :nf[b]].tobytes())
    bytes_ = 40.0 * 2 * N + 12.0 * float(nm.mean()) + 29.0 * n_fixed
    gbps = B * bytes_ / ms / 1e6
    print("B=%d N=%d epilogue=%d matches=%.1f: %.4f ms/launch, %.0f GB/s algorithmic (%.1f%% of 8 TB/s), %.2f us/frame/CU, digest %s" % (
        B, N, epi, nm.mean(), ms, gbps, gbps / 80.0, ms * 1e3 * 256 / B, h.hexdigest()[:16]))
    ctx.close()


if __name__ == "__main__":
    main()
