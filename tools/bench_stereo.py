"""quick device-resident timing of the stereo matcher kernel (development tool)"""
import sys, time
import numpy as np
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from srrg2_proslam_amd import configs, ops, synthetic as syn

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
uniq = 61  # a prime: frame b runs on XCD b mod 8 (see bench.py --unique)
cfg = configs.get("kitti")
frames = ops.StereoFrames(0, B, N, epilogue=True)
t0 = time.time()
host = []
for b in range(uniq):
    rng = np.random.default_rng(syn.seed_for(1, b))
    host.append(syn.stereo_frame(rng, cfg, N, visible_fraction=0.36))
for b in range(uniq):
    fr = host[b]
    frames.upload(b, fr["uv_left"], fr["desc_left"], fr["uv_right"], fr["desc_right"])
# replicate unique frames with a per-frame permutation-free copy (content identical, addresses distinct)
for b in range(uniq, B):
    s = b % uniq
    frames.left_kp[b] = frames.left_kp[s]; frames.right_kp[b] = frames.right_kp[s]
    frames.left_desc[b] = frames.left_desc[s]; frames.right_desc[b] = frames.right_desc[s]
    frames.n_left[b] = frames.n_left[s]; frames.n_right[b] = frames.n_right[s]
print("gen+upload %.1fs" % (time.time() - t0))
ctx = ops.Context(0)
ctx.use_torch_stream()
sp = ops.stereo_params(cfg["stereo_matcher"], cfg["camera"]["rows"], cfg["camera"]["cols"] if os.environ.get("BINNED") == "1" else 0)
tp = ops.triangulator_params(cfg)
for epi in (None, tp):
    for _ in range(3):
        ops.stereo_match_batch(ctx, sp, frames, epi)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iters = 20
    e0.record()
    for _ in range(iters):
        ops.stereo_match_batch(ctx, sp, frames, epi)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    M = frames.n_matches.float().mean().item()
    bytes_frame = 40 * (2 * N) + 12 * M
    print("epilogue=%s B=%d N=%d M=%.0f: %.3f ms/launch, %.2f us/frame, %.2f Mfps, %.1f GB/s algorithmic (%.1f%% of 8 TB/s)" % (
        epi is not None, B, N, M, ms, ms * 1e3 / B, B / ms / 1e3, B * bytes_frame / ms / 1e6, B * bytes_frame / ms / 1e6 / 80.0))
