"""feature extraction throughput: B KITTI-sized 8-bit images resident in HBM
usage: python tools/bench_features.py [B] [synthetic|kitti] [canonical|libstdcxx]
  kitti = the fourteen real KITTI frames the reference's tests hold (tests/golden/ref_kitti.npz, 1241 x 376: sequence 00 "city"
  pairs 0-4 and the two "highway" pairs), replicated over the batch; kitti.conf extractor settings (FAST 15 with non-maximum
  suppression, 1000 keypoints, 3 x 3 detector grid)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from srrg2_proslam_amd import configs, ops, synthetic as syn  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    source = sys.argv[2] if len(sys.argv) > 2 else "synthetic"
    order = sys.argv[3] if len(sys.argv) > 3 else "canonical"
    run(B, source, order)


def run(B, source="kitti", order="libstdcxx", quiet=False):
    cfg = configs.get("kitti")
    uniq = []
    if source == "kitti":
        z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ref_kitti.npz"))
        for key in ("city_left", "city_right", "highway_left", "highway_right"):
            uniq += [im for im in z[key]]
    else:
        for s in range(8):
            l, r, _ = syn.stereo_images(np.random.default_rng(100 + s), cfg)
            uniq += [l, r]
    dev = torch.device("cuda", 0)
    stage = torch.from_numpy(np.stack(uniq)).to(dev)
    img = stage[torch.arange(B, device=dev) % len(uniq)].contiguous()
    rows, cols = img.shape[1], img.shape[2]
    stride = 1024
    kp = torch.zeros((B, stride, 2), dtype=torch.float32, device=dev)
    desc = torch.zeros((B, stride, 32), dtype=torch.uint8, device=dev)
    n = torch.zeros((B,), dtype=torch.int32, device=dev)
    st = torch.zeros((B,), dtype=torch.int32, device=dev)
    ctx = ops.Context(0)
    ctx.use_torch_stream()
    p = ops.extractor_params(selection_order=ops.SELECT_LIBSTDCXX if order == "libstdcxx" else ops.SELECT_CANONICAL)
    for _ in range(2):
        ops.extract_features_batch(ctx, p, img, kp, desc, n, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    iters = 5
    for _ in range(iters):
        ops.extract_features_batch(ctx, p, img, kp, desc, n, st)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    assert int(st.min().item()) >= 0
    nf = n.float().mean().item()
    algo = B * (rows * cols + nf * 44)  # image read once + keypoints (8 B) + descriptors (32 B) + counters written
    if not quiet:
        print("source=%s (%d distinct images) selection=%s" % (source, len(uniq), order))
        print("B=%d images %dx%d, %.0f features/image: %.3f ms/launch, %.2f M images/s, %.1f GB/s algorithmic (%.1f%% of 8 TB/s)" % (
            B, cols, rows, nf, ms, B / ms / 1e3, algo / ms / 1e6, 100 * algo / ms / 1e6 / 8000))
    ctx.close()
    del img, kp, desc, stage
    torch.cuda.empty_cache()
    return {"images_per_launch": B, "image": "%dx%d" % (cols, rows), "source": source, "distinct_images": len(uniq), "selection_order": order,
            "features_per_image": nf, "ms_per_launch": ms, "images_per_s": B / (ms * 1e-3), "algorithmic_bytes_per_launch": float(algo),
            "gbps": float(algo) / ms / 1e6}


if __name__ == "__main__":
    main()
