"""pairwise Hamming distances of REAL descriptors (our extractor on the KITTI stereo pairs the reference's tests hold) against uniform
random rows: what fraction of the left x right pairs of a stereo pair falls below a matcher threshold (the brute-force matcher's
candidate density).  usage (GPU box): python tools/study_descriptor_distances.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from srrg2_proslam_amd import ops  # noqa: E402


def main():
    z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ref_kitti.npz"))
    left = [im for im in z["city_left"]] + [im for im in z["highway_left"]]
    right = [im for im in z["city_right"]] + [im for im in z["highway_right"]]
    dev = torch.device("cuda", 0)
    ctx = ops.Context(0)
    ctx.use_torch_stream()
    img = torch.from_numpy(np.stack(left + right)).to(dev)
    n_img, stride = img.shape[0], 1024
    kp = torch.zeros((n_img, stride, 2), dtype=torch.float32, device=dev)
    desc = torch.zeros((n_img, stride, 32), dtype=torch.uint8, device=dev)
    n = torch.zeros((n_img,), dtype=torch.int32, device=dev)
    st = torch.zeros((n_img,), dtype=torch.int32, device=dev)
    ops.extract_features_batch(ctx, ops.extractor_params(selection_order=ops.SELECT_LIBSTDCXX), img, kp, desc, n, st)
    torch.cuda.synchronize()
    bits = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.uint8, device=dev)
    hist = torch.zeros(257, dtype=torch.float64)
    hist192 = torch.zeros(257, dtype=torch.float64)
    for k in range(len(left)):
        a = ((desc[k, : int(n[k])].unsqueeze(-1) & bits) != 0).reshape(int(n[k]), 256).float()
        b = ((desc[k + len(left), : int(n[k + len(left)])].unsqueeze(-1) & bits) != 0).reshape(int(n[k + len(left)]), 256).float()
        d = (a.sum(1, keepdim=True) + b.sum(1).unsqueeze(0) - 2 * a @ b.t()).round().long()
        hist += torch.bincount(d.flatten().cpu(), minlength=257).double()
        a3, b3 = a[:, :192], b[:, :192]
        d3 = (a3.sum(1, keepdim=True) + b3.sum(1).unsqueeze(0) - 2 * a3 @ b3.t()).round().long()
        hist192 += torch.bincount(d3.flatten().cpu(), minlength=257).double()
    tot = hist.sum().item()
    lv = torch.arange(257, dtype=torch.float64)
    mean = (hist * lv).sum().item() / tot
    sd = ((hist * (lv - mean) ** 2).sum().item() / tot) ** 0.5
    print("real descriptors, %d stereo pairs, %.0f left x right pairs: distance mean %.1f, sd %.1f (uniform random rows: 128, 8)" % (len(left), tot, mean, sd))
    for thr in (25, 33, 50, 75, 100):
        print("  below %3d: %.4f %% of the pairs on 256 bits, %.4f %% on the first 192 bits   (uniform random rows: ~0)" % (
            thr, 100 * hist[:thr].sum().item() / tot, 100 * hist192[:thr].sum().item() / tot))
    ctx.close()


if __name__ == "__main__":
    main()
