import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from srrg2_proslam_amd import ops
z = np.load("tests/golden/ref_kitti.npz")
uniq = []
for key in ("city_left", "city_right", "highway_left", "highway_right"):
    uniq += [im for im in z[key]]
dev = torch.device("cuda", 0)
B = 4096
stage = torch.from_numpy(np.stack(uniq)).to(dev)
img = stage[torch.arange(B, device=dev) % len(uniq)].contiguous()
stride = 2048
kp = torch.zeros((B, stride, 2), dtype=torch.float32, device=dev)
desc = torch.zeros((B, stride, 32), dtype=torch.uint8, device=dev)
inten = torch.zeros((B, stride), dtype=torch.float32, device=dev)
n = torch.zeros((B,), dtype=torch.int32, device=dev); st = torch.zeros((B,), dtype=torch.int32, device=dev)
ctx = ops.Context(0); ctx.use_torch_stream()
p = ops.extractor_params(selection_order=ops.SELECT_LIBSTDCXX)
import inspect
for _ in range(3):
    ops.extract_features_batch(ctx, p, img, kp, desc, n, st, intensity=inten)
torch.cuda.synchronize()
d = inten[:, 1100:1100 + 256].reshape(B, 16, 16).cpu().numpy()
for b in (0, 1, 777, 3000):
    print("image", b)
    for w in range(16):
        t = d[b, w]
        print("  wave %2d total %7.0f push %6.0f (%2d) pop %6.0f (%2d) part %6.0f (%2d) small %6.0f (%2d)" % (w, t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7], t[8]))
m = d.mean(axis=(0, 1))
print("mean per wave: total %.0f push %.0f (%.1f) pop %.0f (%.1f) part %.0f (%.1f) small %.0f (%.1f)" % tuple(m[:9]))
