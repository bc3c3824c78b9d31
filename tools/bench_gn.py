"""Per-kernel times of the headline step alone (matcher, search and Gauss-Newton launches by round), a few seconds on the GPU box.
    python tools/bench_gn.py [batch]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, bench
from srrg2_proslam_amd import configs, synthetic as syn
cfg = configs.get("kitti")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 18432
w = bench.FrameWorkload(cfg, 0, B, 2000, 2000, 896, 251, syn.seed_for(1,0))
for _ in range(2): w.step()
torch.cuda.synchronize()
kt = w.kernel_times(3)
print("gn_ms %.3f search_ms %.3f matcher %.3f rounds %s %s" % (kt["gn_ms"], kt["search_ms"], kt["matcher_ms"], [round(x,2) for x in kt["gn_ms_by_round"]], [round(x,2) for x in kt["search_ms_by_round"]]))
print(w.check(w.snapshot()))
