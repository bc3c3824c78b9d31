"""bench.py's latency_b1 object alone (one frame at a time through the C++ adapters and the bare C-ABI, CPU checker beside it)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from srrg2_proslam_amd import configs, synthetic as syn  # noqa: E402

if __name__ == "__main__":
    cfg = configs.get("kitti")
    n_kp = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    frames = bench.make_unique_frames(cfg, 16, n_kp, n_kp, syn.seed_for(1, 0))
    print(json.dumps(bench.latency_b1(cfg, frames)))
