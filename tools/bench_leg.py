"""One side configuration of bench.py (EuRoC / TUM / KITTI-1000 shaped) alone, with its per-round kernel times.
    python tools/bench_leg.py tum|euroc|kitti_n1000|kitti_real|streamed|small_batch [batch]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from srrg2_proslam_amd import configs, synthetic as syn  # noqa: E402

LEGS = {"euroc": ("euroc", 1000, 1000, 512, 2), "tum": ("tum", 1000, 1000, 1024, 3), "kitti_n1000": ("kitti", 1000, 1000, 512, 1)}

if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "tum"
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 6144
    if name == "kitti_real":
        print(json.dumps(bench.kitti_real_leg(0, configs.get("kitti"), batch)))
    elif name in ("streamed", "small_batch"):
        cfg = configs.get("kitti")
        frames = bench.make_unique_frames(cfg, 13, 2000, 2000, syn.seed_for(1, 0))
        if name == "streamed":
            print(json.dumps(bench.streamed_leg(cfg, frames, 2000, 2000, 896, batch, 0)))
        else:
            print(json.dumps(bench.small_batch_curve(cfg, frames, 2000, 2000, 896, 0)))
    else:
        cname, kp, mv, mf, cidx = LEGS[name]
        print(json.dumps(bench.small_config_leg(name, configs.get(cname), kp, mv, mf, batch, 0, syn.seed_for(cidx, 0) + 31)))
