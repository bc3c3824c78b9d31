#!/usr/bin/env python3
"""experiment: the same step on ONE context (B frames) versus TWO contexts (B/2 frames each) driven by two host threads, so that
the search kernel of one half overlaps the Gauss-Newton kernel of the other"""
import os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from srrg2_proslam_amd import configs, synthetic as syn

def run(n_ctx, B, steps=6):
    cfg = configs.get("kitti")
    ws = [bench.FrameWorkload(cfg, 0, B // n_ctx, 2000, 2000, 896, 16, syn.seed_for(1, 0) + 1000 * i) for i in range(n_ctx)]
    streams = [torch.cuda.Stream() for _ in ws]
    def worker(w, s, n):
        with torch.cuda.stream(s):
            w.ctx.use_torch_stream()
            for _ in range(n):
                w.step()
            s.synchronize()
    for w, s in zip(ws, streams):
        worker(w, s, 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=worker, args=(w, s, steps)) for w, s in zip(ws, streams)]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for w in ws: w.close()
    return B * steps / dt, dt / steps * 1e3

for n in (1, 2, 3):
    fps, ms = run(n, 18432 if n != 3 else 18432)
    print("contexts %d: %.0f frames/s, %.2f ms per step of 18432 frames" % (n, fps, ms))
