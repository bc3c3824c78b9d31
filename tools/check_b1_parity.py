"""frame by frame: the host-handle aligner (prs_pcf_align, finder carried across frames) against the CPU checker on bench frames"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import binding as ob  # noqa: E402
from srrg2_proslam_amd import configs, ops, synthetic as syn  # noqa: E402

cfg = configs.get("kitti")
frames = bench.make_unique_frames(cfg, 12, 2000, 2000, syn.seed_for(1, 0))
sp, tp, pp, ap = bench.oracle_params(cfg)
ctx = ops.Context(0)
hf = ops.ProjectiveFinder(ctx, ops.pcf_params(cfg))
of = ob.ProjectiveFinder(pp)
for stop in (1, 0):
    hap = ops.aligner_params(cfg, stop_at_fixed_point=stop)
    for k, d in enumerate(frames):
        fr, mp = d["fr"], d["mp"]
        corr, _ = ob.stereo_match(fr["uv_left"], fr["desc_left"], fr["uv_right"], fr["desc_right"], sp)
        fixed, src = ob.stereo_assemble(fr["uv_left"], fr["uv_right"], corr)
        fdesc = fr["desc_left"][src]
        scale = ob.info_scale_from_nopt(mp["n_opt"])
        ap.mean_disparity = ob.mean_disparity(fixed)
        of.set_fixed(fixed, fdesc)
        of.set_moving(mp["xyz"], mp["desc"])
        res, oc = ob.align_frame(of, ap, fixed, mp["xyz"], scale, d["X0"])
        hf.set_fixed(fixed, fdesc)
        hf.set_moving(mp["xyz"], mp["desc"], scale)
        X, hc, hres, rc = hf.align(hap, d["X0"])
        Xo = np.array(res.X, np.float32).reshape(4, 4)
        same = np.array_equal(X.view(np.uint32), Xo.view(np.uint32))
        print("stop", stop, "frame", k, "equal" if same else "DIFF %.2e" % (np.linalg.norm(X - Xo) / np.linalg.norm(Xo)), "corr", len(hc), len(oc),
              "executed", hres.iterations_executed, "radius", hf.search_radius, of.search_radius, "md", hres.mean_disparity, ap.mean_disparity)
