#!/usr/bin/env python3
"""Generates the committed golden vectors (tests/golden/*.npz) from the CPU oracle on small seeded
inputs.  The reference holds no golden vectors for this path and cannot be
built here, so these fixtures pin the ORACLE's outputs: the CPU suite checks the oracle still
reproduces them, the GPU suite checks the HIP path against them without needing the oracle's code.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from helpers import aligner_params, make_align_case, oracle_stereo_params, oracle_tri_params, pcf_params_from_cfg  # noqa: E402
from oracle import binding as ob  # noqa: E402
from srrg2_proslam_amd import configs, synthetic as syn  # noqa: E402


def stereo_case(name, cfg_name, seed, n, thickness, jitter):
    cfg = configs.get(cfg_name)
    rng = np.random.default_rng(seed)
    fr = syn.stereo_frame(rng, cfg, n, row_jitter_fraction=jitter)
    m = dict(cfg["stereo_matcher"])
    m["epipolar_line_thickness_pixels"] = thickness
    corr, flags = ob.stereo_match(fr["uv_left"], fr["desc_left"], fr["uv_right"], fr["desc_right"], oracle_stereo_params(ob, m))
    uvuv, src = ob.stereo_assemble(fr["uv_left"], fr["uv_right"], corr)
    xyz, valid = ob.triangulate(uvuv, oracle_tri_params(ob, cfg))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), cfg_name=cfg_name, thickness=thickness,
                        uv_left=fr["uv_left"], desc_left=fr["desc_left"], uv_right=fr["uv_right"], desc_right=fr["desc_right"],
                        corr_fixed=corr["fixed_idx"], corr_moving=corr["moving_idx"], corr_response=corr["response"], flags=flags,
                        fixed_uvuv=uvuv, fixed_src=src, xyz=xyz, valid=valid)
    print(name, len(corr), "matches", len(uvuv), "fixed")


def align_case(name, cfg_name, seed, search):
    cfg, fixed, dfix, mp, T, X0 = make_align_case(cfg_name, seed, 220, 260)
    scale = ob.info_scale_from_nopt(mp["n_opt"])
    f = ob.ProjectiveFinder(pcf_params_from_cfg(ob, cfg, search_type=search))
    f.set_fixed(fixed, dfix)
    f.set_moving(mp["xyz"], mp["desc"])
    md = ob.mean_disparity(fixed) if fixed.shape[1] == 4 else 0.0
    ap = aligner_params(ob, cfg, mean_disparity=md)
    # first linearisation at the initial guess on the first correspondence set
    f.set_local_map_in_sensor(X0)
    corr0, _ = f.compute()
    lin = ob.linearize(ap, X0, corr0, fixed, mp["xyz"], scale)
    # full loop on a fresh finder
    f2 = ob.ProjectiveFinder(pcf_params_from_cfg(ob, cfg, search_type=search))
    f2.set_fixed(fixed, dfix)
    f2.set_moving(mp["xyz"], mp["desc"])
    res, corr = ob.align_frame(f2, ap, fixed, mp["xyz"], scale, X0)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), cfg_name=cfg_name, search=search,
                        fixed=fixed, desc_fixed=dfix, moving=mp["xyz"], desc_moving=mp["desc"], n_opt=mp["n_opt"], X0=X0, T=T,
                        corr0_fixed=corr0["fixed_idx"], corr0_moving=corr0["moving_idx"], corr0_response=corr0["response"],
                        H0=np.array(lin.H, np.float32), b0=np.array(lin.b, np.float32),
                        X=np.array(res.X, np.float32), corr_fixed=corr["fixed_idx"], corr_moving=corr["moving_idx"],
                        corr_response=corr["response"], status=res.status, num_inliers=res.num_inliers, warnings=res.warnings,
                        radius=f2.search_radius, dd=f2.descriptor_distance, iteration=f2.iteration, converged=f2.has_converged)
    print(name, len(corr), "correspondences, status", res.status)


if __name__ == "__main__":
    stereo_case("stereo_kitti_n200_t0", "kitti", 1001, 200, 0, 0.0)
    stereo_case("stereo_kitti_n200_t1", "kitti", 1002, 200, 1, 0.2)
    stereo_case("stereo_euroc_n150_t0", "euroc", 1003, 150, 0, 0.0)
    align_case("align_kitti_circle", "kitti", 2001, 2)
    align_case("align_euroc_square", "euroc", 2002, 1)
    align_case("align_tum_rhombus", "tum", 2003, 3)
    align_case("align_icl_kdtree", "icl", 2004, 0)
