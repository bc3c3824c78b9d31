"""randomised parity sweep of the projective finder + Gauss-Newton aligner through the batched device entry point
(the split search / GN pipeline the bench runs): random dataset configuration, cloud sizes, search pattern, finder and
aligner parameters, initial-guess error, LDS sizing bound.  Every frame is compared with the CPU oracle (test
infrastructure): correspondences bit for bit, pose bits, result and finder state.
usage: python tools/fuzz_align.py [batches] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run(batches, seed, ctx=None, oracle=None, verbose=True, frames_per_batch=6):
    import torch
    from helpers import aligner_params as oracle_aligner_params, corr_equal, make_align_case, pcf_params_from_cfg, rel_frobenius
    from srrg2_proslam_amd import ops
    if oracle is None:
        from oracle import binding as oracle
        oracle.lib()
    own = ctx is None
    if own:
        ctx = ops.Context(0)
    ctx.use_torch_stream()
    rng = np.random.default_rng(seed)
    bad, frames_done, corr_total, worst = [], 0, 0, 0.0
    for bi in range(batches):
        cfg_name = str(rng.choice(["kitti", "kitti", "euroc", "tum", "icl"]))
        fkw = {"search_type": int(rng.integers(0, 4))}
        if rng.random() < 0.5:
            fkw["maximum_search_radius_pixels"] = int(rng.choice([20, 50, 100]))
            fkw["minimum_search_radius_pixels"] = int(rng.choice([5, 10, 20]))
            fkw["search_radius_step_size_pixels"] = int(rng.choice([5, 10]))
        if rng.random() < 0.5:
            fkw["maximum_descriptor_distance"] = float(rng.choice([35.0, 75.0, 100.0]))
            fkw["minimum_descriptor_distance"] = float(rng.choice([15.0, 25.0, 30.0]))
            fkw["maximum_distance_ratio_to_second_best"] = float(rng.choice([0.6, 0.8, 0.95]))
        if rng.random() < 0.3:
            fkw["number_of_solver_iterations_per_projection"] = int(rng.choice([1, 3, 5, 10]))
            fkw["minimum_number_of_iterations"] = int(rng.choice([1, 5, 8]))
        akw = {}
        if rng.random() < 0.4:
            akw["max_iterations"] = int(rng.choice([1, 7, 30, 100]))
        if rng.random() < 0.3:
            akw["chi_threshold"] = float(rng.choice([1.0, 10.0, 100.0]))
        if rng.random() < 0.3:
            akw["damping"] = float(rng.choice([0.0, 1.0, 100.0]))
        cases = []
        for b in range(frames_per_batch):
            n_kp = int(rng.choice([12, 60, 200, 500, 900, 1300]))
            n_mv = int(rng.choice([8, 50, 300, 700, 1000]))
            sig = float(rng.choice([0.0, 0.05, 0.05, 0.3]))
            cases.append(make_align_case(cfg_name, int(rng.integers(1 << 30)), n_kp, n_mv, sigma_t=sig, sigma_r=sig * 0.06))
        cfg = cases[0][0]
        fs = max(max(len(c[1]) for c in cases), 1)
        ms = max(max(len(c[3]["xyz"]) for c in cases), 1)
        fr = ops.AlignFrames(0, len(cases), fs, ms)
        fr.max_fixed = int(rng.choice([0, 0, fs, ((fs + 255) // 256) * 256]))
        for b, (_, fixed, dfix, mp, T, X0) in enumerate(cases):
            fr.upload(b, fixed, dfix, mp["xyz"], oracle.info_scale_from_nopt(mp["n_opt"]), mp["desc"], X0)
        stop = int(rng.integers(0, 2))
        ops.align_batch(ctx, ops.pcf_params(cfg, **fkw), ops.aligner_params(cfg, stop_at_fixed_point=stop, **akw), fr)
        torch.cuda.synchronize()
        for b, (_, fixed, dfix, mp, T, X0) in enumerate(cases):
            scale = oracle.info_scale_from_nopt(mp["n_opt"])
            of = oracle.ProjectiveFinder(pcf_params_from_cfg(oracle, cfg, **fkw))
            of.set_fixed(fixed, dfix)
            of.set_moving(mp["xyz"], mp["desc"])
            md = oracle.mean_disparity(fixed) if fixed.shape[1] == 4 else 0.0
            res, rcorr = oracle.align_frame(of, oracle_aligner_params(oracle, cfg, mean_disparity=md, **akw), fixed, mp["xyz"], scale, X0)
            Xr = np.array(res.X, np.float32)
            Xg = fr.X[b].cpu().numpy()
            gres, st = fr.result_of(b), fr.state_of(b)
            ok = (corr_equal(rcorr, fr.corr_of(b)) and np.array_equal(Xr.view(np.uint32), Xg.view(np.uint32))
                  and gres.status == res.status and gres.warnings == res.warnings and gres.num_inliers == res.num_inliers
                  and gres.num_correspondences == res.num_correspondences
                  and int(st.search_radius_pixels) == of.search_radius and bool(st.has_converged) == of.has_converged
                  and int(st.current_iteration) == of.iteration)
            if np.all(np.isfinite(Xr)) and np.all(np.isfinite(Xg)):
                worst = max(worst, rel_frobenius(Xg.reshape(4, 4), Xr.reshape(4, 4)))
            frames_done += 1
            corr_total += len(rcorr)
            if not ok:
                bad.append((bi, b, cfg_name, fkw, akw, len(fixed), len(mp["xyz"]), fr.max_fixed, len(rcorr), len(fr.corr_of(b))))
                if verbose:
                    print("MISMATCH batch %d frame %d %s %s %s nf %d nm %d max_fixed %d: %d vs %d correspondences" % bad[-1])
    if own:
        ctx.close()
    if verbose:
        print("%d frames in %d batches, %d mismatches, %d correspondences compared, worst pose rel. Frobenius %.3g (seed %d)" % (
            frames_done, batches, len(bad), corr_total, worst, seed))
    return bad, corr_total


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    s = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    sys.exit(1 if run(n, s)[0] else 0)
