"""randomised parity sweep of the projective mergers + landmark estimators: random estimator
(weighted mean / smoother / stereo EKF), binning on / off with random bin grids and merge targets, frame sizes,
sequence lengths, measurement-history capacity, correspondence responses around the appearance threshold,
duplicate correspondences per scene point.  After every merged frame ALL map arrays are compared with the CPU
oracle (test infrastructure) bit for bit.   usage: python tools/fuzz_merge.py [sequences] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run(sequences, seed, ctx=None, verbose=True):
    from oracle import binding as ob
    from oracle import binding_mapping as om
    from srrg2_proslam_amd import configs, ops
    from tests import test_mapping_gpu as tm
    from tests.test_oracle_mapping import merger_params as oracle_merger_params, stereo_scene
    ob.lib()
    own = ctx is None
    if own:
        ctx = ops.Context(0)
    rng = np.random.default_rng(seed)
    cfg = configs.get("kitti")
    K = (cfg["camera"]["fx"], cfg["camera"]["fy"], cfg["camera"]["cx"], cfg["camera"]["cy"])
    bad, merged_total, frames_total = [], 0, 0
    for si in range(sequences):
        kind, variant, _ = tm.CASES[int(rng.integers(0, len(tm.CASES)))]
        max_meas = int(rng.choice([2, 4, 8, 12])) if kind == "smoother" else 0
        binning = int(rng.integers(0, 2))
        kw = {"enable_binning": binning, "target_merges": int(rng.choice([5, 50, 10 ** 6])),
              "row_bins": int(rng.choice([4, 20, 47])), "col_bins": int(rng.choice([6, 60, 155])),
              "max_appearance": float(rng.choice([25.0, 50.0, 70.0]))}
        depth = variant == om.MERGER_DEPTH_EKF  # RGB-D: (u, v, depth) measurements of the ICL camera
        fcfg = configs.get("icl") if depth else cfg
        fK = (fcfg["camera"]["fx"], fcfg["camera"]["fy"], fcfg["camera"]["cx"], fcfg["camera"]["cy"])
        po = oracle_merger_params(fcfg, variant, tm._estimator(kind, fK, (configs.baseline_pixels(cfg), 0.0)), **kw)
        pg = tm._gpu_params(po)
        n_frames = int(rng.integers(3, 8))
        n_kp = int(rng.choice([60, 200, 500, 900]))
        cap = 4000
        m = om.Map(cap, max_meas)
        poses = om.pose_table(n_frames + 1)
        maps = ops.MapBatch(0, 1, cap, max_meas, n_frames + 1, 1024, 1024)
        tm._upload_map(maps, 0, m, poses)
        prev, ok = None, True
        for k in range(n_frames):
            if depth:
                fr = tm.hp.syn.rgbd_frame(np.random.default_rng(int(rng.integers(1 << 30))), fcfg, n_kp)
                fixed, desc = fr["fixed"][:, :3].astype(np.float32), fr["desc_fixed"].copy()
            else:
                _, fixed, desc, xyz = stereo_scene(int(rng.integers(1 << 30)), n_kp=n_kp)
            fixed, desc = fixed[:1024], desc[:1024]
            if prev is not None:
                reuse = int(min(len(fixed), len(prev[1])) * rng.choice([0.2, 0.5, 0.9]))
                desc[:reuse] = prev[1][:reuse]
            Tw = tm._frame_pose(rng, k)
            Ts = Tw.copy()
            corr = np.zeros(0, ob.CORR_DTYPE)
            if k > 0:
                lut = {bytes(d): i for i, d in enumerate(desc)}
                pairs = []
                for s in range(m.n_points):
                    i = lut.get(bytes(m.desc[s]))
                    if i is not None:
                        pairs.append((s, i))
                pairs = pairs[: maps.corr_stride]
                corr = np.zeros(len(pairs), ob.CORR_DTYPE)
                corr["fixed_idx"] = [p[0] for p in pairs]
                corr["moving_idx"] = [p[1] for p in pairs]
                corr["response"] = rng.integers(0, 90, len(pairs)).astype(np.float32)
            rc, res = om.merge(po, Tw, Ts, poses, k, m, fixed, desc, corr)
            if rc != 0:
                break  # (scene full etc.: the loud-error paths have their own tests)
            tm._upload_frame(maps, 0, fixed, desc, corr, Tw, Ts, k)
            ops.merge_batch(ctx, pg, maps)
            ctx.synchronize()
            got = maps.result[0].cpu().numpy()
            try:
                assert (int(got[0]), int(got[1]), int(got[2])) == (res.n_merged, res.n_added, res.flags), "result"
                tm._assert_map_equal(maps, 0, m, poses, k + 1)
            except AssertionError as e:
                ok = False
                bad.append((si, k, kind, binning, kw, n_kp, max_meas, str(e)[:60]))
                if verbose:
                    print("MISMATCH sequence %d frame %d %s binning %d %s n_kp %d max_meas %d: %s" % bad[-1])
                break
            merged_total += res.n_merged
            frames_total += 1
            prev = (fixed, desc)
    if own:
        ctx.close()
    if verbose:
        print("%d sequences, %d frames merged, %d landmark merges compared, %d mismatches (seed %d)" % (
            sequences, frames_total, merged_total, len(bad), seed))
    return bad, merged_total


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    s = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    sys.exit(1 if run(n, s)[0] else 0)
