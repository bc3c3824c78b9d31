"""From pixels to poses on the device: rectified stereo images -> feature extraction (left, right) -> epipolar matcher
(+ adaptor / triangulator) -> scene clipper -> projective finder + GN aligner -> pose update -> merger, every row of
SURVEY.md section 8 (a) and (f) chained on device buffers, against the same chain on the CPU oracle.  The scene is a
static layered world seen from a camera stepping sideways by a quarter baseline per frame."""
import numpy as np
import pytest
import torch

from helpers import aligner_params as oracle_aligner_params, corr_equal, oracle_stereo_params, pcf_params_from_cfg
from oracle import binding_features as of
from oracle import binding_mapping as om
from srrg2_proslam_amd import configs, ops, synthetic as syn
from tests.test_mapping_gpu import _assert_map_equal, _gpu_params
from tests.test_oracle_mapping import merger_params as oracle_merger_params

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def test_images_to_poses_matches_the_oracle_chain(oracle, hip_ctx):
    cfg = configs.get("kitti")
    cam = cfg["camera"]
    K = (cam["fx"], cam["fy"], cam["cx"], cam["cy"])
    B, n_frames, stride, cap, max_meas = 2, 4, 1024, 3000, 8
    seqs = [syn.stereo_image_sequence(np.random.default_rng(70 + b), cfg, n_frames) for b in range(B)]
    step = seqs[0][1]
    est = om.estimator_params(om.EST_SMOOTHER, 4, K, max_dist2=100.0, chi2_delta=1e-6)
    po = oracle_merger_params(cfg, om.MERGER_STEREO_TRIANGULATION, est, max_appearance=100.0, target_merges=10 ** 6)
    pg = _gpu_params(po)
    hip_ctx.use_torch_stream()
    dev = torch.device("cuda", 0)

    sf = ops.StereoFrames(0, B, stride, epilogue=True)
    maps = ops.MapBatch(0, B, cap, max_meas, n_frames + 1, stride, stride)
    maps.measurement, maps.measurement_desc, maps.n_measured = sf.fixed_uvuv, sf.fixed_desc, sf.n_fixed
    clip = ops.ClipScenes(0, B, cap)
    clip.scene_xyzw, clip.scene_desc, clip.n_scene, clip.scene_n_opt = maps.coords, maps.desc, maps.n_points, maps.n_opt
    af = ops.AlignFrames(0, B, stride, cap)
    af.fixed, af.fixed_desc, af.n_fixed = sf.fixed_uvuv, sf.fixed_desc, sf.n_fixed
    af.moving, af.moving_desc, af.n_moving = clip.clipped_xyzw, clip.clipped_desc, clip.n_clipped
    maps.corr, maps.corr_from_aligner, maps.scene_index_map = af.corr, 1, clip.global_indices
    state0 = af.state.clone()
    pose = torch.eye(4, dtype=torch.float32, device=dev).repeat(B, 1, 1).contiguous()
    eye16 = torch.eye(4, dtype=torch.float32, device=dev).reshape(1, 16).repeat(B, 1).contiguous()
    zero_corr = torch.zeros((B,), dtype=torch.int32, device=dev)
    ep = ops.extractor_params()
    sp, tp = ops.stereo_params(cfg["stereo_matcher"], cam["rows"]), ops.triangulator_params(cfg)
    pp, apar = ops.pcf_params(cfg), ops.aligner_params(cfg)
    st = torch.zeros((B,), dtype=torch.int32, device=dev)
    I4 = np.eye(4, dtype=np.float32)

    omaps = [om.Map(cap, max_meas) for _ in range(B)]
    oposes = [om.pose_table(n_frames + 1) for _ in range(B)]
    opose = [I4.copy() for _ in range(B)]
    eo = of.extractor_params()

    for k in range(n_frames):
        lefts = torch.from_numpy(np.stack([seqs[b][0][k][0] for b in range(B)])).to(dev)
        rights = torch.from_numpy(np.stack([seqs[b][0][k][1] for b in range(B)])).to(dev)
        # device chain
        ops.extract_features_batch(hip_ctx, ep, lefts, sf.left_kp, sf.left_desc, sf.n_left, st)
        ops.extract_features_batch(hip_ctx, ep, rights, sf.right_kp, sf.right_desc, sf.n_right, st)
        ops.stereo_match_batch(hip_ctx, sp, sf, tp)
        if k > 0:
            clip.robot_in_local_map.copy_(pose)
            af.state.copy_(state0)
            af.X.copy_(eye16)
            af.n_corr.zero_()
            ops.scene_clip_batch(hip_ctx, pp.projector, I4, clip)
            ops.align_batch(hip_ctx, pp, apar, af)
            ops.pose_compose_batch(hip_ctx, clip.robot_in_local_map, af.X, pose)
            maps.n_corr = af.n_corr
        else:
            maps.n_corr = zero_corr
        maps.measurement_in_world.copy_(pose)
        maps.measurement_in_scene.copy_(pose)
        maps.frame.fill_(k)
        ops.merge_batch(hip_ctx, pg, maps)
        torch.cuda.synchronize()
        # oracle chain
        for b in range(B):
            l, r = seqs[b][0][k]
            uvl, _, dl = of.extract_features(eo, l)
            uvr, _, dr = of.extract_features(eo, r)
            corr, _ = oracle.stereo_match(uvl, dl, uvr, dr, oracle_stereo_params(oracle, cfg["stereo_matcher"]))
            assert corr_equal(corr, sf.matches_of(b)), (k, b, "stereo matches")
            fixed, src = oracle.stereo_assemble(uvl, uvr, corr)
            fdesc = dl[src]
            c, imap = np.zeros(0, oracle.CORR_DTYPE), None
            m = omaps[b]
            if k > 0:
                xyzw = m.coords[: m.n_points].copy()
                xyzw[:, 3] = oracle.info_scale_from_nopt(m.n_opt[: m.n_points])
                cx, cd, gi, _ = oracle.scene_clip(pcf_params_from_cfg(oracle, cfg).projector, opose[b], I4, xyzw, m.desc[: m.n_points])
                f = oracle.ProjectiveFinder(pcf_params_from_cfg(oracle, cfg))
                f.set_fixed(fixed, fdesc)
                f.set_moving(cx[:, :3], cd)
                res, rc = oracle.align_frame(f, oracle_aligner_params(oracle, cfg, mean_disparity=oracle.mean_disparity(fixed)), fixed, cx[:, :3], cx[:, 3], I4)
                f.close()
                assert corr_equal(rc, af.corr_of(b)), (k, b, "aligner correspondences")
                assert res.status == 1 and len(rc) > 60
                opose[b] = oracle.se3_mul(opose[b], oracle.se3_inverse(np.array(res.X, np.float32).reshape(4, 4)))
                assert np.array_equal(_bits(opose[b]).ravel(), _bits(pose[b].cpu().numpy()).ravel()), (k, b, "pose")
                c = rc.copy()
                c["fixed_idx"], c["moving_idx"] = rc["moving_idx"], rc["fixed_idx"]
                imap = np.concatenate([gi, np.zeros(cap - len(gi), np.int32)])
            rcode, mres = om.merge(po, opose[b], opose[b], oposes[b], k, m, fixed, fdesc, c, imap)
            assert rcode == 0
            got = maps.result[b].cpu().numpy()
            assert (int(got[0]), int(got[1]), int(got[2])) == (mres.n_merged, mres.n_added, mres.flags), (k, b)
            _assert_map_equal(maps, b, m, oposes[b], k + 1)
            # the camera steps sideways by a quarter baseline per frame
            truth = np.array([k * step, 0.0, 0.0], np.float32)
            assert np.linalg.norm(opose[b][:3, 3] - truth) < 0.03, (k, b, opose[b][:3, 3], truth)
    assert all(m.n_points > 300 for m in omaps)
