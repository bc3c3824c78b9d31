"""randomised parity sweep of the stereo matcher: random frame shapes (keypoint counts, image heights, row
crowding, duplicate pixels, ragged left/right), random matcher parameters; every case is compared bit for bit
with the CPU oracle (test infrastructure).  usage: python tools/fuzz_matcher.py [cases] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))


def random_case(rng):
    from srrg2_proslam_amd import synthetic as syn
    n_l = int(rng.choice([1, 3, 17, 64, 200, 700, 1024, 1025, 1500, 2000, 2048, 2500]))
    n_r = n_l if rng.random() < 0.5 else int(rng.integers(1, 2049 if n_l <= 2048 else 3000))
    rows = int(rng.choice([8, 40, 100, 376, 480, 1000]))
    cols = int(rng.choice([64, 300, 1241, 5000]))
    n_src = max(n_l, n_r)
    # landmarks: (row, col) with optional crowding on few rows / few columns
    r_span = rows if rng.random() < 0.7 else max(1, rows // int(rng.integers(4, 40)))
    c_span = cols if rng.random() < 0.7 else max(1, cols // int(rng.integers(4, 40)))
    row = rng.integers(0, r_span, n_src)
    col = rng.integers(0, c_span, n_src)
    disp = rng.integers(0, int(rng.choice([1, 20, 120])), n_src)
    frac = rng.random((n_src, 4)).astype(np.float32) * (0.99 if rng.random() < 0.5 else 0.0)
    jitter = (rng.random(n_src) < rng.choice([0.0, 0.1, 0.5])) * rng.integers(-2, 3, n_src)
    uvl = np.stack([col + frac[:, 0], row + frac[:, 1]], axis=1).astype(np.float32)
    uvr = np.stack([np.maximum(col - disp, 0) + frac[:, 2], np.clip(row + jitter, 0, rows - 1) + frac[:, 3]], axis=1).astype(np.float32)
    n_proto = int(rng.choice([4, 50, n_src]))
    proto = syn.random_descriptors(rng, n_proto)
    pick = rng.integers(0, n_proto, n_src) if n_proto < n_src else np.arange(n_src)
    flip = float(rng.choice([0.0, 0.02, 0.1, 0.4]))
    dl = syn.flip_bits(rng, proto[pick], flip)
    dr = syn.flip_bits(rng, proto[pick], flip)
    pl, pr = rng.permutation(n_src)[:n_l], rng.permutation(n_src)[:n_r]
    fr = {"uv_left": uvl[pl], "desc_left": dl[pl], "uv_right": uvr[pr], "desc_right": dr[pr]}
    m = {"maximum_descriptor_distance": float(rng.choice([25.0, 75.0, 100.5, 255.0, 256.0, 300.0])),
         "maximum_distance_ratio_to_second_best": float(rng.choice([0.5, 0.8, 0.99, 1.5])),
         "minimum_matching_ratio": 0.3,
         "maximum_disparity_pixels": int(rng.choice([0, 10, 100, 400])),
         "epipolar_line_thickness_pixels": int(rng.choice([0, 0, 1, 2, 5]))}
    return fr, m, rows


def check_epilogue(oracle, ctx, fr, m, rows, ref, rflags):
    """batched device entry point with the fused stereo adaptor + triangulator: correspondences, kept rows,
    descriptor copies and triangulated points against the oracle chain"""
    import torch
    from helpers import corr_equal, oracle_tri_params
    from srrg2_proslam_amd import configs, ops
    cfg = configs.get("kitti")
    stride = max(len(fr["uv_left"]), len(fr["uv_right"]), 1)
    frames = ops.StereoFrames(0, 2, stride, epilogue=True)
    for b in range(2):  # the same pair twice: frame 1 also checks that nothing leaks between frames of a workgroup
        frames.upload(b, fr["uv_left"], fr["desc_left"], fr["uv_right"], fr["desc_right"])
    tp = ops.triangulator_params(cfg)
    ctx.use_torch_stream()
    ops.stereo_match_batch(ctx, ops.stereo_params(m, rows, 0), frames, tp)
    torch.cuda.synchronize()
    uvuv, src = oracle.stereo_assemble(fr["uv_left"], fr["uv_right"], ref)
    xyz, valid = oracle.triangulate(uvuv, oracle_tri_params(oracle, cfg))
    for b in range(2):
        nf = int(frames.n_fixed[b].item())
        g = frames.fixed_xyz[b, :nf].cpu().numpy()
        if not (corr_equal(ref, frames.matches_of(b)) and int(frames.status[b].item()) == rflags and nf == len(uvuv)
                and np.array_equal(frames.fixed_uvuv[b, :nf].cpu().numpy(), uvuv)
                and np.array_equal(frames.fixed_desc[b, :nf].cpu().numpy(), fr["desc_left"][src])
                and np.array_equal(g[:, :3].view(np.uint32), xyz.view(np.uint32))
                and np.array_equal(g[:, 3] != 0, valid.astype(bool))):
            return False
    return True


def run(cases, seed, ctx=None, oracle=None, verbose=True, epilogue=True):
    from helpers import corr_equal, oracle_stereo_params
    from srrg2_proslam_amd import ops
    if oracle is None:
        from oracle import binding as oracle
        oracle.lib()
    own = ctx is None
    if own:
        ctx = ops.Context(0)
    rng = np.random.default_rng(seed)
    bad, total_matches = [], 0
    for c in range(cases):
        fr, m, rows = random_case(rng)
        ref, rflags = oracle.stereo_match(fr["uv_left"], fr["desc_left"], fr["uv_right"], fr["desc_right"], oracle_stereo_params(oracle, m))
        got, gflags = ops.stereo_match(ctx, ops.stereo_params(m, rows, 0), fr["uv_left"], fr["desc_left"], fr["uv_right"], fr["desc_right"])
        total_matches += len(ref)
        ok_epilogue = True
        if epilogue and len(fr["uv_left"]) <= 2048 and len(fr["uv_right"]) <= 2048:
            ok_epilogue = check_epilogue(oracle, ctx, fr, m, rows, ref, rflags)
        if not (corr_equal(ref, got) and rflags == gflags and ok_epilogue):
            bad.append((c, len(fr["uv_left"]), len(fr["uv_right"]), rows, m, len(ref), len(got)))
            if verbose:
                print("MISMATCH case %d: nL %d nR %d rows %d %s: %d vs %d matches" % bad[-1])
    if own:
        ctx.close()
    if verbose:
        print("%d cases, %d mismatches, %d matches compared (seed %d)" % (cases, len(bad), total_matches, seed))
    return bad, total_matches


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    s = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    sys.exit(1 if run(n, s)[0] else 0)
