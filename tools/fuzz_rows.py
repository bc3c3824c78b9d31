"""randomised parity sweeps of the remaining rows of SURVEY 8f against the CPU oracle (test infrastructure):
scene clipper (random clouds around / behind / outside the frustum, poses, sensor offsets, descriptors),
bijective brute-force matcher (tie-heavy descriptor pools, ragged sizes, thresholds) and intensity feature
extraction (random textures, image sizes, thresholds, grids, targets).
usage: python tools/fuzz_rows.py [cases per row] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def fuzz_clip(cases, rng, ctx, oracle):
    from srrg2_proslam_amd import _lib, ops
    from tests import helpers as hp
    po = hp.clip_projector(oracle)
    pg = _lib.Projector(po.fx, po.fy, po.cx, po.cy, po.canvas_cols, po.canvas_rows, po.range_min, po.range_max)
    bad = kept = 0
    for c in range(cases):
        n = int(rng.choice([0, 1, 63, 64, 65, 500, 2047, 2049, 7000, 40000]))
        xyzw = np.empty((n, 4), np.float32)
        xyzw[:, 0] = rng.uniform(-8, 8, n)
        xyzw[:, 1] = rng.uniform(-6, 6, n)
        xyzw[:, 2] = rng.uniform(-2, 12, n)
        xyzw[:, 3] = rng.uniform(1, 4, n)
        if n and rng.random() < 0.3:  # points exactly on the range / image borders
            xyzw[: n // 4, 2] = np.float32(rng.choice([po.range_min, po.range_max, 0.0]))
        desc = rng.integers(0, 256, (n, 32), dtype=np.uint8) if rng.random() < 0.6 else None
        R = hp.rot("y", rng.uniform(-0.6, 0.6)) @ hp.rot("x", rng.uniform(-0.4, 0.4)) @ hp.rot("z", rng.uniform(-3.2, 3.2) if rng.random() < 0.2 else 0.0)
        R[:3, 3] = rng.uniform(-1.0, 1.0, 3)
        S = np.eye(4, dtype=np.float32)
        if rng.random() < 0.5:
            S = hp.rot("z", rng.uniform(-0.2, 0.2))
            S[:3, 3] = rng.uniform(-0.3, 0.3, 3)
        ref = oracle.scene_clip(po, R.astype(np.float32), S.astype(np.float32), xyzw, desc)
        got = ops.scene_clip(ctx, pg, R.astype(np.float32), S.astype(np.float32), xyzw, desc)
        ok = (len(ref[0]) == len(got[0]) and np.array_equal(ref[0].view(np.uint32), got[0].view(np.uint32))
              and (ref[1] is None or np.array_equal(ref[1], got[1])) and np.array_equal(ref[2], got[2]) and ref[3] == got[3])
        kept += len(ref[0])
        if not ok:
            bad += 1
            print("CLIP MISMATCH case %d n %d: %d vs %d kept" % (c, n, len(ref[0]), len(got[0])))
    return bad, kept


def fuzz_bruteforce(cases, rng, ctx, oracle):
    from srrg2_proslam_amd import ops
    from tests import helpers as hp
    from tests.test_bruteforce_gpu import _tie_heavy
    bad = total = 0
    for c in range(cases):
        nf, nm = int(rng.choice([1, 2, 65, 300, 1000, 2100])), int(rng.choice([1, 3, 64, 500, 1500]))
        protos = int(rng.choice([3, 40, 5000]))
        seed = int(rng.integers(1 << 30))
        df = _tie_heavy(np.random.default_rng(seed), protos, nf, int(rng.integers(0, 12)))
        dm = _tie_heavy(np.random.default_rng(seed), protos, nm, int(rng.integers(0, 12)))
        max_dist, ratio = float(rng.choice([5.0, 20.0, 33.5, 50.0, 120.0])), float(rng.choice([0.5, 0.8, 0.95, 1.0, 1.5]))
        ref, rflags = oracle.bruteforce_match(df, dm, max_dist, ratio)
        clouds = ops.BruteforceClouds(0, 1, nf, nm, candidate_capacity=nf * nm)
        clouds.upload(0, df, dm)
        ctx.set_bruteforce_dense_phase((ops.BF_DENSE_MATRIX_WHEN_FULL, ops.BF_DENSE_POPCOUNT, ops.BF_DENSE_MATRIX)[c % 3])  # every kernel family
        ops.bruteforce_match_batch(ctx, ops.bruteforce_params(max_dist, ratio), clouds)
        ctx.synchronize()
        got, gflags = clouds.matches_of(0), int(clouds.status[0].item())
        total += len(ref)
        if not (hp.corr_equal(ref, got) and rflags == gflags):
            bad += 1
            print("BRUTEFORCE MISMATCH case %d nf %d nm %d protos %d dist %g ratio %g: %d vs %d" % (c, nf, nm, protos, max_dist, ratio, len(ref), len(got)))
    ctx.set_bruteforce_dense_phase(ops.BF_DENSE_MATRIX_WHEN_FULL)
    # full batches (more cloud pairs than half the CUs): the default takes the fused matrix-core shape
    for c in range(max(1, cases // 20)):
        B, fs, ms = 136, int(rng.choice([96, 700, 1200])), int(rng.choice([80, 640, 1100]))
        protos, flips = int(rng.choice([3, 40, 5000])), int(rng.integers(0, 40))
        max_dist, ratio = float(rng.choice([5.0, 20.0, 33.5, 50.0, 120.0])), float(rng.choice([0.5, 0.8, 0.95, 1.0, 1.5]))
        clouds = ops.BruteforceClouds(0, B, fs, ms, candidate_capacity=min(fs * ms, 400000))
        inputs, seed = [], int(rng.integers(1 << 30))
        for b in range(B):
            nf, nm = int(rng.integers(1, fs + 1)), int(rng.integers(1, ms + 1))
            df = _tie_heavy(np.random.default_rng(seed + b), protos, nf, flips) if b % 17 < 3 else np.random.default_rng(seed + b).integers(0, 256, (nf, 32), dtype=np.uint8)
            dm = _tie_heavy(np.random.default_rng(seed + b), protos, nm, flips) if b % 17 < 3 else df[np.random.default_rng(b).integers(0, nf, nm)].copy()
            inputs.append((df, dm))
            clouds.upload(b, df, dm)
        ops.bruteforce_match_batch(ctx, ops.bruteforce_params(max_dist, ratio), clouds)
        ctx.synchronize()
        for b in [0, 1, 2, 17, 18, 19, 53, B - 1]:
            ref, rflags = oracle.bruteforce_match(inputs[b][0], inputs[b][1], max_dist, ratio)
            got, gflags = clouds.matches_of(b), int(clouds.status[b].item())
            total += len(ref)
            if not (hp.corr_equal(ref, got) and rflags == gflags):
                bad += 1
                print("BRUTEFORCE BATCH MISMATCH batch %d pair %d fs %d ms %d dist %g ratio %g: %d vs %d (flags %d %d)" % (
                    c, b, fs, ms, max_dist, ratio, len(ref), len(got), rflags, gflags))
    return bad, total


def fuzz_features(cases, rng, ctx):
    import torch
    from oracle import binding_features as of
    from srrg2_proslam_amd import ops
    bad = total = 0
    dev = torch.device("cuda", 0)
    for c in range(cases):
        rows, cols = int(rng.choice([7, 40, 68, 97, 132, 133, 240, 376, 480])), int(rng.choice([7, 64, 68, 131, 132, 133, 197, 640, 1241]))  # 132 = the smallest size with a tile whose halo lies inside the image
        # texture: blocks + blobs + noise of random contrast (corners of every strength), sometimes flat regions
        block = int(rng.choice([4, 8, 16]))
        base = rng.integers(0, 256, (rows // block + 1, cols // block + 1)).astype(np.float32)
        img = np.kron(base, np.ones((block, block), np.float32))[:rows, :cols]
        img += rng.normal(0, float(rng.choice([0.0, 3.0, 12.0])), img.shape)
        if rng.random() < 0.3:
            img[: rows // 2] = 128
        img = np.clip(img, 0, 255).astype(np.uint8)
        threshold = int(rng.choice([1, 5, 15, 25, 60, 120, 254]))
        nms = int(rng.random() < 0.8)
        target = int(rng.choice([50, 300, 1000, 10 ** 6]))
        grid = (int(rng.integers(1, 5)), int(rng.integers(1, 6)))
        stride = 8192
        std = int(rng.random() < 0.5)  # selection in the reference's std::sort order (wave-cooperative introsort replay on the device)
        raw_cap = int(rng.choice([0, 0, 16384, 32768]))
        po = of.extractor_params(threshold, nms, target, grid[0], grid[1], of.SELECT_LIBSTDCXX if std else of.SELECT_CANONICAL)
        pg = ops.extractor_params(threshold, nms, target, grid[0], grid[1], ops.SELECT_LIBSTDCXX if std else ops.SELECT_CANONICAL, raw_cap)
        try:
            uv, oi, od = of.extract_features(po, img, capacity=1 << 20)
        except RuntimeError:
            continue  # more raw detections than the restatement's selection holds
        pad = int(rng.choice([0, 0, 3, 64]))  # row pitch > cols
        wide = torch.zeros((1, rows, cols + pad), dtype=torch.uint8, device=dev)
        wide[:, :, :cols] = torch.from_numpy(img[None]).to(dev)
        t = wide[:, :, :cols]
        kp = torch.zeros((1, stride, 2), dtype=torch.float32, device=dev)
        desc = torch.zeros((1, stride, 32), dtype=torch.uint8, device=dev)
        inten = torch.zeros((1, stride), dtype=torch.float32, device=dev)
        n = torch.zeros((1,), dtype=torch.int32, device=dev)
        st = torch.zeros((1,), dtype=torch.int32, device=dev)
        ops.extract_features_batch(ctx, pg, t, kp, desc, n, st, inten)
        ctx.synchronize()
        status, ng = int(st[0].item()), int(n[0].item())
        if status != 0:
            continue  # more raw detections / keypoints than the device capacities: a loud error by contract
        total += len(uv)
        ok = (ng == len(uv) and np.array_equal(kp[0, :ng].cpu().numpy(), uv) and np.array_equal(desc[0, :ng].cpu().numpy(), od)
              and np.array_equal(inten[0, :ng].cpu().numpy(), oi))
        if not ok:
            bad += 1
            print("FEATURES MISMATCH case %d %dx%d pitch %d threshold %d nms %d target %d grid %s std %d raw_cap %d: %d vs %d" % (c, rows, cols, cols + pad, threshold, nms, target, grid, std, raw_cap, len(uv), ng))
    return bad, total


def run(cases, seed, ctx=None, oracle=None, verbose=True):
    from srrg2_proslam_amd import ops
    if oracle is None:
        from oracle import binding as oracle
        oracle.lib()
    own = ctx is None
    if own:
        ctx = ops.Context(0)
    rng = np.random.default_rng(seed)
    out = {"clip": fuzz_clip(cases, rng, ctx, oracle), "bruteforce": fuzz_bruteforce(cases, rng, ctx, oracle),
           "features": fuzz_features(max(cases // 4, 1), rng, ctx)}
    if own:
        ctx.close()
    if verbose:
        for k, (bad, total) in out.items():
            print("%s: %d mismatches, %d items compared" % (k, bad, total))
    return out


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    s = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    sys.exit(1 if any(b for b, _ in run(n, s).values()) else 0)
