"""GPU parity of the bijective brute-force matcher through the C-ABI:
correspondence vectors identical to the oracle INCLUDING order ((response, fixed) ascending)."""
import numpy as np
import pytest

from srrg2_proslam_amd import ops
from tests import helpers as hp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["popcount", "matrix", "default"])
def hip_ctx(request):
    """every test of this module on the three dense-phase modes (prs_context_set_bruteforce_dense_phase): the popcount kernels, the
    matrix cores whatever the batch size (v_mfma_i32_16x16x64_i8: the fused shape where one workgroup takes a pair, else the split
    kernel that re-scores the entries it selects), and the default (matrix cores for batches that fill the chip)"""
    ctx = ops.Context(0)
    if request.param != "default":
        ctx.set_bruteforce_dense_phase(ops.BF_DENSE_MATRIX if request.param == "matrix" else ops.BF_DENSE_POPCOUNT)
    yield ctx
    ctx.close()


def _tie_heavy(rng, n_base, n, flips):
    """descriptors drawn from few prototypes with a handful of flipped bits: many equal distances,
    shared best partners, pools with conflicts (bruteforce_impl.cpp:247-266)"""
    base = rng.integers(0, 256, (n_base, 32), dtype=np.uint8)
    out = base[rng.integers(0, n_base, n)].copy()
    for i in range(n):
        for _ in range(int(rng.integers(0, flips + 1))):
            b = int(rng.integers(0, 256))
            out[i, b >> 3] ^= np.uint8(1 << (b & 7))
    return out


@pytest.mark.parametrize("max_dist,ratio", [(50.0, 0.9), (25.0, 0.8), (75.0, 0.5), (33.5, 0.95), (256.0, 1.5)])
def test_stereo_pair_descriptors(oracle, hip_ctx, max_dist, ratio):
    n = 2000 if max_dist < 100 else 300  # threshold 256 keeps every pair: n^2 candidates
    cfg, fr = hp.kitti_frame(41, n=n)
    ref, rflags = oracle.bruteforce_match(fr["desc_left"], fr["desc_right"], max_dist, ratio)
    if max_dist >= 100:
        from srrg2_proslam_amd import _lib
        clouds = ops.BruteforceClouds(0, 1, n, n, candidate_capacity=n * n)
        clouds.upload(0, fr["desc_left"], fr["desc_right"])
        ops.bruteforce_match_batch(hip_ctx, ops.bruteforce_params(max_dist, ratio), clouds)
        hip_ctx.synchronize()
        got, gflags = clouds.matches_of(0), int(clouds.status[0].item())
    else:
        got, gflags = ops.bruteforce_match(hip_ctx, ops.bruteforce_params(max_dist, ratio), fr["desc_left"], fr["desc_right"])
    assert len(ref) > (100 if max_dist < 100 else 0)
    assert hp.corr_equal(ref, got) and rflags == gflags


def test_cloud_versus_itself(oracle, hip_ctx):
    # identical clouds: every point matches itself with response 0 (tests/test_correspondence_finders.cpp:160-181)
    rng = np.random.default_rng(2)
    d = rng.integers(0, 256, (777, 32), dtype=np.uint8)
    got, _ = ops.bruteforce_match(hip_ctx, ops.bruteforce_params(), d, d)
    ref, _ = oracle.bruteforce_match(d, d, 50.0, 0.9)
    assert hp.corr_equal(ref, got)
    assert len(got) == 777 and np.array_equal(got["fixed_idx"], got["moving_idx"]) and np.all(got["response"] == 0)


@pytest.mark.parametrize("seed", range(6))
def test_pools_with_ties_and_conflicts(oracle, hip_ctx, seed):
    rng = np.random.default_rng(100 + seed)
    nf, nm = int(rng.integers(50, 1500)), int(rng.integers(50, 1500))
    df = _tie_heavy(rng, 40, nf, 6)
    dm = _tie_heavy(np.random.default_rng(100 + seed), 40, nm, 6)  # same prototypes
    for max_dist, ratio in ((20.0, 0.9), (12.0, 0.7), (30.0, 1.0)):
        ref, rflags = oracle.bruteforce_match(df, dm, max_dist, ratio)
        clouds = ops.BruteforceClouds(0, 1, nf, nm, candidate_capacity=nf * nm)
        clouds.upload(0, df, dm)
        ops.bruteforce_match_batch(hip_ctx, ops.bruteforce_params(max_dist, ratio), clouds)
        hip_ctx.synchronize()
        assert hp.corr_equal(ref, clouds.matches_of(0)), (seed, max_dist, ratio)
        assert int(clouds.status[0].item()) == rflags


def test_batched_ragged_and_large_fixed_clouds(oracle, hip_ctx):
    rng = np.random.default_rng(9)
    sizes = [(0, 10), (10, 0), (1, 1), (1500, 2600), (2600, 1500), (1025, 64), (333, 4000)]
    clouds = ops.BruteforceClouds(0, len(sizes), 2600, 4000)
    inputs = []
    for b, (nf, nm) in enumerate(sizes):
        proto = rng.integers(0, 256, (max(nf, nm, 1), 32), dtype=np.uint8)
        df = proto[:nf].copy()
        dm = proto[rng.permutation(max(nf, nm, 1))[:nm]].copy()
        flips = rng.integers(0, 256, (nm, 10))
        for i in range(nm):
            for bit in flips[i][: int(rng.integers(0, 11))]:
                dm[i, bit >> 3] ^= np.uint8(1 << (bit & 7))
        inputs.append((df, dm))
        clouds.upload(b, df, dm)
    ops.bruteforce_match_batch(hip_ctx, ops.bruteforce_params(50.0, 0.9), clouds)
    hip_ctx.synchronize()
    for b, (df, dm) in enumerate(inputs):
        ref, rflags = oracle.bruteforce_match(df, dm, 50.0, 0.9)
        assert hp.corr_equal(ref, clouds.matches_of(b)), b
        assert int(clouds.status[b].item()) == rflags, b


def test_many_pairs_share_workgroups(oracle, hip_ctx):
    # more cloud pairs than workgroups in flight: per-workgroup scratch is reused pair after pair
    rng = np.random.default_rng(17)
    B, n = 600, 96
    clouds = ops.BruteforceClouds(0, B, n, n)
    inputs = []
    for b in range(B):
        nf, nm = int(rng.integers(1, n + 1)), int(rng.integers(1, n + 1))
        df = _tie_heavy(rng, 8, nf, 4)
        dm = np.concatenate([df[: min(nf, nm)], _tie_heavy(rng, 8, nm, 4)])[:nm]
        inputs.append((df, dm))
        clouds.upload(b, df, dm)
    ops.bruteforce_match_batch(hip_ctx, ops.bruteforce_params(16.0, 0.9), clouds)
    hip_ctx.synchronize()
    for b in range(0, B, 7):
        ref, rflags = oracle.bruteforce_match(inputs[b][0], inputs[b][1], 16.0, 0.9)
        assert hp.corr_equal(ref, clouds.matches_of(b)), b
        assert int(clouds.status[b].item()) == rflags, b


def test_error_and_warning_contract(hip_ctx):
    import ctypes as C
    from srrg2_proslam_amd import _lib
    L = _lib.load()
    p = ops.bruteforce_params()
    d = np.zeros((4, 32), np.uint8)
    out = np.zeros(4, dtype=ops.CORR_DTYPE)
    n = C.c_int32(0)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    # unset buffers throw in the reference (bruteforce_impl.cpp:203-216)
    assert L.prs_bruteforce_match(hip_ctx._h, C.byref(p), None, 4, vp(d), 4, vp(out), 4, C.byref(n)) == -1
    assert b"fixed not set" in L.prs_last_error(hip_ctx._h)
    assert L.prs_bruteforce_match(hip_ctx._h, C.byref(p), vp(d), 4, None, 4, vp(out), 4, C.byref(n)) == -1
    assert b"moving not set" in L.prs_last_error(hip_ctx._h)
    assert L.prs_bruteforce_match(hip_ctx._h, C.byref(p), vp(d), 4, vp(d), 4, None, 4, C.byref(n)) == -1
    assert b"correspondences not set" in L.prs_last_error(hip_ctx._h)
    # empty clouds: warnings, empty result (:217-226, :237-242)
    assert L.prs_bruteforce_match(hip_ctx._h, C.byref(p), vp(d), 0, vp(d), 4, vp(out), 4, C.byref(n)) == 3 and n.value == 0
    # all-equal descriptors: n^2 candidates at distance 0 exceed the default candidate capacity; the host-pointer entry
    # grows the list and answers like the reference (every candidate is ambiguous: no match, warning)
    big = np.zeros((300, 32), np.uint8)
    o2 = np.zeros(300, dtype=ops.CORR_DTYPE)
    from oracle import binding as ob
    ref, ref_flags = ob.bruteforce_match(big, big, p.maximum_descriptor_distance, p.maximum_distance_ratio_to_second_best)
    assert L.prs_bruteforce_match(hip_ctx._h, C.byref(p), vp(big), 300, vp(big), 300, vp(o2), 300, C.byref(n)) == ref_flags
    assert n.value == len(ref)


def test_dense_phase_setter(hip_ctx):
    # prs_context_set_bruteforce_dense_phase: the three PRS_BF_DENSE_* modes are accepted, anything else is refused loudly
    probe = ops.Context(0)
    try:
        for mode in (ops.BF_DENSE_POPCOUNT, ops.BF_DENSE_MATRIX_WHEN_FULL, ops.BF_DENSE_MATRIX):
            probe.set_bruteforce_dense_phase(mode)
        with pytest.raises(RuntimeError, match="PRS_BF_DENSE"):
            probe.set_bruteforce_dense_phase(7)
    finally:
        probe.close()


def test_registration_state_in_global_memory(oracle, hip_ctx, monkeypatch):
    # the distance bitmaps and the level lists live in LDS when they fit (the usual case); PRS_BF_GLOBAL_STATE keeps them in the
    # scratch rows in global memory, the path large clouds x large thresholds take
    monkeypatch.setenv("PRS_BF_GLOBAL_STATE", "1")
    rng = np.random.default_rng(77)
    df = _tie_heavy(rng, 40, 900, 6)
    dm = _tie_heavy(rng, 40, 800, 6)
    dm[:300] = df[rng.permutation(900)[:300]]
    for max_dist, ratio in ((40.0, 0.9), (12.0, 1.0)):
        ref, rflags = oracle.bruteforce_match(df, dm, max_dist, ratio)
        clouds = ops.BruteforceClouds(0, 1, 900, 800, candidate_capacity=900 * 800)
        clouds.upload(0, df, dm)
        ops.bruteforce_match_batch(hip_ctx, ops.bruteforce_params(max_dist, ratio), clouds)
        hip_ctx.synchronize()
        assert hp.corr_equal(ref, clouds.matches_of(0)) and rflags == int(clouds.status[0].item()), (max_dist, ratio)


def test_full_batch_with_dense_candidates_and_two_row_passes(oracle, hip_ctx):
    # more cloud pairs than half the CUs (the default takes the fused matrix-core shape), fixed clouds beyond 1024 rows (a second pass
    # over the moving cloud), descriptors drawn from few prototypes: every tile of every tile row holds candidates, wave segments
    # are drained many times per pair
    rng = np.random.default_rng(23)
    B, fs, ms = 140, 1300, 1100
    clouds = ops.BruteforceClouds(0, B, fs, ms, candidate_capacity=200000)
    inputs = []
    for b in range(B):
        nf = int(rng.integers(900, fs + 1)) if b % 3 else int(rng.integers(1, 200))
        nm = int(rng.integers(700, ms + 1)) if b % 5 else int(rng.integers(1, 100))
        df = _tie_heavy(rng, 60, nf, 40)
        dm = _tie_heavy(np.random.default_rng(23), 60, nm, 40)  # (the same prototypes)
        inputs.append((df, dm))
        clouds.upload(b, df, dm)
    for max_dist, ratio in ((50.0, 0.9), (31.0, 0.8)):
        ops.bruteforce_match_batch(hip_ctx, ops.bruteforce_params(max_dist, ratio), clouds)
        hip_ctx.synchronize()
        for b in list(range(0, B, 13)) + [B - 1]:
            ref, rflags = oracle.bruteforce_match(inputs[b][0], inputs[b][1], max_dist, ratio)
            assert hp.corr_equal(ref, clouds.matches_of(b)), (b, max_dist)
            assert int(clouds.status[b].item()) == rflags, (b, max_dist)


@pytest.mark.parametrize("two_workgroups", ["0", "1"])
def test_real_descriptors_of_kitti_stereo_pairs(oracle, hip_ctx, monkeypatch, two_workgroups):
    # (the fused matrix-core shape as one 1024-thread workgroup per CU and as two 512-thread ones -- what batches of two cloud pairs
    #  per CU take when the clouds leave room for two in the LDS, as these do; forced either way here)
    monkeypatch.setenv("PRS_BF_TWO_WORKGROUPS", two_workgroups)
    # the descriptors our extractor finds in the reference's KITTI test images (left cloud = fixed, right cloud = moving): 1.6 % of the
    # pairs are within 50 bits, 7 % within 75 -- nothing like uniform random rows.  140 cloud pairs (the seven stereo pairs replicated):
    # the default takes the fused matrix-core shape; every distinct pair against the checker.
    import os
    import torch
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_kitti.npz"))
    left = [im for im in z["city_left"]] + [im for im in z["highway_left"]]
    right = [im for im in z["city_right"]] + [im for im in z["highway_right"]]
    dev = torch.device("cuda", 0)
    img = torch.from_numpy(np.stack(left + right)).to(dev)
    n_img, stride = img.shape[0], 1024
    kp = torch.zeros((n_img, stride, 2), dtype=torch.float32, device=dev)
    desc = torch.zeros((n_img, stride, 32), dtype=torch.uint8, device=dev)
    n = torch.zeros((n_img,), dtype=torch.int32, device=dev)
    st = torch.zeros((n_img,), dtype=torch.int32, device=dev)
    ectx = ops.Context(0)
    ectx.use_torch_stream()
    ops.extract_features_batch(ectx, ops.extractor_params(), img, kp, desc, n, st)
    torch.cuda.synchronize()
    ectx.close()
    pairs, B = len(left), 140
    sel = torch.arange(B, device=dev) % pairs
    clouds = ops.BruteforceClouds(0, B, stride, stride, candidate_capacity=200000)
    clouds.fixed_desc.copy_(desc[sel])
    clouds.moving_desc.copy_(desc[sel + pairs])
    clouds.n_fixed.copy_(n[sel])
    clouds.n_moving.copy_(n[sel + pairs])
    torch.cuda.synchronize()
    hd, hn = desc.cpu().numpy(), n.cpu().numpy()
    for max_dist, ratio in ((50.0, 0.9), (75.0, 0.8)):
        ops.bruteforce_match_batch(hip_ctx, ops.bruteforce_params(max_dist, ratio), clouds)
        hip_ctx.synchronize()
        for b in list(range(pairs)) + [B - 1]:
            k = b % pairs
            ref, rflags = oracle.bruteforce_match(hd[k, : hn[k]], hd[k + pairs, : hn[k + pairs]], max_dist, ratio)
            assert len(ref) > 50
            assert hp.corr_equal(ref, clouds.matches_of(b)), (b, max_dist)
            assert int(clouds.status[b].item()) == rflags, (b, max_dist)


def test_capacity_overflow_is_reported_per_cloud_pair(oracle, hip_ctx):
    # a cloud pair with more candidates than candidate_capacity reports PRS_ERR_CAPACITY and no matches; its neighbours in the batch
    # are unaffected (full batch: the default runs the fused matrix-core shape, whose drains stop storing at the capacity)
    rng = np.random.default_rng(5)
    B, n = 140, 200
    clouds = ops.BruteforceClouds(0, B, n, n, candidate_capacity=1000)
    inputs = []
    for b in range(B):
        if b % 10 == 3:
            df = np.tile(rng.integers(0, 256, (1, 32), dtype=np.uint8), (n, 1))  # all rows equal: n^2 candidates at distance 0
            dm = df.copy()
        else:
            df = rng.integers(0, 256, (n, 32), dtype=np.uint8)
            dm = df[rng.permutation(n)].copy()
        inputs.append((df, dm))
        clouds.upload(b, df, dm)
    ops.bruteforce_match_batch(hip_ctx, ops.bruteforce_params(50.0, 0.9), clouds)
    hip_ctx.synchronize()
    status = clouds.status.cpu().numpy()
    for b in range(B):
        if b % 10 == 3:
            assert status[b] == -2 and int(clouds.n_matches[b].item()) == 0, (b, status[b])  # PRS_ERR_CAPACITY
        elif b % 7 == 0:
            ref, rflags = oracle.bruteforce_match(inputs[b][0], inputs[b][1], 50.0, 0.9)
            assert hp.corr_equal(ref, clouds.matches_of(b)) and status[b] == rflags, b


@pytest.mark.parametrize("two_workgroups", ["0", "1"])
def test_forty_cloud_pairs(oracle, hip_ctx, monkeypatch, two_workgroups):
    monkeypatch.setenv("PRS_BF_TWO_WORKGROUPS", two_workgroups)
    # 32 or more cloud pairs of at least 256 x 64 points: the default switches to the fused matrix-core shape (fewer workgroups than CUs)
    rng = np.random.default_rng(41)
    B, fs, ms = 40, 420, 380
    clouds = ops.BruteforceClouds(0, B, fs, ms, candidate_capacity=fs * ms)
    inputs = []
    for b in range(B):
        nf, nm = int(rng.integers(1, fs + 1)), int(rng.integers(1, ms + 1))
        df = _tie_heavy(rng, 30, nf, 30) if b % 2 else rng.integers(0, 256, (nf, 32), dtype=np.uint8)
        dm = _tie_heavy(np.random.default_rng(41), 30, nm, 30) if b % 2 else df[rng.integers(0, nf, nm)].copy()
        inputs.append((df, dm))
        clouds.upload(b, df, dm)
    ops.bruteforce_match_batch(hip_ctx, ops.bruteforce_params(45.0, 0.85), clouds)
    hip_ctx.synchronize()
    for b in range(0, B, 3):
        ref, rflags = oracle.bruteforce_match(inputs[b][0], inputs[b][1], 45.0, 0.85)
        assert hp.corr_equal(ref, clouds.matches_of(b)) and int(clouds.status[b].item()) == rflags, b
