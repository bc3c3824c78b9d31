"""GPU parity of the intensity feature extractor through the C-ABI: keypoints, intensities and
descriptors identical to the oracle INCLUDING order (all integer / byte work), on synthetic stereo images, plus the
chain images -> extractor (left, right) -> epipolar matcher on device buffers."""
import numpy as np
import pytest
import torch

from helpers import corr_equal, oracle_stereo_params
from oracle import binding_features as of
from srrg2_proslam_amd import configs, ops, synthetic as syn

pytestmark = pytest.mark.gpu


def _run(hip_ctx, params, images, stride):
    B = len(images)
    rows, cols = images[0].shape
    dev = torch.device("cuda", 0)
    img = torch.from_numpy(np.stack(images)).to(dev).contiguous()
    kp = torch.zeros((B, stride, 2), dtype=torch.float32, device=dev)
    desc = torch.zeros((B, stride, 32), dtype=torch.uint8, device=dev)
    inten = torch.zeros((B, stride), dtype=torch.float32, device=dev)
    n = torch.zeros((B,), dtype=torch.int32, device=dev)
    st = torch.zeros((B,), dtype=torch.int32, device=dev)
    ops.extract_features_batch(hip_ctx, params, img, kp, desc, n, st, inten)
    hip_ctx.synchronize()
    return kp.cpu().numpy(), desc.cpu().numpy(), inten.cpu().numpy(), n.cpu().numpy(), st.cpu().numpy()


@pytest.mark.parametrize("order", ["canonical", "libstdcxx"])
@pytest.mark.parametrize("threshold,nms,target,grid", [(15, 1, 1000, (3, 3)), (25, 1, 300, (2, 5)), (60, 1, 2000, (1, 1)), (40, 1, 10 ** 6, (4, 4)), (8, 1, 400, (1, 1))])
def test_extractor_parity_on_synthetic_stereo_images(oracle, hip_ctx, threshold, nms, target, grid, order):
    cfg = configs.get("kitti")
    images = []
    for seed in (11, 12):
        l, r, _ = syn.stereo_images(np.random.default_rng(seed), cfg)
        images += [l, r]
    std = order == "libstdcxx"  # the reference's std::sort tie order (one wave per region replays introsort)
    po = of.extractor_params(threshold, nms, target, grid[0], grid[1], of.SELECT_LIBSTDCXX if std else of.SELECT_CANONICAL)
    pg = ops.extractor_params(threshold, nms, target, grid[0], grid[1], ops.SELECT_LIBSTDCXX if std else ops.SELECT_CANONICAL, 32768)
    stride = 8192
    kp, desc, inten, n, st = _run(hip_ctx, pg, images, stride)
    for b, img in enumerate(images):
        uv, oi, od = of.extract_features(po, img, capacity=stride)
        assert st[b] == 0 and n[b] == len(uv) and len(uv) > 100, (b, st[b], n[b], len(uv))
        assert np.array_equal(kp[b, : n[b]], uv), b
        assert np.array_equal(desc[b, : n[b]], od), b
        assert np.array_equal(inten[b, : n[b]], oi), b


def test_raw_detection_capacity_is_a_loud_error(hip_ctx):
    # without non-maximum suppression a low threshold yields more than the 8192 raw detections the selection holds
    cfg = configs.get("kitti")
    l, _, _ = syn.stereo_images(np.random.default_rng(11), cfg)
    kp, desc, inten, n, st = _run(hip_ctx, ops.extractor_params(5, 0, 2000, 1, 1), [l], 4096)
    assert st[0] == -2 and n[0] == 0


@pytest.mark.parametrize("nms", [1, 0])
def test_noise_overflows_the_compass_lists(oracle, hip_ctx, nms):
    """White noise at a low threshold: more than half of a wave's pixels pass the compass test, the tile kernel's survivor
    lists overflow (lanes score their pixel themselves) and suppression falls back from the walk over the lists to the
    sweep over all pixels; with nms = 0 every corner is a detection.  Same features as the checker, in the same order."""
    rng = np.random.default_rng(77)
    images = [rng.integers(0, 256, (200, 264), dtype=np.uint8), rng.integers(90, 166, (131, 197), dtype=np.uint8)]
    for order_o, order_g in ((of.SELECT_LIBSTDCXX, ops.SELECT_LIBSTDCXX), (of.SELECT_CANONICAL, ops.SELECT_CANONICAL)):
        po = of.extractor_params(4, nms, 600, 2, 2, order_o)
        pg = ops.extractor_params(4, nms, 600, 2, 2, order_g, 32768)
        for img in images:
            kp, desc, inten, n, st = _run(hip_ctx, pg, [img], 2048)
            uv, oi, od = of.extract_features(po, img, capacity=2048)
            assert st[0] == 0 and n[0] == len(uv) and len(uv) > 100, (st[0], n[0], len(uv))
            assert np.array_equal(kp[0, : n[0]], uv) and np.array_equal(desc[0, : n[0]], od) and np.array_equal(inten[0, : n[0]], oi)


def test_batch_of_21_images_keeps_every_image_to_itself(oracle, hip_ctx):
    """A batch that is not a multiple of eight (the describe pass folds image and chunk into the workgroup index, eight images at
    a time; the tile pass appends to per-image lists with atomics): seven different images in shuffled triplicate, every copy
    must equal the checker's extraction of its own image, in the reference's order."""
    cfg = configs.get("kitti")
    rng = np.random.default_rng(3)
    distinct = []
    for seed in (21, 22, 23, 24):
        l, r, _ = syn.stereo_images(np.random.default_rng(seed), cfg)
        distinct += [l, r]
    distinct = distinct[:7]
    order = rng.permutation(np.repeat(np.arange(7), 3))
    po = of.extractor_params(15, 1, 1000, 3, 3, of.SELECT_LIBSTDCXX)
    pg = ops.extractor_params(15, 1, 1000, 3, 3, ops.SELECT_LIBSTDCXX)
    want = [of.extract_features(po, img, capacity=2048) for img in distinct]
    kp, desc, inten, n, st = _run(hip_ctx, pg, [distinct[i] for i in order], 2048)
    for b, i in enumerate(order):
        uv, oi, od = want[i]
        assert st[b] == 0 and n[b] == len(uv), (b, i, st[b], n[b], len(uv))
        assert np.array_equal(kp[b, : n[b]], uv) and np.array_equal(desc[b, : n[b]], od) and np.array_equal(inten[b, : n[b]], oi), (b, i)


def test_small_odd_sized_and_flat_images(oracle, hip_ctx):
    rng = np.random.default_rng(5)
    po, pg = of.extractor_params(12, 1, 200, 2, 2), ops.extractor_params(12, 1, 200, 2, 2)
    for rows, cols in ((37, 41), (64, 64), (101, 333)):
        img = (rng.integers(0, 2, ((rows + 5) // 6, (cols + 5) // 6)) * 200 + 20).astype(np.uint8)
        img = np.kron(img, np.ones((6, 6), np.uint8))[:rows, :cols].copy()
        flat = np.full((rows, cols), 77, np.uint8)
        kp, desc, inten, n, st = _run(hip_ctx, pg, [img, flat], 512)
        uv, oi, od = of.extract_features(po, img, capacity=512)
        assert n[0] == len(uv) and np.array_equal(kp[0, : n[0]], uv) and np.array_equal(desc[0, : n[0]], od)
        assert n[1] == 0 and st[1] == 2  # no keypoints: warning, empty cloud


def test_images_to_matches_on_device(oracle, hip_ctx):
    """left and right images -> extractor writes straight into the matcher's input arrays -> epipolar matcher"""
    cfg = configs.get("kitti")
    B, stride = 3, 1024
    lefts, rights = [], []
    for b in range(B):
        l, r, _ = syn.stereo_images(np.random.default_rng(40 + b), cfg)
        lefts.append(l)
        rights.append(r)
    sf = ops.StereoFrames(0, B, stride, epilogue=False)
    dev = sf.left_kp.device
    pg = ops.extractor_params()
    st = torch.zeros((B,), dtype=torch.int32, device=dev)
    ops.extract_features_batch(hip_ctx, pg, torch.from_numpy(np.stack(lefts)).to(dev), sf.left_kp, sf.left_desc, sf.n_left, st)
    ops.extract_features_batch(hip_ctx, pg, torch.from_numpy(np.stack(rights)).to(dev), sf.right_kp, sf.right_desc, sf.n_right, st)
    sp = ops.stereo_params(cfg["stereo_matcher"], cfg["camera"]["rows"])
    ops.stereo_match_batch(hip_ctx, sp, sf)
    hip_ctx.synchronize()
    po = of.extractor_params()
    for b in range(B):
        uvl, _, dl = of.extract_features(po, lefts[b])
        uvr, _, dr = of.extract_features(po, rights[b])
        ref, _ = oracle.stereo_match(uvl, dl, uvr, dr, oracle_stereo_params(oracle, cfg["stereo_matcher"]))
        assert len(ref) > 200
        assert corr_equal(ref, sf.matches_of(b)), b


def test_host_pointer_entry_point_on_a_reference_image(oracle, hip_ctx):
    """prs_extract_features (one image, host buffers: what an adapter's compute(image) binds) = the oracle on KITTI frame 00,
    in the reference's selection order; a capacity below the result is a loud error"""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_kitti.npz"))
    img = z["city_left"][0]
    pg = ops.extractor_params(5, 1, 500, 3, 3, ops.SELECT_LIBSTDCXX, 32768)
    uv, inten, desc = ops.extract_features(hip_ctx, pg, img)
    ouv, ointen, odesc = of.extract_features(of.extractor_params(5, 1, 500, 3, 3, of.SELECT_LIBSTDCXX), img)
    assert len(uv) == 446  # tests/test_feature_extractors.cpp / fixtures.hpp: 446 features on the left image
    assert np.array_equal(uv, ouv) and np.array_equal(inten, ointen) and np.array_equal(desc, odesc)
    with pytest.raises(ops.ProslamHipError) as ei:
        ops.extract_features(hip_ctx, pg, img, capacity=100)
    assert ei.value.status == ops._lib.ERR_CAPACITY


def test_selection_order_equals_libstdcxx_sort(oracle, hip_ctx):
    """prs_selection_order (the device replay of std::sort the extractor's PRS_SELECT_LIBSTDCXX selection runs) against the
    CPU restatement, which tests/test_oracle_features.py::test_F7 pins to g++'s own std::sort: random responses with many
    ties, every small size (the in-register path and its <= 16 leaves), sizes around the 64-item hand-over, big regions
    (work queue), sorted / reversed / constant runs and structured inputs that exhaust the depth limit (heapsort on one
    lane, from the queue path and from inside the in-register path).  Bit-exact: the permutation itself is compared."""
    rng = np.random.default_rng(11)
    cases = []
    for n in list(range(0, 70)) + [100, 127, 128, 129, 257, 1000, 1650, 6000, 20000, 32768]:
        for vals in (2, 5, 30, 250):
            cases.append(rng.integers(1, vals + 1, n))
        cases += [np.arange(n) % 255 + 1, (np.arange(n)[::-1] % 255 + 1).copy(), np.full(n, 7)]
    heap_cases = 0
    for n in (40, 64, 200, 1000, 5000):
        cases.append(np.concatenate([np.arange(0, n, 2), np.arange(1, n, 2)]) % 250 + 1)
        for m in (3, 17, 101):
            cases.append((np.arange(n) * 7919) % m + 1)
        cases.append((np.abs(np.arange(n) - n // 2) + rng.integers(0, 2, n)) % 255 + 1)
        cases.append(255 - (np.abs(np.arange(n) - n // 2) + rng.integers(0, 2, n)) % 255)
    # median-of-three killers on small value ranges: long organ-pipe / sawtooth runs
    for n in (48, 64, 90, 300, 3000):
        k = np.arange(n)
        cases.append(np.where(k % 2 == 0, k // 2 % 200 + 1, 201 + (k // 2) % 50))
        cases.append(np.minimum(k, n - 1 - k) % 254 + 1)
    for x in cases:
        x = np.asarray(x, dtype=np.int64)
        assert x.size == 0 or (x.min() >= 1 and x.max() <= 255)
        want = of.std_sort_desc(x)
        got = ops.selection_order(hip_ctx, x.astype(np.uint8))
        assert np.array_equal(got, want), "n=%d" % len(x)
    with pytest.raises(ops.ProslamHipError):
        ops.selection_order(hip_ctx, np.array([3, 0, 5], dtype=np.uint8))  # a response of 0 does not exist
