"""GPU parity: HIP ORB extraction (through the C-ABI) vs the CPU oracle on the same seeded inputs.
Bar: bit-exact pyramid / blur / FAST candidates / keypoints / descriptors (BASELINE.json north_star)."""
import numpy as np
import pytest
import torch  # noqa: F401  (before the library loads, so that both resolve the same HIP runtime)

from eao_fusion_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import eao_fusion_amd as E
    assert E.load().eao_device_check() == 0, E.load().eao_last_error()
    return E


def _compare_frame(E, oracle, ext, orc, img, frame=0, stages=True):
    okps, odesc = orc.extract(img)
    if stages:
        for l in range(ext.nlevels):
            assert np.array_equal(ext.level_image(l, frame), orc.level_image(l)), "pyramid level %d" % l
            cg = ext.level_candidates(l, frame)
            co = orc.level_candidates(l)
            assert cg.shape == co.shape and np.array_equal(cg, co), "FAST candidates level %d" % l
            ob = orc.level_image(l, blurred=True)
            if ob is not None:
                assert np.array_equal(ext.level_image(l, frame, blurred=True), ob), "blur level %d" % l
    return okps, odesc


@pytest.mark.parametrize("seed,n_rect,n_small", [(1000, 400, 1000), (1001, 400, 1000), (7, 40, 0), (8, 5, 0)])
def test_single_frame_bit_exact(gpu, oracle, seed, n_rect, n_small):
    img = synth.synth_frame(seed, n_rect=n_rect, n_small=n_small)
    ext = gpu.ORBextractor(1000, 1.2, 8, 20, 7)
    orc = oracle.OrbOracle(1000, 1.2, 8, 20, 7)
    kps, desc = ext(img)
    okps, odesc = _compare_frame(gpu, oracle, ext, orc, img)
    assert len(kps) == len(okps)
    for field in ("x", "y", "size", "response", "octave", "class_id", "angle"):
        assert np.array_equal(kps[field], okps[field]), field
    assert np.array_equal(desc, odesc)


def test_tables_and_getters(gpu, oracle):
    ext = gpu.ORBextractor(1000, 1.2, 8, 20, 7)
    t = oracle.OrbOracle(1000, 1.2, 8, 20, 7).tables()
    assert ext.GetLevels() == 8 and ext.GetScaleFactor() == pytest.approx(1.2)
    assert np.array_equal(ext.GetScaleFactors(), t["scale"])
    assert np.array_equal(ext.GetInverseScaleFactors(), t["inv_scale"])
    assert np.array_equal(ext.GetScaleSigmaSquares(), t["sigma2"])
    assert np.array_equal(ext.GetInverseScaleSigmaSquares(), t["inv_sigma2"])
    assert list(ext.mnFeaturesPerLevel) == [217, 181, 151, 126, 105, 87, 73, 60]


def test_batch_matches_single_and_oracle(gpu, oracle):
    imgs = synth.synth_frames(6, seed0=1100)
    imgs[3] = 77          # a frame without a single corner inside the batch: zero keypoints for it, the others unaffected
    ext = gpu.ORBextractor(1000, 1.2, 8, 20, 7)
    orc = oracle.OrbOracle(1000, 1.2, 8, 20, 7)
    kps, desc = ext.extract_batch(imgs)
    for f in range(len(imgs)):
        okps, odesc = orc.extract(imgs[f])
        assert np.array_equal(kps[f], okps), "frame %d keypoints" % f
        assert np.array_equal(desc[f], odesc), "frame %d descriptors" % f


def test_host_batch_with_rows_narrower_than_the_device_pitch(gpu, oracle):
    """Round 6: small host-API uploads are staged through a pinned buffer of the handle, row by row when the caller's stride is not the device pitch (752 px rows, 768 B
    pitch) -- a batch of such frames against the oracle and against its own single-frame calls; then a batch too large for the staging buffer (the runtime's 2-D copy)."""
    imgs = np.stack([synth.synth_frame(2100 + f, w=752, h=480, n_rect=250, n_small=1200) for f in range(3)])
    ext = gpu.ORBextractor(800, 1.2, 8, 20, 7)
    orc = oracle.OrbOracle(800, 1.2, 8, 20, 7)
    kps, desc = ext.extract_batch(imgs)
    for f in range(len(imgs)):
        okps, odesc = orc.extract(imgs[f])
        assert np.array_equal(kps[f], okps) and np.array_equal(desc[f], odesc), "frame %d" % f
        k1, d1 = ext(imgs[f])
        assert np.array_equal(k1, kps[f]) and np.array_equal(d1, desc[f]), "frame %d alone" % f
    big = np.stack([imgs[f % 3] for f in range(24)])      # 24 x 768 x 480 B = 8.8 MB: beyond the staging buffer
    kb, db = ext.extract_batch(big)
    for f in range(len(big)):
        assert np.array_equal(kb[f], kps[f % 3]) and np.array_equal(db[f], desc[f % 3]), "frame %d of the large batch" % f


@pytest.mark.parametrize("cfg,shape", [((500, 1.2, 8, 20, 7), (480, 640)), ((2000, 1.2, 8, 20, 7), (376, 1241)),
                                       ((1200, 1.1, 4, 15, 5), (480, 752)), ((300, 1.5, 3, 30, 10), (240, 320))])
def test_other_configurations(gpu, oracle, cfg, shape):
    """KITTI-like wide frames (3 initial quad-tree nodes), EuRoC size, other scale factors / level counts."""
    img = synth.synth_frame(42, w=shape[1], h=shape[0], n_rect=300, n_small=1500)
    ext = gpu.ORBextractor(*cfg)
    orc = oracle.OrbOracle(*cfg)
    kps, desc = ext(img)
    okps, odesc = _compare_frame(gpu, oracle, ext, orc, img)
    assert np.array_equal(kps, okps)
    assert np.array_equal(desc, odesc)


def test_empty_and_flat_images(gpu, oracle):
    ext = gpu.ORBextractor(1000, 1.2, 8, 20, 7)
    assert ext(np.zeros((0, 0), np.uint8)) == (None, None)      # empty image: outputs untouched
    kps, desc = ext(np.full((480, 640), 77, np.uint8))           # no corner anywhere: zero keypoints
    assert len(kps) == 0 and desc.shape == (0, 32)
    with pytest.raises(gpu.EaoError):
        ext(np.zeros((40, 40), np.uint8))                        # smaller than one FAST cell: loud error


@pytest.mark.parametrize("B", [64, 104])
def test_full_batch64_properties(gpu, oracle, B):
    """BASELINE configs[1] size (64 frames: side-stream FAST of levels 0-2 with their quad-trees behind, frame -> XCD affinity)
    and a batch beyond 96 frames (all quad-trees after FAST): determinism across calls + parity of EVERY frame with the oracle."""
    imgs = synth.synth_frames(B)
    ext = gpu.ORBextractor(1000, 1.2, 8, 20, 7)
    k1, d1 = ext.extract_batch(imgs)
    k2, d2 = ext.extract_batch(imgs)
    for f in range(B):
        assert np.array_equal(k1[f], k2[f]) and np.array_equal(d1[f], d2[f])
        assert 1000 <= len(k1[f]) <= 1016
        assert np.all(np.diff(k1[f]["octave"]) >= 0)
    # EVERY frame of the batch against the oracle (21 ms per frame on the CPU), not a spot check
    import threading
    from concurrent.futures import ThreadPoolExecutor
    tl = threading.local()

    def one(f):
        if not hasattr(tl, "orc"):
            tl.orc = oracle.OrbOracle(1000, 1.2, 8, 20, 7)
        return tl.orc.extract(imgs[f])
    with ThreadPoolExecutor(8) as ex:
        refs = list(ex.map(one, range(B)))
    for f in range(B):
        okps, odesc = refs[f]
        assert np.array_equal(k1[f], okps) and np.array_equal(d1[f], odesc), "frame %d of %d" % (f, B)


def test_mvImagePyramid_has_reference_border(gpu, oracle):
    """eao_orb_pyramid (one launch into the handle's pinned block) against the oracle's level images padded with numpy's BORDER_REFLECT_101, every level,
    every pixel; then eao_orb_extract_ref (what the class-surface adapter calls) with the pyramid exported behind the extraction itself."""
    ext = gpu.ORBextractor(1000, 1.2, 8, 20, 7)
    img = synth.synth_frame(3)
    k0, d0 = ext(img)
    pyr = ext.mvImagePyramid
    assert pyr[0].shape == (480 + 38, 640 + 38)
    assert np.array_equal(pyr[0][19:-19, 19:-19], img)
    assert np.array_equal(pyr[0][0, 19:-19], img[19])      # BORDER_REFLECT_101
    ref = oracle.OrbOracle(1000, 1.2, 8, 20, 7)
    ref.extract(img)
    for l in range(8):
        assert np.array_equal(pyr[l], np.pad(ref.level_image(l), 19, mode="reflect")), "level %d" % l
    for l, lv in enumerate(ext.image_pyramid(3)):             # another border width, asked for on demand
        assert np.array_equal(lv, np.pad(ref.level_image(l), 3, mode="reflect")), "level %d, border 3" % l
    ext.set_keep_pyramid(19)
    img2 = synth.synth_frame(4)
    k2, d2 = ext.extract_ref(img2)
    ok2, od2 = ref.extract(img2)
    assert np.array_equal(k2, ok2) and np.array_equal(d2, od2)
    for l, lv in enumerate(ext.mvImagePyramid):              # exported in the extraction's own stream pass: the call only hands out views
        assert np.array_equal(lv, np.pad(ref.level_image(l), 19, mode="reflect")), "level %d (kept)" % l
    ext.set_keep_pyramid(-1)
    k1, d1 = ext.extract_ref(img)
    assert np.array_equal(k1, k0) and np.array_equal(d1, d0)
    e, _ = ext.extract_ref(np.zeros((0, 0), np.uint8))
    assert len(e) == 0


def test_random_configurations(gpu, oracle):
    """A seeded sample of tools/sweep_orb.py: random landscape sizes, feature counts, scale factors, level counts, FAST
    thresholds and textures (including pure noise, where every pixel is a corner candidate) -- keypoints and descriptors
    bit-exact against the oracle.  (The full sweep ran 140 configurations without a mismatch.)"""
    rng = np.random.default_rng(20260101)
    done = 0
    while done < 12:
        w = int(rng.integers(320, 1400)); h = int(rng.integers(200, min(w, 900) + 1))
        nfeat = int(rng.choice([100, 300, 500, 1000, 1500, 2000]))
        sf = float(rng.choice([1.1, 1.2, 1.25, 1.3, 1.5, 2.0]))
        top = int(np.floor(np.log(min(w, h) / 90.0) / np.log(sf))) + 1
        nlev = int(rng.integers(2, max(3, min(8, top) + 1)))
        ini = int(rng.choice([10, 20, 30, 50])); mn = min(int(rng.choice([3, 5, 7, 9])), ini)
        n_rect = int(rng.choice([0, 5, 40, 400])); n_small = int(rng.choice([0, 100, 1000]))
        img = synth.synth_frame(int(rng.integers(0, 1 << 30)), w, h, n_rect, n_small)
        if rng.random() < 0.2:
            img = rng.integers(0, 256, (h, w), dtype=np.uint8)
        try:
            k0, d0 = oracle.OrbOracle(nfeat, sf, nlev, ini, mn).extract(img)
        except ValueError:
            continue                                           # geometry neither side supports
        k1, d1 = gpu.ORBextractor(nfeat, sf, nlev, ini, mn)(img)
        assert np.array_equal(k0, k1) and np.array_equal(d0, d1), (w, h, nfeat, sf, nlev, ini, mn)
        done += 1


@pytest.mark.parametrize("seed,cfg", [(9000, (1000, 1.2, 8, 20, 7)), (9001, (2000, 1.2, 8, 20, 7)), (9002, (800, 1.3, 5, 20, 7))])
def test_compute_stereo_matches(gpu, oracle, seed, cfg):
    """Frame::ComputeStereoMatches (src/Frame.cc:841-1013) on the device-resident pyramids of the two extractors:
    mvuRight / mvDepth bit-exact against the oracle, and most keypoints recover their band's disparity."""
    left, right = synth.synth_stereo_pair(seed)
    el, er = gpu.ORBextractor(*cfg), gpu.ORBextractor(*cfg)
    kl, dl = el(left)
    kr, dr = er(right)
    ol, orr = oracle.OrbOracle(*cfg), oracle.OrbOracle(*cfg)
    okl, odl = ol.extract(left)
    okr, odr = orr.extract(right)
    assert np.array_equal(kl, okl) and np.array_equal(kr, okr)
    bf = np.float32(40.0)
    mb = np.float32(40.0 / 535.4)
    ur, dp = gpu.compute_stereo_matches(el, er, kl, dl, kr, dr, mb, bf)
    our, odp = oracle.stereo_matches(ol, orr, okl, odl, okr, odr, mb, bf)
    assert np.array_equal(ur, our) and np.array_equal(dp, odp)
    m = ur >= 0
    assert m.sum() > 0.3 * len(kl)
    disp = kl["x"][m] - ur[m]
    assert np.all(disp >= 0) and np.all(disp < bf / mb) and np.median(np.abs(disp - np.round(disp))) < 0.35


@pytest.mark.parametrize("w,h,pad,shift", [(487, 360, 0, 0), (640, 480, 3, 0), (642, 400, 2, 1), (640, 480, 0, 0)])
def test_device_api_unaligned_frames(gpu, oracle, w, h, pad, shift):
    """eao_orb_extract_batch_device reads the caller's frames in place: odd widths / pitches and a base pointer that is not
    4-byte aligned take the byte-wise variants of the resize, FAST staging, blur and orientation loads; the result must not
    depend on the layout (3 frames, bit-exact against the oracle; the aligned case runs through the same entry point)."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so.7")          # the runtime the library itself is linked against (no torch in this module)
    hip.hipMalloc.argtypes = [C.c_void_p, C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]

    def dmalloc(nbytes):
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), nbytes) == 0
        return p
    B = 3
    frames = np.stack([synth.synth_frame(4200 + f, w, h) for f in range(B)])
    pitch = w + pad
    host = np.zeros(B * h * pitch + 8, np.uint8)
    host[shift:shift + B * h * pitch].reshape(B, h, pitch)[:, :, :w] = frames
    d_img = dmalloc(host.nbytes)
    assert hip.hipMemcpy(d_img, host.ctypes.data, host.nbytes, 1) == 0
    ext = gpu.ORBextractor(1000, 1.2, 8, 20, 7)
    cap = ext.max_keypoints(w, h)
    d_k, d_d, d_n = dmalloc(B * cap * 28), dmalloc(B * cap * 32), dmalloc(B * 4)
    ext.extract_batch_device(d_img.value + shift, w, h, pitch, h * pitch, B, d_k.value, d_d.value, cap, d_n.value, None)
    assert hip.hipDeviceSynchronize() == 0
    n = np.zeros(B, np.int32)
    kps = np.zeros((B, cap), gpu.KP_DTYPE)
    desc = np.zeros((B, cap, 32), np.uint8)
    assert hip.hipMemcpy(n.ctypes.data, d_n, n.nbytes, 2) == 0 and hip.hipMemcpy(kps.ctypes.data, d_k, kps.nbytes, 2) == 0
    assert hip.hipMemcpy(desc.ctypes.data, d_d, desc.nbytes, 2) == 0
    for p_ in (d_img, d_k, d_d, d_n):
        hip.hipFree(p_)
    orc = oracle.OrbOracle(1000, 1.2, 8, 20, 7)
    for f in range(B):
        k0, d0 = orc.extract(frames[f])
        assert n[f] == len(k0) and np.array_equal(kps[f, :n[f]], k0) and np.array_equal(desc[f, :n[f]], d0), (f, w, h, pad, shift)


@pytest.mark.parametrize("cfg", [(989, 353, 3000, 1.5, 2, 20, 3, 400, 1000), (714, 502, 3000, 2.0, 2, 50, 3, 40, 1000), (640, 480, 5000, 1.2, 3, 20, 7, 400, 1000)])
def test_thousands_of_features_on_few_levels(gpu, oracle, cfg):
    """Quotas whose quad-tree node lists do not fit in LDS (3000 features on two levels need 150+ KB): upstream's
    DistributeOctTree takes any N (reference src/ORBextractor.cc:539) -- the same algorithm then runs over a global workspace."""
    w, h, nfeat, sf, nlev, ini, mn, n_rect, n_small = cfg
    img = synth.synth_frame(4242, w, h, n_rect, n_small)
    k0, d0 = oracle.OrbOracle(nfeat, sf, nlev, ini, mn).extract(img)
    k1, d1 = gpu.ORBextractor(nfeat, sf, nlev, ini, mn)(img)
    assert len(k0) == len(k1) and len(k0) > 1000
    assert np.array_equal(k0, k1) and np.array_equal(d0, d1)


def test_device_api_alternating_streams(gpu, oracle):
    """One handle, consecutive device calls on DIFFERENT streams: the calls share the handle's pyramid and candidate scratch, so a
    call on another stream than the previous one first waits for it (on the host).  Four calls alternating between two torch
    streams, no synchronisation in between from the caller: every result equals the oracle's."""
    imgs = synth.synth_frames(4, seed0=1300)
    ext = gpu.ORBextractor(1000, 1.2, 8, 20, 7)
    orc = oracle.OrbOracle(1000, 1.2, 8, 20, 7)
    H, W = imgs.shape[1:]
    cap = ext.max_keypoints(W, H)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    d_img = [torch.from_numpy(imgs[f:f + 1].copy()).cuda() for f in range(4)]
    d_k = [torch.zeros((1, cap, 28), dtype=torch.uint8, device="cuda") for _ in range(4)]
    d_d = [torch.zeros((1, cap, 32), dtype=torch.uint8, device="cuda") for _ in range(4)]
    d_n = [torch.zeros(1, dtype=torch.int32, device="cuda") for _ in range(4)]
    torch.cuda.synchronize()
    for f in range(4):
        st = streams[f & 1]
        ext.extract_batch_device(d_img[f].data_ptr(), W, H, W, W * H, 1, d_k[f].data_ptr(), d_d[f].data_ptr(), cap, d_n[f].data_ptr(), st.cuda_stream)
    torch.cuda.synchronize()
    for f in range(4):
        okps, odesc = orc.extract(imgs[f])
        n = int(d_n[f].item())
        assert n == len(okps)
        assert np.array_equal(d_k[f].cpu().numpy()[0, :n].reshape(-1).view(gpu.KP_DTYPE), okps), "frame %d keypoints" % f
        assert np.array_equal(d_d[f].cpu().numpy()[0, :n], odesc), "frame %d descriptors" % f


def test_streaming_host_api_equals_extract_batch(gpu):
    """eao_orb_stream_*: pinned slots, asynchronous submit, three streams.  Five different batches through a ring of three slots
    (the upload of one overlaps the extraction of another), submitted ahead of the waits and with a ragged last batch: every
    frame's keypoints and descriptors equal eao_orb_extract_batch's, bit for bit; a slot can be refilled after its wait."""
    ext = gpu.ORBextractor(1000, 1.2, 8, 20, 7)
    ref = gpu.ORBextractor(1000, 1.2, 8, 20, 7)
    B = 8
    slots = ext.stream_create(640, 480, B, 3)
    batches = [synth.synth_frames(B, seed0=4000 + 10 * k) for k in range(5)]
    batches[3][2] = 77                     # a frame without corners
    want = [ref.extract_batch(b) for b in batches]
    sizes = [B, B, B, B, 5]                # the last batch is short
    got = [None] * 5
    for k in range(5 + 2):
        if k < 5:
            s = k % 3
            if k >= 3:                      # the slot is reused: its previous results were collected two steps ago
                assert got[k - 3] is not None
            slots[s]["frames"][:sizes[k]] = batches[k][:sizes[k]]
            ext.stream_submit(s, sizes[k])
        if k >= 2:
            j = k - 2
            kk, dd = ext.stream_wait(j % 3)
            got[j] = ([x.copy() for x in kk[:sizes[j]]], [x.copy() for x in dd[:sizes[j]]])
    for k in range(5):
        for f in range(sizes[k]):
            assert np.array_equal(got[k][0][f], want[k][0][f]), "batch %d frame %d keypoints" % (k, f)
            assert np.array_equal(got[k][1][f], want[k][1][f]), "batch %d frame %d descriptors" % (k, f)
    with pytest.raises(gpu.EaoError):
        ext.stream_submit(7, B)             # no such slot
