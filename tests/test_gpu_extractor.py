"""GPU parity: ORBextractor path (HIP, through the C ABI) vs the CPU oracle. Bit-exact."""
import numpy as np
import pytest

from conftest import corners_to_array

pytestmark = pytest.mark.gpu

FIELDS = ("x", "y", "size", "angle", "response", "octave", "class_id")


@pytest.fixture(autouse=True, params=["default", "direct_any_batch", "fast_early_out", "padded_level0"])
def fast_stage_variant(request, monkeypatch):
    """Every test of this module runs four times: as shipped (level 0 read straight from the caller's image -- direct mode
    -- for aligned batches of 8 frames or more, the round-3 FAST kernel), with direct mode for any batch size, with the
    FAST kernel's exact wave-level early-out on top of that (k_fast_detect<true>), and with direct mode off (level 0
    always copied into a padded plane, the round-3 data path).  The environment variables are read when a handle is created."""
    for v in ("ORBGPU_FAST_EARLY_OUT", "ORBGPU_DEBUG_NO_DIRECT0", "ORBGPU_DEBUG_DIRECT0_MIN"):
        monkeypatch.delenv(v, raising=False)
    if request.param in ("direct_any_batch", "fast_early_out"):
        monkeypatch.setenv("ORBGPU_DEBUG_DIRECT0_MIN", "1")
    if request.param == "fast_early_out":
        monkeypatch.setenv("ORBGPU_FAST_EARLY_OUT", "1")
    elif request.param == "padded_level0":
        monkeypatch.setenv("ORBGPU_DEBUG_NO_DIRECT0", "1")
    return request.param


def assert_same_keypoints(gk, gd, ok, od, what):
    assert len(gk) == len(ok), "%s: key point count %d != oracle %d" % (what, len(gk), len(ok))
    for f in FIELDS:
        a = np.ascontiguousarray(gk[f]).view(np.uint32)
        b = np.ascontiguousarray(ok[f]).view(np.uint32)
        bad = np.nonzero(a != b)[0]
        assert len(bad) == 0, "%s: field %s differs at %d rows, first %d: gpu %r oracle %r" % (
            what, f, len(bad), bad[0], gk[f][bad[0]], ok[f][bad[0]])
    assert gd.shape == od.shape and np.array_equal(gd, od), "%s: descriptor bits differ in %d rows" % (
        what, int((gd != od).any(1).sum()))


def check_stages(gpu, ge, oe, frame, nlevels, what):
    """Pyramid bytes, blurred bytes, FAST candidates and quadtree selection, level by level."""
    for l in range(nlevels):
        raw, pitch = ge.debug_read(gpu.DBG_PYRAMID_PADDED, frame, l)
        op = oe.pyramid_level(l)
        hh, ww = op.shape
        gp = raw.reshape(-1, pitch)[:hh, :ww]
        assert np.array_equal(gp, op), "%s L%d: padded pyramid differs at %d px" % (what, l, int((gp != op).sum()))
        ob = oe.blurred_level(l)
        if ob is not None:
            braw, _ = ge.debug_read(gpu.DBG_BLURRED_PADDED, frame, l)
            gb = braw.reshape(-1, pitch)[19:hh - 19, 19:ww - 19]
            assert np.array_equal(gb, ob), "%s L%d: blurred level differs at %d px" % (what, l, int((gb != ob).sum()))
        gc, _ = ge.debug_read(gpu.DBG_CANDIDATES, frame, l)
        oc = corners_to_array(oe.level_candidates(l))
        assert gc.shape == oc.shape and np.array_equal(gc, oc), "%s L%d: FAST candidates differ (%d vs %d)" % (
            what, l, len(gc), len(oc))
        gs, _ = ge.debug_read(gpu.DBG_SELECTED, frame, l)
        os_ = corners_to_array(oe.level_selected(l))
        assert gs.shape == os_.shape and np.array_equal(gs, os_), "%s L%d: quadtree selection differs (%d vs %d)" % (
            what, l, len(gs), len(os_))


@pytest.mark.parametrize("w,h,nfeat,batch", [(640, 480, 1000, 3), (1280, 960, 2000, 2), (640, 480, 1000, 9)])
def test_extract_matches_oracle_stage_by_stage(gpu, oracle, w, h, nfeat, batch):
    """C2 / C3 configurations (BASELINE.json configs[1], configs[2]) on the seeded synthetic stream."""
    from orb_slam2_map_amd.synth import Stream
    st = Stream(w, h, 1234)
    imgs = st.gray_batch(0, batch)
    ge = gpu.ORBextractor(nfeat, max_batch=batch)
    oe = oracle.Extractor(nfeat)
    gk, gd = ge.extract_batch(imgs)
    for f in range(batch):
        ok, od = oe.extract(imgs[f])
        check_stages(gpu, ge, oe, f, 8, "%dx%d frame %d" % (w, h, f))
        assert_same_keypoints(gk[f], gd[f], ok, od, "%dx%d frame %d" % (w, h, f))
        assert len(gk[f]) <= ge.max_keypoints(w, h)


@pytest.mark.parametrize("nfeat,sf,nl,ini,mn", [(500, 1.5, 5, 20, 7), (1500, 1.1, 12, 20, 7), (1000, 1.2, 8, 30, 10),
                                                 (800, 2.0, 3, 12, 12), (1000, 1.2, 8, 7, 20), (300, 1.3, 6, 40, 0),
                                                 (1000, 2.5, 3, 20, 7)])
def test_non_default_parameters(gpu, oracle, stream640, nfeat, sf, nl, ini, mn):
    """Other YAML settings (Tracking.cc:193-197): scale factors (2.5 takes the generic resize kernel),
    level counts, thresholds -- including iniThFAST < minThFAST and a zero threshold."""
    img = stream640.frame(9)[0]
    ge = gpu.ORBextractor(nfeat, sf, nl, ini, mn)
    oe = oracle.Extractor(nfeat, sf, nl, ini, mn)
    gk, gd = ge(img)
    ok, od = oe.extract(img)
    check_stages(gpu, ge, oe, 0, nl, "params %s" % ((nfeat, sf, nl, ini, mn),))
    assert_same_keypoints(gk, gd, ok, od, "params %s" % ((nfeat, sf, nl, ini, mn),))


@pytest.mark.parametrize("w,h,sf,nl,batch", [(176, 144, 1.1, 8, 1), (176, 144, 1.1, 8, 40), (220, 170, 1.15, 8, 3)])
def test_small_image_many_levels(gpu, oracle, w, h, sf, nl, batch):
    """Small frames with many levels: the top levels are a handful of work items each, so one wave of k_fast_detect spans
    three or four levels and reserves a range in each level's key array (both quadtree variants: batch 40 and single)."""
    from orb_slam2_map_amd.synth import Stream
    st = Stream(w, h, 77)
    imgs = np.stack([st.frame(t)[0] for t in range(batch)])
    ge = gpu.ORBextractor(400, sf, nl, 20, 7, max_batch=batch)
    oe = oracle.Extractor(400, sf, nl, 20, 7)
    gk, gd = ge.extract_batch(imgs)
    for f in (0, batch - 1):
        ok, od = oe.extract(imgs[f])
        if f == batch - 1:
            check_stages(gpu, ge, oe, f, nl, "small %dx%d frame %d" % (w, h, f))
        assert_same_keypoints(gk[f], gd[f], ok, od, "small %dx%d frame %d" % (w, h, f))


def test_reconfigure_between_sizes(gpu, oracle):
    """One handle, images of different sizes in sequence (the geometry tables are rebuilt)."""
    from orb_slam2_map_amd.synth import Stream
    ge = gpu.ORBextractor(1000, max_batch=2)
    oe = oracle.Extractor(1000)
    for (w, h, seed) in ((640, 480, 5), (400, 300, 6), (752, 480, 7), (640, 480, 8)):
        imgs = Stream(w, h, seed).gray_batch(0, 2)
        gk, gd = ge.extract_batch(imgs)
        for f in range(2):
            ok, od = oe.extract(imgs[f])
            assert_same_keypoints(gk[f], gd[f], ok, od, "%dx%d seed %d frame %d" % (w, h, seed, f))


def test_getters_match_oracle(gpu, oracle):
    for nfeat, sf, nl in ((1000, 1.2, 8), (2000, 1.2, 8), (500, 1.5, 5), (1500, 1.1, 12)):
        ge = gpu.ORBextractor(nfeat, sf, nl)
        oe = oracle.Extractor(nfeat, sf, nl)
        assert ge.GetLevels() == nl
        assert np.float32(ge.GetScaleFactor()) == np.float32(sf)
        assert np.array_equal(ge.GetScaleFactors(), oe.scale_factors())
        assert np.array_equal(ge.GetInverseScaleFactors(), oe.inv_scale_factors())
        assert np.array_equal(ge.GetScaleSigmaSquares(), oe.sigma2())
        assert np.array_equal(ge.GetInverseScaleSigmaSquares(), oe.inv_sigma2())
        assert np.array_equal(ge.quotas(), oe.quotas())


@pytest.mark.parametrize("w,h", [(333, 251), (752, 480), (1241, 376), (320, 240), (4000, 1200)])
def test_ragged_sizes(gpu, oracle, w, h):
    """Sizes that are not multiples of anything (EuRoC 752x480, KITTI 1241x376, odd), and one near the 4096-pixel limit of
    the key format (132 x 38 cells at level 0: the cell indices of the quadtree's order key need 8 bits)."""
    from orb_slam2_map_amd.synth import Stream
    st = Stream(w, h, 77)
    img = st.frame(3)[0]
    ge = gpu.ORBextractor(1000)
    oe = oracle.Extractor(1000)
    gk, gd = ge(img)
    ok, od = oe.extract(img)
    check_stages(gpu, ge, oe, 0, 8, "%dx%d" % (w, h))
    assert_same_keypoints(gk, gd, ok, od, "%dx%d" % (w, h))


def test_strided_input_and_single_call(gpu, oracle, stream640):
    """operator() on a cv::Mat ROI: row stride larger than the width."""
    big = np.zeros((480, 700), np.uint8)
    img = stream640.frame(5)[0]
    big[:, :640] = img
    view = big[:, :640]
    import ctypes as C
    ge = gpu.ORBextractor(1000)
    cap = ge.max_keypoints(640, 480)
    kps = np.zeros(cap, gpu.KEYPOINT_DTYPE)
    desc = np.zeros((cap, 32), np.uint8)
    n = C.c_int32()
    gpu.check(ge.L.orbgpu_extract(ge.h, view.ctypes.data_as(C.c_void_p), 640, 480, view.strides[0],
                                  kps.ctypes.data_as(C.c_void_p), desc.ctypes.data_as(C.c_void_p), cap, C.byref(n)))
    ok, od = oracle.Extractor(1000).extract(img)
    assert_same_keypoints(kps[:n.value], desc[:n.value], ok, od, "strided")


def test_low_texture_threshold_fallback(gpu, oracle):
    """Smooth image with faint structure: most cells fall back to minThFAST (ORBextractor.cc:812-816),
    fewer candidates than the quota, so the quadtree ends with singleton nodes."""
    rng = np.random.default_rng(3)
    yy, xx = np.mgrid[0:480, 0:640]
    img = 120 + 20 * np.sin(xx / 37.0) + 15 * np.cos(yy / 23.0)
    for _ in range(60):
        cx, cy = rng.integers(30, 610), rng.integers(30, 450)
        img[cy:cy + 9, cx:cx + 9] += rng.integers(9, 16)
    img = np.clip(np.rint(img), 0, 255).astype(np.uint8)
    ge = gpu.ORBextractor(1000)
    oe = oracle.Extractor(1000)
    gk, gd = ge(img)
    ok, od = oe.extract(img)
    assert 0 < len(ok) < 900
    check_stages(gpu, ge, oe, 0, 8, "low texture")
    assert_same_keypoints(gk, gd, ok, od, "low texture")


def test_constant_and_empty_images(gpu, oracle):
    ge = gpu.ORBextractor(1000)
    gk, gd = ge(np.full((480, 640), 128, np.uint8))
    assert len(gk) == 0 and gd.shape == (0, 32)
    gk, gd = ge(np.zeros((0, 0), np.uint8))  # ORBextractor.cc:1046: empty image -> no output
    assert len(gk) == 0


def test_dense_corners_high_candidate_count(gpu, oracle):
    """Checkerboard + noise: far more candidates than the quota in every cell."""
    rng = np.random.default_rng(11)
    yy, xx = np.mgrid[0:480, 0:640]
    img = ((xx // 6 + yy // 6) % 2) * 120 + 60 + rng.integers(-8, 9, (480, 640))
    img = np.clip(img, 0, 255).astype(np.uint8)
    ge = gpu.ORBextractor(1000)
    oe = oracle.Extractor(1000)
    gk, gd = ge(img)
    ok, od = oe.extract(img)
    check_stages(gpu, ge, oe, 0, 8, "checkerboard")
    assert_same_keypoints(gk, gd, ok, od, "checkerboard")


def test_full_value_range_extremes(gpu, oracle):
    """Salt-and-pepper blobs and uniform noise over the whole 0..255 range: ring values 0 and 255, centre minus ring
    differences up to +-255 and corner scores up to the 255 clamp.  The score network compares the 16-bit integer
    lanes as f16 bit patterns (subnormals), so the extremes of the range are the cases to pin."""
    rng = np.random.default_rng(29)
    img = rng.integers(0, 256, (480, 640)).astype(np.uint8)
    img[100:200, 100:300] = 0
    img[250:330, 330:600] = 255
    for _ in range(400):
        cx, cy = rng.integers(20, 620), rng.integers(20, 460)
        img[cy:cy + rng.integers(1, 4), cx:cx + rng.integers(1, 4)] = rng.choice([0, 255])
    ge = gpu.ORBextractor(1000)
    oe = oracle.Extractor(1000)
    gk, gd = ge(img)
    ok, od = oe.extract(img)
    assert len(ok) > 900 and ok["response"].max() >= 200
    check_stages(gpu, ge, oe, 0, 8, "extremes")
    assert_same_keypoints(gk, gd, ok, od, "extremes")


@pytest.mark.parametrize("qcap", [0, 5, 64])
def test_fast_key_queue_overflow(gpu, oracle, stream640, qcap, monkeypatch):
    """k_fast_detect queues rows with corners in LDS and falls back to one atomic per key when a wave's queue is full.
    The test hook shrinks the queue so that every wave takes the fallback (0), or mixes both paths (5, 64)."""
    monkeypatch.setenv("ORBGPU_DEBUG_FAST_QUEUE", str(qcap))
    ge = gpu.ORBextractor(1000)
    monkeypatch.delenv("ORBGPU_DEBUG_FAST_QUEUE")
    oe = oracle.Extractor(1000)
    rng = np.random.default_rng(5)
    noise = rng.integers(0, 256, (480, 640)).astype(np.uint8)
    for what, img in (("stream", stream640.frame(0)[0]), ("noise", noise)):
        gk, gd = ge(img)
        ok, od = oe.extract(img)
        check_stages(gpu, ge, oe, 0, 8, "queue %d %s" % (qcap, what))
        assert_same_keypoints(gk, gd, ok, od, "queue %d %s" % (qcap, what))


def test_too_small_image_is_rejected(gpu):
    ge = gpu.ORBextractor(1000)
    with pytest.raises(gpu.OrbGpuError) as ei:
        ge(np.zeros((120, 160), np.uint8))
    assert ei.value.status == gpu.EINVAL


def test_capacity_error(gpu, stream640):
    import ctypes as C
    ge = gpu.ORBextractor(1000)
    img = stream640.frame(0)[0]
    kps = np.zeros(100, gpu.KEYPOINT_DTYPE)
    desc = np.zeros((100, 32), np.uint8)
    n = C.c_int32()
    rc = ge.L.orbgpu_extract(ge.h, img.ctypes.data_as(C.c_void_p), 640, 480, 640, kps.ctypes.data_as(C.c_void_p),
                             desc.ctypes.data_as(C.c_void_p), 100, C.byref(n))
    assert rc == gpu.ECAPACITY


def test_golden_fixture_on_gpu(gpu):
    """Committed golden vectors (tests/golden, produced by the oracle) replayed on the GPU."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "extract_640x480_seed1234_f0.npz"))
    from orb_slam2_map_amd.synth import Stream
    img = Stream(640, 480, 1234).frame(0)[0]
    gk, gd = gpu.ORBextractor(1000)(img)
    assert len(gk) == len(g["x"])
    for f in FIELDS:
        assert np.array_equal(np.ascontiguousarray(gk[f]).view(np.uint32), g[f].view(np.uint32)), f
    assert np.array_equal(gd, g["desc"])


def test_batch_equals_single_and_is_deterministic(gpu, oracle, stream640):
    """Size-independent properties at a bench-like batch size: batched == per-frame == oracle, run twice == same.
    40 frames: batches of 32 and more take the many-workgroups variant of the quadtree kernel, single frames the one that
    keeps a level's keys in LDS."""
    imgs = stream640.gray_batch(10, 40)
    ge = gpu.ORBextractor(1000, max_batch=40)
    k1, d1 = ge.extract_batch(imgs)
    k2, d2 = ge.extract_batch(imgs)
    single = gpu.ORBextractor(1000)
    for f in range(40):
        assert np.array_equal(k1[f], k2[f]) and np.array_equal(d1[f], d2[f])
    for f in (0, 7, 39):
        ks, ds = single(imgs[f])
        assert np.array_equal(ks, k1[f]) and np.array_equal(ds, d1[f])
    oe = oracle.Extractor(1000)
    for f in (3, 33):
        ok, od = oe.extract(imgs[f])
        check_stages(gpu, ge, oe, f, 8, "batch frame %d" % f)
        assert_same_keypoints(k1[f], d1[f], ok, od, "batch frame %d" % f)
    # structural invariants of the reference's output (E3', E4, E8)
    for f in range(40):
        k = k1[f]
        assert 900 <= len(k) <= ge.max_keypoints(640, 480)
        assert np.all(np.diff(k["octave"]) >= 0), "levels are concatenated in order"
        assert np.all((k["angle"] >= 0) & (k["angle"] < 360))
        assert set(np.unique(k["size"])) <= {31, 37, 44, 53, 64, 77, 92, 111}


def test_pyramid_getter(gpu, oracle, stream640):
    import ctypes as C
    img = stream640.frame(2)[0]
    ge = gpu.ORBextractor(1000)
    ge(img)
    oe = oracle.Extractor(1000)
    oe.extract(img)
    for l in (0, 3, 7):
        op = oe.pyramid_level(l)
        hh, ww = op.shape[0] - 38, op.shape[1] - 38
        dst = np.zeros((hh, ww), np.uint8)
        w_, h_ = C.c_int32(), C.c_int32()
        gpu.check(ge.L.orbgpu_extractor_get_pyramid_level(ge.h, 0, l, dst.ctypes.data_as(C.c_void_p), ww, C.byref(w_),
                                                          C.byref(h_)))
        assert (w_.value, h_.value) == (ww, hh)
        assert np.array_equal(dst, op[19:19 + hh, 19:19 + ww])


def test_host_entry_replays_a_graph(gpu, oracle, stream640):
    """orbgpu_extract captures its launch sequence as a hipGraph on the second call of a configuration and replays
    it afterwards; results stay bit-exact, and a change of image size re-captures."""
    ge = gpu.ORBextractor(1000)
    oe = oracle.Extractor(1000)
    for t in range(4):
        g = stream640.frame(20 + t)[0]
        k, d = ge(g)
        ok, od = oe.extract(g)
        assert k.tobytes() == ok.tobytes() and np.array_equal(d, od)
    assert ge.graph_state() == 1, "the graph path must be in use after the second call (state %d)" % ge.graph_state()
    small = np.ascontiguousarray(stream640.frame(3)[0][:300, :400])
    for _ in range(3):
        k, d = ge(small)
        ok, od = oe.extract(small)
        assert k.tobytes() == ok.tobytes() and np.array_equal(d, od)
    assert ge.graph_state() == 1
    ge.set_profiling(True)  # stage events between the launches: plain launches, same results
    k, d = ge(small)
    assert k.tobytes() == ok.tobytes()


@pytest.mark.parametrize("w,h,nfeat", [(640, 480, 1000), (1280, 960, 2000)])
@pytest.mark.parametrize("kcap", [0, 8, 512, "last"])
def test_quadtree_keys_beyond_lds(gpu, oracle, w, h, nfeat, kcap, monkeypatch):
    """Single frames: k_quadtree keeps the first keys of a level in LDS and the rest in memory; the test hook
    (ORBGPU_DEBUG_QT_KEYS, read by orbgpu_extractor_create) shrinks the LDS share so that both kinds of key take part
    in every sweep.  kcap 512 at 1280x960 / 2000 features are the parameters of the run that faulted in round 2's
    scratch (gpurun_out/qts_512.log); "last" puts the boundary 1..8 keys before the end of level 0."""
    from orb_slam2_map_amd.synth import Stream
    img = Stream(w, h, 1234).frame(2)[0]
    oe = oracle.Extractor(nfeat)
    ok, od = oe.extract(img)
    if kcap == "last":
        kcap = (len(oe.level_candidates(0)) - 1) // 8 * 8  # (the share is a multiple of 8): 1..8 keys stay in memory
    monkeypatch.setenv("ORBGPU_DEBUG_QT_KEYS", str(kcap))
    ge = gpu.ORBextractor(nfeat)
    monkeypatch.delenv("ORBGPU_DEBUG_QT_KEYS")
    for rep in range(3):  # plain launches, graph capture, graph replay
        gk, gd = ge(img)
        lds_keys, threads, lds_bytes = ge.quadtree_config(1)
        assert lds_keys == (kcap + 7) // 8 * 8, "the hook did not take: %d keys in LDS, asked for %d" % (lds_keys, kcap)
        assert threads == (1024 if w * h >= 700000 else 512) and lds_bytes <= 159 * 1024
        check_stages(gpu, ge, oe, 0, 8, "%dx%d qt keys %d rep %d" % (w, h, kcap, rep))
        assert_same_keypoints(gk, gd, ok, od, "%dx%d qt keys %d rep %d" % (w, h, kcap, rep))
    assert len(oe.level_candidates(0)) > lds_keys, "level 0 must have keys beyond the LDS share"


@pytest.mark.parametrize("w,h,nfeat", [(640, 480, 1000), (1280, 960, 2000), (1000, 200, 600)])
def test_single_frame_quadtree_with_and_without_prefilter(gpu, oracle, w, h, nfeat, monkeypatch):
    """Single frames filter / count their FAST keys in a device-wide kernel of its own (k_qt_prefilter) before the
    quadtree workgroups start; ORBGPU_DEBUG_QT_NOPRE keeps that sweep inside k_quadtree<true>.  Both equal the oracle
    (the wide image has five initial nodes per level: the bin -> node mapping of the prefiltered path)."""
    from orb_slam2_map_amd.synth import Stream
    imgs = [Stream(w, h, 77).frame(t)[0] for t in (0, 3)]
    oe = oracle.Extractor(nfeat, 1.2, 8 if h >= 480 else 4)
    for nopre in (False, True):
        if nopre:
            monkeypatch.setenv("ORBGPU_DEBUG_QT_NOPRE", "1")
        ge = gpu.ORBextractor(nfeat, 1.2, 8 if h >= 480 else 4)
        monkeypatch.delenv("ORBGPU_DEBUG_QT_NOPRE", raising=False)
        for rep in range(2):
            for img in imgs:  # alternating frames: the counters a call leaves behind must be clean for the next
                gk, gd = ge(img)
                ok, od = oe.extract(img)
                check_stages(gpu, ge, oe, 0, oe.nlevels if hasattr(oe, "nlevels") else (8 if h >= 480 else 4), "%dx%d nopre %d" % (w, h, nopre))
                assert_same_keypoints(gk, gd, ok, od, "%dx%d nopre %d rep %d" % (w, h, nopre, rep))


def test_two_handles_share_the_quadtree_kernel(gpu, oracle, stream640, stream1280):
    """The dynamic-LDS limit of k_quadtree is a property of the kernel, not of a handle: a handle configured for a
    small geometry must not take away what a handle with a large one launches with."""
    big, small = gpu.ORBextractor(2000), gpu.ORBextractor(300)
    img_b, img_s = stream1280.frame(1)[0], np.ascontiguousarray(stream640.frame(1)[0][:240, :320])
    kb, db = big(img_b)
    ks, ds = small(img_s)      # configures (and used to lower the limit) after `big`
    kb2, db2 = big(img_b)      # must still launch
    assert kb.tobytes() == kb2.tobytes() and np.array_equal(db, db2)
    ok, od = oracle.Extractor(2000).extract(img_b)
    assert_same_keypoints(kb2, db2, ok, od, "big handle after a small one configured")
    ok, od = oracle.Extractor(300).extract(img_s)
    assert_same_keypoints(ks, ds, ok, od, "small handle")


def test_device_input_with_unaligned_rows(gpu, oracle, stream640):
    """The level-0 border kernel has a table-driven fast path for 4-byte aligned rows; a device image whose rows start
    at odd addresses (stride 641, base pointer + 1) takes the byte-wise kernel and must give the same frame."""
    torch = pytest.importorskip("torch")
    img = stream640.frame(9)[0]
    ok, od = oracle.Extractor(1000).extract(img)
    ge = gpu.ORBextractor(1000)
    cap = ge.max_keypoints(640, 480)
    for stride, shift in ((641, 0), (640, 1), (644, 2), (640, 0)):
        buf = torch.zeros(480 * stride + 8, dtype=torch.uint8, device="cuda")
        view = buf[shift:shift + 480 * stride].view(480, stride)
        view[:, :640] = torch.from_numpy(img).cuda()
        kps = torch.zeros((1, cap, 7), dtype=torch.float32, device="cuda")
        desc = torch.zeros((1, cap, 32), dtype=torch.uint8, device="cuda")
        nout = torch.zeros(1, dtype=torch.int32, device="cuda")
        ge.extract_batch_device(buf.data_ptr() + shift, 1, 640, 480, stride, 480 * stride, kps.data_ptr(), desc.data_ptr(),
                                cap, nout.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        n = int(nout[0])
        gk = kps[0, :n].cpu().numpy().view(gpu.KEYPOINT_DTYPE).reshape(-1)
        assert_same_keypoints(gk, desc[0, :n].cpu().numpy(), ok, od, "stride %d shift %d" % (stride, shift))


@pytest.mark.parametrize("w,h,nfeat,flat", [(640, 480, 1000, 0.6), (1280, 960, 2000, 0.45), (640, 480, 1000, 0.95)])
def test_fast_early_out_on_images_with_flat_regions(gpu, oracle, w, h, nfeat, flat, fast_stage_variant):
    """The early-out only ever fires where a wavefront's whole row segment is clear of corner candidates: a stream with
    flat regions (Stream(..., flat_fraction)), a constant image and a smooth ramp, stage by stage against the oracle; the
    option toggled on one handle gives the same bytes as the other kernel."""
    from orb_slam2_map_amd.synth import Stream
    st = Stream(w, h, 4321, flat_fraction=flat)
    yy, xx = np.mgrid[0:h, 0:w]
    imgs = np.stack([st.frame(2)[0], st.frame(3)[0], np.full((h, w), 117, np.uint8), ((xx + yy) // 9).astype(np.uint8)])
    ge = gpu.ORBextractor(nfeat, max_batch=len(imgs))
    oe = oracle.Extractor(nfeat)
    for on in (fast_stage_variant == "fast_early_out", fast_stage_variant != "fast_early_out"):
        ge.set_fast_early_out(on)
        gk, gd = ge.extract_batch(imgs)
        for f in range(len(imgs)):
            ok, od = oe.extract(imgs[f])
            check_stages(gpu, ge, oe, f, 8, "flat %.2f %dx%d frame %d early-out %s" % (flat, w, h, f, on))
            assert_same_keypoints(gk[f], gd[f], ok, od, "flat %.2f %dx%d frame %d early-out %s" % (flat, w, h, f, on))
    assert len(gk[0]) > nfeat // 4  # the textured part still fills most of the quota


def test_direct_mode_level0_is_materialised_on_demand(gpu, oracle, stream640, monkeypatch):
    """Device-resident, aligned input: no stage writes level 0 of the padded pyramid (the kernels read the image); the
    getters of mvImagePyramid[0] and of the padded debug view produce it when asked, from the image of the last call.
    Strides that are multiples of 4 but not of 64, and a stride > width, stay on the direct path."""
    import torch
    monkeypatch.setenv("ORBGPU_DEBUG_DIRECT0_MIN", "1")  # (two frames: below the shipped threshold of 8)
    g0, g1 = stream640.frame(21)[0], stream640.frame(22)[0]
    oe = oracle.Extractor(1000)
    for stride in (640, 644, 704):
        ge = gpu.ORBextractor(1000, max_batch=2)
        cap = ge.max_keypoints(640, 480)
        buf = torch.zeros((2, 480, stride), dtype=torch.uint8, device="cuda")
        buf[0, :, :640] = torch.from_numpy(g0).cuda()
        buf[1, :, :640] = torch.from_numpy(g1).cuda()
        kps = torch.zeros((2, cap, 7), dtype=torch.float32, device="cuda")
        desc = torch.zeros((2, cap, 32), dtype=torch.uint8, device="cuda")
        n = torch.zeros(2, dtype=torch.int32, device="cuda")
        ge.extract_batch_device(buf.data_ptr(), 2, 640, 480, stride, 480 * stride, kps.data_ptr(), desc.data_ptr(), cap, n.data_ptr(),
                                torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        for f, g in enumerate((g0, g1)):
            ok, od = oe.extract(g)
            nf = int(n[f])
            gk = np.frombuffer(kps[f, :nf].cpu().numpy().tobytes(), gpu.KEYPOINT_DTYPE)
            assert_same_keypoints(gk, desc[f, :nf].cpu().numpy(), ok, od, "stride %d frame %d" % (stride, f))
            lvl0, w0, h0 = ge.get_pyramid_level(f, 0)
            assert (w0, h0) == (640, 480) and np.array_equal(lvl0, g)
            check_stages(gpu, ge, oe, f, 8, "stride %d frame %d" % (stride, f))
