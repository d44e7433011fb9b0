"""GPU parity for the background-thread matchers against the oracle, through the C ABI:
SearchByBoW(KF, KF), SearchForTriangulation, Fuse (both overloads, candidate phase) and SearchBySim3."""
import numpy as np
import pytest

import scenario
from test_gpu_matcher_proj import build_c3

pytestmark = pytest.mark.gpu


def _pts(mp, wp, rng, bad_frac=0.1):
    return {"bad": (rng.random(len(wp)) < bad_frac).astype(np.uint8), "world_pos": wp, "normal": mp["normal"],
            "min_dist": mp["min_dist"],
            # a hair below the creation distance ratio: a point seen exactly at its creation distance on the top level
            # predicts level nlevels (the reference then reads mvScaleFactors out of range; here ORBGPU_ELEVEL)
            "max_dist": (mp["max_dist"] * np.float32(0.9995)).astype(np.float32), "desc": mp["desc"]}


@pytest.mark.parametrize("th,seed", [(3.0, 5), (2.5, 6), (6.0, 7)])
def test_fuse_candidates(gpu, oracle, th, seed):
    """LocalMapping::SearchInNeighbors (LocalMapping.cc:489, 514): map points of neighbouring key frames projected
    into a key frame; the best key point per point after the level and chi-square gates, no claims."""
    st, Tcw, gf, of, mp, wp, dsc, octv, ang, cur_k = build_c3(gpu, oracle, 640, 480, 1000, 3, seed)
    rng = np.random.default_rng(seed)
    ge = gpu.ORBextractor(1000)
    inv_s2 = ge.GetInverseScaleSigmaSquares()
    log_sf = float(np.log(np.float32(ge.GetScaleFactors()[1])))
    pts = _pts(mp, wp, rng)
    fx, fy, cx, cy, bf = (float(v) for v in (st.fx, st.fy, st.cx, st.cy, st.bf))
    ng, bg = gpu.fuse(gf, Tcw, fx, fy, cx, cy, bf, log_sf, pts, th, inv_s2)
    no, bo = oracle.fuse(of, Tcw, fx, fy, cx, cy, bf, log_sf, pts, th, inv_s2)
    assert no > 200, no
    assert ng == no and np.array_equal(bg, bo), "%d vs %d candidates, %d rows differ" % (ng, no, int((bg != bo).sum()))
    # several points may pick the same key point: there are no claims in Fuse
    assert len(np.unique(bo[bo >= 0])) < int((bo >= 0).sum())


@pytest.mark.parametrize("th,scale", [(4.0, 1.0), (4.0, 1.7), (10.0, 0.6)])
def test_fuse_sim3_candidates(gpu, oracle, th, scale):
    st, Tcw, gf, of, mp, wp, dsc, octv, ang, cur_k = build_c3(gpu, oracle, 640, 480, 1000, 3, 21)
    rng = np.random.default_rng(int(10 * scale))
    ge = gpu.ORBextractor(1000)
    log_sf = float(np.log(np.float32(ge.GetScaleFactors()[1])))
    Scw = Tcw.copy()
    Scw[:3, :] *= np.float32(scale)
    pts = _pts(mp, wp, rng)
    fx, fy, cx, cy = (float(v) for v in (st.fx, st.fy, st.cx, st.cy))
    ng, bg = gpu.fuse_sim3(gf, Scw, fx, fy, cx, cy, log_sf, pts, th)
    no, bo = oracle.fuse_sim3(of, Scw, fx, fy, cx, cy, log_sf, pts, th)
    assert no > 200
    assert ng == no and np.array_equal(bg, bo)


def _two_keyframes(gpu, oracle, t0=40, t1=41):
    from orb_slam2_map_amd.synth import Stream
    st = Stream(640, 480, 1234)
    ge = gpu.ORBextractor(1000, max_batch=2)
    fr = [st.frame(t0), st.frame(t1)]
    ks, ds = ge.extract_batch(np.stack([f[0] for f in fr]))
    sf = ge.GetScaleFactors()
    g = [scenario.make_frame(gpu, ks[i], ds[i], fr[i][2], st, sf) for i in range(2)]
    o = [scenario.make_frame(oracle, ks[i], ds[i], fr[i][2], st, sf) for i in range(2)]
    (px, py), (qx, qy) = st.offset(t0), st.offset(t1)
    return st, ge, fr, ks, ds, g, o, (qx - px, qy - py)


def _cam_points(ks, depth, shift, st, rng, jitter=0.6):
    """Camera coordinates whose projection is the key point's pixel minus `shift` (+ jitter) at the key point's depth."""
    fx, fy, cx, cy = float(st.fx), float(st.fy), float(st.cx), float(st.cy)
    u = ks["x"].astype(np.float64) - shift[0] + rng.normal(0, jitter, len(ks))
    v = ks["y"].astype(np.float64) - shift[1] + rng.normal(0, jitter, len(ks))
    d = depth[ks["y"].astype(np.int64), ks["x"].astype(np.int64)].astype(np.float64)
    d = np.where(d > 0, d, 2.0)
    return np.stack([(u - cx) * d / fx, (v - cy) * d / fy, d], 1)


@pytest.mark.parametrize("th,s12,pre", [(7.5, 1.0, 0.0), (7.5, 1.15, 0.2), (15.0, 0.9, 0.1)])
def test_search_by_sim3(gpu, oracle, th, s12, pre):
    """LoopClosing::ComputeSim3 (LoopClosing.cc:324): both key frames' map points moved through the similarity
    x1 = s12 R12 x2 + t12 into the other frame, best key point in the window, kept where the two directions agree.
    Scenario: the map points of key frame 1 are placed so that the similarity carries them exactly onto the matching
    image content of key frame 2 (the stream is a pure image shift), and vice versa."""
    st, ge, fr, ks, ds, g, o, shift = _two_keyframes(gpu, oracle)
    rng = np.random.default_rng(int(th) + int(10 * s12))
    sf = np.asarray(ge.GetScaleFactors(), np.float32)
    log_sf = float(np.log(np.float32(sf[1])))
    T1w = scenario.rigid(0.02, -0.01, 0.03, (0.1, -0.05, 0.2))
    T2w = scenario.rigid()
    T12 = scenario.rigid(-0.015, 0.02, 0.01, (0.05, 0.02, -0.03)).astype(np.float64)
    R12, t12 = T12[:3, :3], T12[:3, 3]
    A, B = T1w.astype(np.float64), T2w.astype(np.float64)
    x2_des = _cam_points(ks[0], fr[0][2], shift, st, rng)                      # where key frame 1's points must land
    x1 = s12 * x2_des @ R12.T + t12
    P1 = (x1 - A[:3, 3]) @ A[:3, :3]                                           # world = R1w^T (x1 - t1w)
    x1_des = _cam_points(ks[1], fr[1][2], (-shift[0], -shift[1]), st, rng)
    x2 = (x1_des - t12) @ R12 / s12                                            # R12^T (x1 - t12) / s
    P2 = (x2 - B[:3, 3]) @ B[:3, :3]

    def pts(P, cam, k, d):
        n = len(k)
        max_d = (np.linalg.norm(cam, axis=1) * sf[k["octave"]] * 0.9995).astype(np.float32)  # see _pts
        return {"bad": (rng.random(n) < 0.15).astype(np.uint8), "world_pos": P.astype(np.float32),
                "normal": np.zeros((n, 3), np.float32), "min_dist": (max_d / sf[-1]).astype(np.float32),
                "max_dist": max_d, "desc": d}
    pts1, pts2 = pts(P1, x2_des, ks[0], ds[0]), pts(P2, x1_des, ks[1], ds[1])
    a1 = (rng.random(len(ks[0])) < pre).astype(np.uint8)
    a2 = (rng.random(len(ks[1])) < pre).astype(np.uint8)
    fx, fy, cx, cy = (float(v) for v in (st.fx, st.fy, st.cx, st.cy))
    args = (T1w, T2w, float(s12), R12.astype(np.float32), t12.astype(np.float32), fx, fy, cx, cy, log_sf, log_sf)
    ng, mg = gpu.search_by_sim3(g[0], g[1], *args, pts1, a1, pts2, a2, th)
    no, mo = oracle.search_by_sim3(o[0], o[1], *args, pts1, a1, pts2, a2, th)
    assert no > 100, no
    assert ng == no and np.array_equal(mg, mo), "%d vs %d, %d differ" % (ng, no, int((mg != mo).sum()))


@pytest.mark.parametrize("only_stereo,check_ori,levelsup", [(False, True, 2), (True, True, 3), (False, False, 1)])
def test_search_for_triangulation(gpu, oracle, only_stereo, check_ori, levelsup):
    """LocalMapping::CreateNewMapPoints (LocalMapping.cc:268): unmatched key points of two key frames under the same
    vocabulary node, Hamming <= TH_LOW, epipole and epipolar-line tests.  The synthetic stream is a pure image shift:
    F12 = the skew matrix of the shift direction (epipole at infinity)."""
    st, ge, fr, ks, ds, g, o, shift = _two_keyframes(gpu, oracle, 50, 52)
    rng = np.random.default_rng(levelsup)
    v = scenario.synthetic_vocabulary(10, 4, 31)
    gv = gpu.ORBVocabulary(10, 4, v["parent"], v["is_leaf"], v["desc"], v["weight"])
    ov = oracle.Vocabulary(10, 4, v["parent"], v["is_leaf"], v["desc"], v["weight"])
    tg = [gv.transform(d, levelsup) for d in ds]
    to = [ov.transform(d, levelsup) for d in ds]
    e = np.array([shift[0], shift[1], 0.0]) / max(np.hypot(*shift), 1e-9)
    F12 = np.array([[0, -e[2], e[1]], [e[2], 0, -e[0]], [-e[1], e[0], 0]], np.float32)
    ex, ey = float(320 + 1e6 * e[0]), float(240 + 1e6 * e[1])
    sig2 = ge.GetScaleSigmaSquares()
    h1 = (rng.random(len(ks[0])) < 0.4).astype(np.uint8)
    h2 = (rng.random(len(ks[1])) < 0.4).astype(np.uint8)
    ng, mg = gpu.search_for_triangulation(g[0], h1, tg[0]["node_id"], g[1], h2, tg[1]["node_id"], F12, ex, ey, sig2,
                                          only_stereo, check_ori)
    no, mo = oracle.search_for_triangulation(o[0], h1, to[0], o[1], h2, to[1], F12, ex, ey, sig2, only_stereo, check_ori)
    assert no > 30, no
    assert ng == no and np.array_equal(mg, mo), "%d vs %d, %d differ" % (ng, no, int((mg != mo).sum()))
    # a near epipole switches the mono-mono distance test on: same answer on both sides
    ng2, mg2 = gpu.search_for_triangulation(g[0], h1, tg[0]["node_id"], g[1], h2, tg[1]["node_id"], F12, 300.0, 200.0,
                                            sig2, only_stereo, check_ori)
    no2, mo2 = oracle.search_for_triangulation(o[0], h1, to[0], o[1], h2, to[1], F12, 300.0, 200.0, sig2, only_stereo,
                                               check_ori)
    assert ng2 == no2 and np.array_equal(mg2, mo2)
    gv.close()


@pytest.mark.parametrize("ratio,check_ori,levelsup", [(0.75, True, 2), (0.9, False, 3)])
def test_search_by_bow_keyframes(gpu, oracle, ratio, check_ori, levelsup):
    """LoopClosing::ComputeSim3 (LoopClosing.cc:266): key frame to key frame, both sides restricted to key points
    with a good map point, strict `< TH_LOW`."""
    st, ge, fr, ks, ds, g, o, shift = _two_keyframes(gpu, oracle, 60, 61)
    rng = np.random.default_rng(levelsup)
    v = scenario.synthetic_vocabulary(10, 4, 8)
    gv = gpu.ORBVocabulary(10, 4, v["parent"], v["is_leaf"], v["desc"], v["weight"])
    ov = oracle.Vocabulary(10, 4, v["parent"], v["is_leaf"], v["desc"], v["weight"])
    tg = [gv.transform(d, levelsup) for d in ds]
    to = [ov.transform(d, levelsup) for d in ds]
    v1 = (rng.random(len(ks[0])) < 0.8).astype(np.uint8)
    v2 = (rng.random(len(ks[1])) < 0.8).astype(np.uint8)
    ng, mg = gpu.search_by_bow_keyframes(ds[0], ks[0]["angle"], v1, tg[0]["node_id"], ds[1], ks[1]["angle"], v2,
                                         tg[1]["node_id"], ratio, check_ori)
    no, mo = oracle.search_by_bow_keyframes(ds[0], ks[0]["angle"], v1, to[0], ds[1], ks[1]["angle"], v2, to[1], ratio,
                                            check_ori)
    assert no > 20, no
    assert ng == no and np.array_equal(mg, mo), "%d vs %d, %d differ" % (ng, no, int((mg != mo).sum()))
    assert np.all(v2[mo[mo >= 0]] == 1) and np.all(v1[np.nonzero(mo >= 0)[0]] == 1)
    gv.close()


@pytest.mark.parametrize("window,ratio,check_ori,round2", [(100, 0.9, True, False), (40, 0.8, False, False),
                                                           (100, 0.9, True, True)])
def test_search_for_initialization(gpu, oracle, window, ratio, check_ori, round2):
    """Tracking::MonocularInitialization (Tracking.cc:877): level-0 key points of the initial frame against the
    current frame in a 100-px window around the previously matched position; later rows steal a key point when
    strictly closer.  round2: a second call with the updated vbPrevMatched, as the tracker does frame after frame."""
    from orb_slam2_map_amd.synth import Stream
    st = Stream(640, 480, 1234)
    ge = gpu.ORBextractor(2000, max_batch=3)  # the initialisation extractor uses 2 x nFeatures (Tracking.cc:211)
    fr = [st.frame(70), st.frame(71), st.frame(72)]
    ks, ds = ge.extract_batch(np.stack([f[0] for f in fr]))
    sf = ge.GetScaleFactors()
    g = [scenario.make_frame(gpu, ks[i], ds[i], fr[i][2], st, sf) for i in range(3)]
    o = [scenario.make_frame(oracle, ks[i], ds[i], fr[i][2], st, sf) for i in range(3)]
    pm0 = np.stack([ks[0]["x"], ks[0]["y"]], 1).astype(np.float32)  # mvbPrevMatched starts at the key points themselves
    ng, mg, pg = gpu.search_for_initialization(g[0], g[1], pm0, window, ratio, check_ori)
    no, mo, po = oracle.search_for_initialization(o[0], o[1], pm0, window, ratio, check_ori)
    assert no > 100, no
    assert ng == no and np.array_equal(mg, mo) and np.array_equal(pg, po), "%d vs %d" % (ng, no)
    assert np.all(ks[0]["octave"][mo >= 0] == 0) and np.all(ks[1]["octave"][mo[mo >= 0]] == 0)
    if round2:
        ng2, mg2, pg2 = gpu.search_for_initialization(g[0], g[2], pg, window, ratio, check_ori)
        no2, mo2, po2 = oracle.search_for_initialization(o[0], o[2], po, window, ratio, check_ori)
        assert ng2 == no2 and np.array_equal(mg2, mo2) and np.array_equal(pg2, po2) and no2 > 100
