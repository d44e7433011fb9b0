"""Device-resident MapPoint table + device-resident frames (orbgpu_mappoint_table_*, orbgpu_frame_*,
orbgpu_search_local_points_table, orbgpu_search_by_projection_last_table) against the oracle's
SearchByProjection on the same scenario (reference: ORBmatcher.cc:45-137, :1328-1470, Tracking.cc:1447-1497,
MapPoint.cc:242-313)."""
import numpy as np
import pytest

import scenario

pytestmark = pytest.mark.gpu


def make_gframe(gpu, oframe):
    return gpu.Frame(oframe.kp_x, oframe.kp_y, oframe.octave, oframe.angle, oframe.u_right, oframe.desc,
                     float(oframe.max_x), float(oframe.max_y), oframe.scale_factors)


def test_table_rows_survive_growth_and_edits(gpu):
    rng = np.random.default_rng(5)
    t = gpu.MapPointTable(initial_rows=0)
    n = 5000  # beyond the first 1024-row allocation: rows and hash are carried over twice
    ids = rng.permutation(100000)[:n].astype(np.int64)
    wp = rng.normal(0, 3, (n, 3)).astype(np.float32)
    nr = rng.normal(0, 1, (n, 3)).astype(np.float32)
    mn, mx = rng.uniform(0.1, 1, n).astype(np.float32), rng.uniform(2, 9, n).astype(np.float32)
    ds = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    nobs = rng.integers(0, 4, n).astype(np.int32)
    for a in range(0, n, 700):
        sl = slice(a, a + 700)
        t.upsert(ids[sl], wp[sl], nr[sl], mn[sl], mx[sl], ds[sl], nobs[sl])
    assert t.rows() == n
    for i in (0, 699, 700, 1023, 1024, 4999):
        r = t.read(ids[i])
        assert np.array_equal(r["world_pos"], wp[i]) and np.array_equal(r["normal"], nr[i]) and np.array_equal(r["desc"], ds[i])
        assert r["min_dist"] == mn[i] and r["max_dist"] == mx[i] and r["has_observations"] == int(nobs[i] > 0) and r["bad"] == 0
    # partial update of known ids (SetWorldPos / ComputeDistinctiveDescriptors): the other attributes stay
    sel = np.array([3, 1500, 4000])
    t.upsert(ids[sel], world_pos=wp[sel] + 1)
    t.upsert(ids[sel[:2]], desc=255 - ds[sel[:2]])
    assert t.rows() == n
    r = t.read(ids[1500])
    assert np.array_equal(r["world_pos"], wp[1500] + 1) and np.array_equal(r["desc"], 255 - ds[1500])
    assert np.array_equal(r["normal"], nr[1500]) and r["max_dist"] == mx[1500]
    r = t.read(ids[4000])
    assert np.array_equal(r["world_pos"], wp[4000] + 1) and np.array_equal(r["desc"], ds[4000])
    # SetBadFlag / observation counts; unknown ids are ignored
    assert t.set_bad(np.array([ids[7], ids[8], 999999], np.int64)) == 2
    assert t.read(ids[7])["bad"] == 1 and t.read(ids[9])["bad"] == 0
    assert t.set_observations(ids[:2], [0, 5]) == 2
    assert t.read(ids[0])["has_observations"] == 0 and t.read(ids[1])["has_observations"] == 1
    # a new point without attributes: zeros, observed, not bad
    t.upsert(np.array([123456789012], np.int64))
    r = t.read(123456789012)
    assert t.rows() == n + 1 and not r["world_pos"].any() and r["has_observations"] == 1 and r["bad"] == 0
    # refused: an id twice in one call (and the table is as before), negative ids, unknown reads
    with pytest.raises(RuntimeError, match="twice"):
        t.upsert(np.array([555, 777, 555], np.int64))
    assert t.rows() == n + 1
    with pytest.raises(RuntimeError, match="not in the table"):
        t.read(555)
    t.upsert(np.array([555], np.int64), world_pos=np.ones((1, 3), np.float32))
    assert t.rows() == n + 2 and np.array_equal(t.read(555)["world_pos"], np.ones(3, np.float32))
    with pytest.raises(RuntimeError, match="negative"):
        t.upsert(np.array([-4], np.int64))
    # retain: the table is bounded by the live map -- only the listed ids stay (shuffled; an unknown id and a repeated one
    # are passed over), rows are renumbered, attributes and flags travel with their id, dropped ids are unknown afterwards
    keep = rng.permutation(n)[:1800]
    listed = np.concatenate([ids[keep], [987654321], ids[keep[:3]]]).astype(np.int64)
    assert t.retain(listed) == (n + 2) - 1800 and t.rows() == 1800
    for i in keep[[0, 1, 900, 1799]]:
        r = t.read(ids[i])
        want_wp = wp[i] + 1 if i in (3, 1500, 4000) else wp[i]
        assert np.array_equal(r["world_pos"], want_wp) and np.array_equal(r["normal"], nr[i]) and r["min_dist"] == mn[i]
        assert r["bad"] == int(i in (7, 8))
    gone = np.setdiff1d(np.arange(n), keep)[:5]
    for i in gone:
        with pytest.raises(RuntimeError, match="not in the table"):
            t.read(ids[i])
    assert t.set_bad(ids[gone]) == 0
    # the shrunk table grows again, a dropped id may come back as a new point
    back = ids[gone[:2]]
    t.upsert(np.concatenate([back, np.arange(2_000_000, 2_003_000)]).astype(np.int64),
             world_pos=np.full((3002, 3), 7, np.float32))
    assert t.rows() == 1800 + 3002 and np.array_equal(t.read(back[0])["world_pos"], np.full(3, 7, np.float32))
    assert np.array_equal(t.read(ids[keep[5]])["desc"], ds[keep[5]] if keep[5] not in (3, 1500) else 255 - ds[keep[5]])


def local_map_scenario(gpu, oracle, w, h, nfeat, nprev, obs_zero, seed):
    from orb_slam2_map_amd.synth import Stream
    rng = np.random.default_rng(seed)
    t_cur = 12
    st = Stream(w, h, 1234)
    ts = [t_cur - 1 - i for i in range(nprev)] + [t_cur]
    frames = [st.frame(t) for t in ts]
    ge = gpu.ORBextractor(nfeat, max_batch=nprev + 1)
    ks, ds = ge.extract_batch(np.stack([f[0] for f in frames]))
    sf = np.asarray(ge.GetScaleFactors(), np.float32)
    Tcw = scenario.rigid()
    ox, oy = st.offset(t_cur)
    wp, dsc, octv = [], [], []
    for i, t in enumerate(ts[:-1]):
        px, py = st.offset(t)
        P, _ = scenario.world_points_from_prev(ks[i], frames[i][2], (ox - px, oy - py), st, Tcw, rng)
        wp.append(P), dsc.append(ds[i]), octv.append(ks[i]["octave"])
    wp, dsc, octv = np.concatenate(wp), np.concatenate(dsc), np.concatenate(octv)
    wp[::37] = -wp[::37]
    mp = scenario.local_map(oracle, st, Tcw, wp, dsc, octv, sf, rng, obs_zero, vary=True)
    of = scenario.make_frame(oracle, ks[-1], ds[-1], frames[-1][2], st, sf)
    return st, sf, Tcw, wp, mp, of, rng


@pytest.mark.parametrize("w,h,nfeat,nprev,th,obs_zero", [(640, 480, 1000, 3, 3.0, 0.0), (640, 480, 1000, 2, 5.0, 0.3),
                                                           (1280, 960, 2000, 5, 3.0, 0.1)])
def test_search_local_points_over_the_table(gpu, oracle, w, h, nfeat, nprev, th, obs_zero):
    """Both flavours (mTrack* from the host / isInFrustum on the device) give the oracle's SearchByProjection; ids are
    arbitrary 64-bit numbers in a shuffled table that holds more points than the call lists; the frame's existing
    associations arrive as ids and come back as list positions.  The last case is C3's size (1280x960, 2000 features,
    ~10 k local map points)."""
    st, sf, Tcw, wp, mp, of, rng = local_map_scenario(gpu, oracle, w, h, nfeat, nprev, obs_zero, 77)
    m = len(wp)
    assert 0.4 * m < mp["in_view"].sum() < 0.95 * m
    # the table: the m list points + 500 others, inserted in shuffled order, ids sparse
    extra = 500
    ids_all = (rng.permutation(10 * (m + extra))[:m + extra].astype(np.int64) * 7919 + 11)
    ids = ids_all[:m]
    x_wp = rng.normal(0, 2, (extra, 3)).astype(np.float32)
    x_obs = (np.arange(extra) % 2).astype(np.int32)  # half of the outside points have no observations
    order = rng.permutation(m + extra)
    A = {"world_pos": np.concatenate([wp, x_wp]), "normal": np.concatenate([mp["normal"], np.zeros((extra, 3), np.float32)]),
         "min_dist": np.concatenate([mp["min_dist"], np.zeros(extra, np.float32)]),
         "max_dist": np.concatenate([mp["max_dist"], np.zeros(extra, np.float32)]),
         "desc": np.concatenate([mp["desc"], rng.integers(0, 256, (extra, 32), dtype=np.uint8)]),
         "n_obs": np.concatenate([mp["obs_pos"].astype(np.int32) * 3, x_obs])}
    tbl = gpu.MapPointTable(initial_rows=64)
    for a in range(0, m + extra, 3000):
        sel = order[a:a + 3000]
        tbl.upsert(ids_all[sel], A["world_pos"][sel], A["normal"][sel], A["min_dist"][sel], A["max_dist"][sel], A["desc"][sel],
                   A["n_obs"][sel])
    assert tbl.set_bad(ids[mp["bad"] != 0]) == int((mp["bad"] != 0).sum())
    # existing associations of the frame: to list points, to outside points with / without observations
    k0 = np.full(of.n, -1, np.int32)
    kp_ids = np.full(of.n, -1, np.int64)
    pre = rng.choice(of.n, 300, replace=False)
    k0[pre[:100]] = rng.integers(0, m, 100)
    kp_ids[pre[:100]] = ids[k0[pre[:100]]]
    k0[pre[100:200]] = -2
    kp_ids[pre[100:200]] = ids_all[m + 1 + 2 * np.arange(100)]      # odd extra rows: observed -> hold their key point
    kp_ids[pre[200:]] = ids_all[m + 2 * np.arange(100)]             # even extra rows: no observations -> do not block
    no, ko = oracle.search_by_projection(of, mp, th, 0.8, k0)
    assert no > 50
    want = ko.copy()
    fx, fy, cx, cy, bf = (float(v) for v in (st.fx, st.fy, st.cx, st.cy, st.bf))
    log_sf = float(np.log(np.float32(sf[1])))
    dfr = gpu.DeviceFrame().upload(make_gframe(gpu, of))
    # (a) drop-in at the ORBmatcher level: mTrack* members from the host
    n1, k1 = gpu.search_local_points_table(dfr, tbl, ids, None, fx, fy, cx, cy, bf, log_sf, th, 0.8, scratch=mp, kp_ids=kp_ids)
    assert n1 == no and np.array_equal(k1, want), "scratch mode: %d vs %d, %d differ" % (n1, no, int((k1 != want).sum()))
    # (b) Tracking::SearchLocalPoints on the device: only ids and the pose go up
    n2, k2, trk = gpu.search_local_points_table(dfr, tbl, ids, Tcw, fx, fy, cx, cy, bf, log_sf, th, 0.8, kp_ids=kp_ids,
                                                want_track=True)
    assert n2 == no and np.array_equal(k2, want), "device mode: %d vs %d, %d differ" % (n2, no, int((k2 != want).sum()))
    live = mp["bad"] == 0
    assert np.array_equal(trk["in_view"][live], mp["in_view"][live])
    sel = live & (mp["in_view"] != 0)
    for k in ("proj_x", "proj_y", "proj_xr", "view_cos", "level"):
        assert np.array_equal(trk[k][sel], mp[k][sel]), k
    # (c) the host-only skip (mnLastFrameSeen == F.mnId): rows skipped by the caller == rows flagged bad for the oracle
    skip = (rng.random(m) < 0.2).astype(np.uint8)
    mp2 = dict(mp)
    mp2["bad"] = (mp["bad"] | skip).astype(np.uint8)
    no3, ko3 = oracle.search_by_projection(of, mp2, th, 0.8, k0)
    n3, k3 = gpu.search_local_points_table(dfr, tbl, ids, Tcw, fx, fy, cx, cy, bf, log_sf, th, 0.8, skip=skip, kp_ids=kp_ids)
    assert n3 == no3 and np.array_equal(k3, ko3)
    # (d) no associations at all (kp_ids = None) and an id the table has never seen
    no4, ko4 = oracle.search_by_projection(of, mp, th, 0.8, np.full(of.n, -1, np.int32))
    n4, k4 = gpu.search_local_points_table(dfr, tbl, ids, Tcw, fx, fy, cx, cy, bf, log_sf, th, 0.8)
    assert n4 == no4 and np.array_equal(k4, ko4)
    # (e) ids the table has not been told about yet (LocalMapping publishes a point before its attributes are final,
    #     LocalMapping.cc:434-440): unknown list rows are skipped rows, unknown key-point ids count as held
    unk = rng.choice(m, 40, replace=False)
    late_ids = ids.copy()
    late_ids[unk] = 4242424242 + np.arange(40)
    mp5 = dict(mp)
    mp5["bad"] = mp["bad"].copy()
    mp5["bad"][unk] = 1
    k05 = np.full(of.n, -1, np.int32)
    kp5 = np.full(of.n, -1, np.int64)
    held = rng.choice(of.n, 25, replace=False)
    k05[held] = -2
    kp5[held] = 777000000 + np.arange(25)  # associations to points the table does not know: treated as held
    no5, ko5 = oracle.search_by_projection(of, mp5, th, 0.8, k05)
    n5b, k5b = gpu.search_local_points_table(dfr, tbl, late_ids, Tcw, fx, fy, cx, cy, bf, log_sf, th, 0.8, kp_ids=kp5)
    assert n5b == no5 and np.array_equal(k5b, ko5)
    assert tbl.last_unknown() == (40, 25)
    gpu.search_local_points_table(dfr, tbl, ids, Tcw, fx, fy, cx, cy, bf, log_sf, th, 0.8)
    assert tbl.last_unknown() == (0, 0)
    # (f) an empty list: nothing is searched, the associations are still translated through the table -- a key point
    #     holding an observed outside point is -2, one holding an unobserved point is -1 (as with a non-empty list)
    n6, k6 = gpu.search_local_points_table(dfr, tbl, np.zeros(0, np.int64), Tcw, fx, fy, cx, cy, bf, log_sf, th, 0.8, kp_ids=kp_ids)
    want6 = np.full(of.n, -1, np.int32)
    want6[pre[:100]] = np.where(mp["obs_pos"][k0[pre[:100]]] != 0, -2, -1)  # list points are outside an empty list
    want6[pre[100:200]] = -2
    assert n6 == 0 and np.array_equal(k6, want6)
    # (g) after a retain() of exactly the listed + held points (shuffled order -> every row number changes) the search
    #     gives the same result
    live = np.concatenate([ids, ids_all[m:]])
    assert tbl.retain(rng.permutation(live)) == 0
    n7, k7 = gpu.search_local_points_table(dfr, tbl, ids, Tcw, fx, fy, cx, cy, bf, log_sf, th, 0.8, kp_ids=kp_ids)
    assert n7 == no and np.array_equal(k7, want)
    # the table path and the host-pointer entry point agree as well
    n5, k5 = gpu.ORBmatcher(0.8, True).SearchByProjection(make_gframe(gpu, of), mp, th, k0)
    assert n5 == no and np.array_equal(k5, want)


@pytest.mark.parametrize("th,mono,obs_zero,motion", [(15.0, False, 0.0, "none"), (7.0, False, 0.4, "forward"),
                                                      (15.0, True, 0.2, "backward")])
def test_search_by_projection_last_frame_over_the_table(gpu, oracle, th, mono, obs_zero, motion):
    """TrackWithMotionModel's matcher with both frames device-resident and the last frame's map points looked up by id."""
    from orb_slam2_map_amd.synth import Stream
    rng = np.random.default_rng(int(th) + 5 * int(mono) + len(motion))
    w, h = 640, 480
    st = Stream(w, h, 1234)
    fr = [st.frame(30), st.frame(31)]
    ge = gpu.ORBextractor(1000, max_batch=2)
    ks, ds = ge.extract_batch(np.stack([f[0] for f in fr]))
    sf = np.asarray(ge.GetScaleFactors(), np.float32)
    Tcw = scenario.rigid()
    olast = scenario.make_frame(oracle, ks[0], ds[0], fr[0][2], st, sf)
    ocur = scenario.make_frame(oracle, ks[1], ds[1], fr[1][2], st, sf)
    n0, n1 = olast.n, ocur.n
    (px, py), (ox, oy) = st.offset(30), st.offset(31)
    P, _ = scenario.world_points_from_prev(ks[0], fr[0][2], (ox - px, oy - py), st, Tcw, rng)
    Tlast = Tcw.copy()
    if motion == "forward":
        Tlast[2, 3] += 0.5
    elif motion == "backward":
        Tlast[2, 3] -= 0.5
    mp_desc = rng.integers(0, 256, (n0, 32), dtype=np.uint8)
    near = rng.random(n0) < 0.9  # most map points carry (nearly) the key point's descriptor, as a tracked map does
    mp_desc[near] = ds[0][near]
    last = {"has_mp": (rng.random(n0) < 0.8).astype(np.uint8), "outlier": (rng.random(n0) < 0.05).astype(np.uint8),
            "obs_pos": (rng.random(n0) >= obs_zero).astype(np.uint8), "world_pos": P, "desc": mp_desc,
            "kp_octave": ks[0]["octave"], "kp_angle": ks[0]["angle"], "Tcw": Tlast}
    fx, fy, cx, cy, bf = (float(v) for v in (st.fx, st.fy, st.cx, st.cy, st.bf))
    mb = bf / fx
    k0 = np.full(n1, -1, np.int32)
    pre = rng.choice(n1, 60, replace=False)
    k0[pre[:30]] = -2
    no, ko = oracle.search_by_projection_last(ocur, Tcw, fx, fy, cx, cy, bf, mb, last, th, mono, True, k0)
    assert no > 100, no
    # table: one point per last-frame key point that has one (+ two outside points the current frame already holds)
    ids = rng.permutation(50000)[:n0 + 2].astype(np.int64) + 17
    has = last["has_mp"] != 0
    tbl = gpu.MapPointTable()
    tbl.upsert(ids[:n0][has], world_pos=P[has], desc=mp_desc[has], n_obs=last["obs_pos"][has].astype(np.int32))
    tbl.upsert(ids[n0:], n_obs=np.array([2, 0], np.int32))
    last_ids = np.where(has, ids[:n0], -1)
    cur_ids = np.full(n1, -1, np.int64)
    cur_ids[pre[:30]] = ids[n0]        # held by an observed outside point (-2 for the oracle)
    cur_ids[pre[30:]] = ids[n0 + 1]    # held by a point without observations: does not block (-1 for the oracle)
    dl = gpu.DeviceFrame().upload(make_gframe(gpu, olast))
    dc = gpu.DeviceFrame().upload(make_gframe(gpu, ocur))
    ng, kg = gpu.search_by_projection_last_table(dc, Tcw, dl, Tlast, tbl, last_ids, fx, fy, cx, cy, bf, mb, th, mono, True,
                                                 last_outlier=last["outlier"], cur_kp_ids=cur_ids)
    assert ng == no and np.array_equal(kg, ko), "%d vs %d, %d differ" % (ng, no, int((kg != ko).sum()))
    # the current frame becomes the next call's last frame without another upload: same handle, roles swapped
    ng2, kg2 = gpu.search_by_projection_last_table(dl, Tlast, dc, Tcw, tbl, np.full(n1, -1, np.int64), fx, fy, cx, cy, bf, mb, th,
                                                   mono, True)
    assert ng2 == 0 and np.all(kg2 == -1)
