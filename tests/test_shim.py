"""C++ shim (orb_slam2_map_amd/shim/orbgpu_shim.hpp): the reference's three class interfaces on top
of the C ABI.  CPU part: it compiles and links with g++.  GPU part: extractor -> SearchByProjection ->
PointCloudMapping through the shim equal the oracle."""
import os
import struct
import subprocess
import tempfile

import numpy as np
import pytest

import scenario

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "orb_slam2_map_amd")


def build_exe(tmpdir):
    import __graft_entry__ as ge
    if not os.path.exists(os.path.join(PKG, "liborbgpu.so")):
        ge.build()
    exe = os.path.join(tmpdir, "shim_test")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(PKG, "shim"),
           os.path.join(ROOT, "tests", "shim_test.cpp"), "-o", exe, "-L" + PKG, "-lorbgpu", "-Wl,-rpath," + PKG,
           "-Wl,-rpath,/opt/rocm/lib", "-pthread"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    return exe


def test_shim_compiles_and_links():
    with tempfile.TemporaryDirectory() as d:
        exe = build_exe(d)
        r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        assert r.returncode == 2 and "usage" in r.stderr


def check_table_section(oracle, buf, off, n, of, mp, wp, ko, no, k_last, n_last, Tcw, st):
    """The shim's MapPoint-table flavours of SearchByProjection(F, vpMapPoints, th), SearchLocalPoints and
    SearchByProjection(Cur, Last) against the oracle (the stand-in map point i carries id i)."""
    m = len(wp)
    nm_tab = struct.unpack_from("<i", buf, off)[0]
    ids_tab = np.frombuffer(buf, np.int32, n, off + 4)
    off += 4 + 4 * n
    assert nm_tab == no and np.array_equal(ids_tab, ko), "table flavour (mTrack* from the host)"
    nm_dev = struct.unpack_from("<i", buf, off)[0]
    ids_dev = np.frombuffer(buf, np.int32, n, off + 4)
    off += 4 + 4 * n
    assert nm_dev == no and np.array_equal(ids_dev, ko), "table flavour (isInFrustum on the device)"
    nseen = struct.unpack_from("<i", buf, off)[0]
    seen = np.frombuffer(buf, np.uint8, m, off + 4)
    off += 4 + m
    assert np.array_equal(seen, (mp["bad"] == 0).astype(np.uint8)) and nseen == int((mp["bad"] == 0).sum())
    last = {"has_mp": (np.arange(n_last) % 5 != 4).astype(np.uint8), "outlier": (np.arange(n_last) % 17 == 0).astype(np.uint8),
            "obs_pos": mp["obs_pos"][:n_last], "world_pos": wp[:n_last], "desc": mp["desc"][:n_last],
            "kp_octave": k_last["octave"], "kp_angle": k_last["angle"], "Tcw": Tcw}
    fx, fy, cx, cy, bf = (float(v) for v in (st.fx, st.fy, st.cx, st.cy, st.bf))
    nlo, klo = oracle.search_by_projection_last(of, Tcw, fx, fy, cx, cy, bf, bf / fx, last, 15.0, False, True,
                                                np.full(n, -1, np.int32))
    assert nlo > 100
    for what in ("host pointers", "table"):
        nl = struct.unpack_from("<i", buf, off)[0]
        idl = np.frombuffer(buf, np.int32, n, off + 4)
        off += 4 + 4 * n
        assert nl == nlo and np.array_equal(idl, klo), "SearchByProjection(Cur, Last), %s: %d vs %d" % (what, nl, nlo)
    return off


@pytest.mark.gpu
def test_shim_projection_c3_size(gpu, oracle, stream1280):
    """C3's size through the C++ shim (1280x960, 2000 features, ~10 k local map points): the table flavours equal the
    oracle, and the run prints the host wall time per call of every flavour (the figure INTEGRATION.md quotes)."""
    st = stream1280
    rng = np.random.default_rng(12)
    t_cur = 12
    g, rgb, depth = st.frame(t_cur)
    oe = oracle.Extractor(2000)
    ok, od = oe.extract(g)
    sf = oe.scale_factors()
    Tcw = scenario.rigid()
    wp, dsc, octv, k_last = [], [], [], None
    ox, oy = st.offset(t_cur)
    for t in (11, 10, 9, 8, 7):
        gp, _, dp = st.frame(t)
        k, d = oracle.Extractor(2000).extract(gp)
        px, py = st.offset(t)
        P, _ = scenario.world_points_from_prev(k, dp, (ox - px, oy - py), st, Tcw, rng)
        wp.append(P), dsc.append(d), octv.append(k["octave"])
        k_last = k if k_last is None else k_last
    n_first = len(wp[0])
    wp, dsc, octv = np.concatenate(wp), np.concatenate(dsc), np.concatenate(octv)
    mp = scenario.local_map(oracle, st, Tcw, wp, dsc, octv, sf, rng, obs_zero_frac=0.1)
    mp["bad"] = (mp["bad"] | (mp["in_view"] == 0) | (octv == len(sf) - 1)).astype(np.uint8)
    th = 3.0
    with tempfile.TemporaryDirectory() as d:
        exe = build_exe(d)
        scen, outp = os.path.join(d, "scen.bin"), os.path.join(d, "out.bin")
        with open(scen, "wb") as f:
            f.write(struct.pack("<4i", 1280, 960, 2000, len(wp)))
            f.write(struct.pack("<7f", float(st.fx), float(st.fy), float(st.cx), float(st.cy), float(st.bf), th,
                                float(np.log(np.float32(sf[1])))))  # mfLogScaleFactor as scenario.local_map used it
            f.write(g.tobytes()), f.write(rgb.tobytes()), f.write(depth.tobytes()), f.write(Tcw.astype("<f4").tobytes())
            rec = np.zeros(len(wp), np.dtype([("f", "u1", 3), ("lvl", "<i4"), ("s", "<f4", 4), ("wp", "<f4", 3), ("nr", "<f4", 3),
                                              ("d", "<f4", 2), ("desc", "u1", 32)]))
            rec["f"] = np.stack([mp["in_view"], mp["bad"], mp["obs_pos"]], 1)
            rec["lvl"] = mp["level"]
            rec["s"] = np.stack([mp["view_cos"], mp["proj_x"], mp["proj_y"], mp["proj_xr"]], 1)
            rec["wp"], rec["nr"] = wp, mp["normal"]
            rec["d"] = np.stack([mp["min_dist"], mp["max_dist"]], 1)
            rec["desc"] = mp["desc"]
            assert rec.dtype.itemsize == 3 + 4 + 16 + 12 + 12 + 8 + 32
            f.write(rec.tobytes())
            f.write(struct.pack("<i", n_first))
            lrec = np.zeros(n_first, np.dtype([("xy", "<f4", 2), ("o", "<i4"), ("a", "<f4")]))
            lrec["xy"] = np.stack([k_last["x"], k_last["y"]], 1)
            lrec["o"], lrec["a"] = k_last["octave"], k_last["angle"]
            f.write(lrec.tobytes())
        r = subprocess.run([exe, scen, outp, "--projection-only"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 0, r.stdout
        print(r.stdout)
        buf = open(outp, "rb").read()
    assert "shim timing" in r.stdout
    n = struct.unpack_from("<i", buf, 0)[0]
    off = 4
    kps = np.frombuffer(buf, oracle.KEYPOINT_DTYPE, n, off)
    off += 28 * n
    desc = np.frombuffer(buf, np.uint8, 32 * n, off).reshape(n, 32)
    off += 32 * n
    assert n == len(ok) and kps.tobytes() == ok.tobytes() and np.array_equal(desc, od)
    nm = struct.unpack_from("<i", buf, off)[0]
    k2m = np.frombuffer(buf, np.int32, n, off + 4)
    off += 4 + 4 * n
    of = scenario.make_frame(oracle, ok, od, depth, st, sf)
    no, ko = oracle.search_by_projection(of, mp, th, 0.8, np.full(n, -1, np.int32))
    assert len(wp) > 9000 and nm == no and np.array_equal(k2m, ko) and no > 500
    off += 4  # DescriptorDistance
    off = check_table_section(oracle, buf, off, n, of, mp, wp, ko, no, k_last, n_first, Tcw, st)
    assert off == len(buf)


@pytest.mark.gpu
def test_shim_end_to_end(gpu, oracle, stream640):
    st = stream640
    rng = np.random.default_rng(11)
    t_cur = 12
    g, rgb, depth = st.frame(t_cur)
    oe = oracle.Extractor(1000)
    ok, od = oe.extract(g)
    sf = oe.scale_factors()
    Tcw = scenario.rigid()
    # local map from two earlier frames (oracle extraction; the shim extracts the current frame itself)
    wp, dsc, octv, k_last = [], [], [], None
    ox, oy = st.offset(t_cur)
    for t in (11, 10):
        gp, _, dp = st.frame(t)
        k, d = oracle.Extractor(1000).extract(gp)
        px, py = st.offset(t)
        P, _ = scenario.world_points_from_prev(k, dp, (ox - px, oy - py), st, Tcw, rng)
        wp.append(P), dsc.append(d), octv.append(k["octave"])
        k_last = k if k_last is None else k_last
    n_first = len(wp[0])  # the map points made from frame t_cur - 1
    wp, dsc, octv = np.concatenate(wp), np.concatenate(dsc), np.concatenate(octv)
    mp = scenario.local_map(oracle, st, Tcw, wp, dsc, octv, sf, rng, obs_zero_frac=0.1)
    # points outside the frustum (or whose predicted level leaves the pyramid: an error for the Sim3 matcher, which the
    # reference leaves undefined) are flagged bad; SearchByProjection(F, ...) skips them either way
    # (points created on the top level predict level 7 or 8 depending on the last ulp of the camera distance)
    mp["bad"] = (mp["bad"] | (mp["in_view"] == 0) | (octv == len(sf) - 1)).astype(np.uint8)
    th = 3.0
    with tempfile.TemporaryDirectory() as d:
        exe = build_exe(d)
        scen, outp = os.path.join(d, "scen.bin"), os.path.join(d, "out.bin")
        with open(scen, "wb") as f:
            f.write(struct.pack("<4i", 640, 480, 1000, len(wp)))
            f.write(struct.pack("<7f", float(st.fx), float(st.fy), float(st.cx), float(st.cy), float(st.bf), th,
                                float(np.log(np.float32(sf[1])))))  # mfLogScaleFactor as scenario.local_map used it
            f.write(g.tobytes()), f.write(rgb.tobytes()), f.write(depth.tobytes()), f.write(Tcw.astype("<f4").tobytes())
            for i in range(len(wp)):
                f.write(struct.pack("<3B", int(mp["in_view"][i]), int(mp["bad"][i]), int(mp["obs_pos"][i])))
                f.write(struct.pack("<i", int(mp["level"][i])))
                f.write(struct.pack("<4f", float(mp["view_cos"][i]), float(mp["proj_x"][i]), float(mp["proj_y"][i]),
                                    float(mp["proj_xr"][i])))
                f.write(wp[i].astype("<f4").tobytes())
                f.write(mp["normal"][i].astype("<f4").tobytes())
                f.write(struct.pack("<2f", float(mp["min_dist"][i]), float(mp["max_dist"][i])))
                f.write(mp["desc"][i].tobytes())
            # the last frame of SearchByProjection(Cur, Last): the key points the first n_first map points were made from
            f.write(struct.pack("<i", n_first))
            for i in range(n_first):
                f.write(struct.pack("<2fif", float(k_last["x"][i]), float(k_last["y"][i]), int(k_last["octave"][i]),
                                    float(k_last["angle"][i])))
            # vocabulary for the SearchByBoW part + how many map points play the key frame's features
            voc = scenario.synthetic_vocabulary(10, 3, 99)
            n_kf = n_first
            f.write(struct.pack("<4i", voc["k"], voc["L"], len(voc["parent"]), n_kf))
            f.write(voc["parent"].astype("<i4").tobytes()), f.write(voc["is_leaf"].tobytes())
            f.write(voc["desc"].tobytes()), f.write(voc["weight"].astype("<f8").tobytes())
        r = subprocess.run([exe, scen, outp], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 0, r.stdout
        print(r.stdout)
        buf = open(outp, "rb").read()
    n = struct.unpack_from("<i", buf, 0)[0]
    off = 4
    kps = np.frombuffer(buf, oracle.KEYPOINT_DTYPE, n, off)
    off += 28 * n
    desc = np.frombuffer(buf, np.uint8, 32 * n, off).reshape(n, 32)
    off += 32 * n
    assert n == len(ok) and kps.tobytes() == ok.tobytes() and np.array_equal(desc, od)
    nm = struct.unpack_from("<i", buf, off)[0]
    off += 4
    k2m = np.frombuffer(buf, np.int32, n, off)
    off += 4 * n
    of = scenario.make_frame(oracle, ok, od, depth, st, sf)
    no, ko = oracle.search_by_projection(of, mp, th, 0.8, np.full(n, -1, np.int32))
    assert nm == no and np.array_equal(k2m, ko) and no > 50
    dist = struct.unpack_from("<i", buf, off)[0]
    off += 4
    assert dist == oracle.descriptor_distance(mp["desc"][0], mp["desc"][-1])
    off = check_table_section(oracle, buf, off, n, of, mp, wp, ko, no, k_last, n_first, Tcw, st)
    # loop-closing projection through the shim (Sim3 = 1.5 x the rigid pose) and distinctive descriptors
    ns = struct.unpack_from("<i", buf, off)[0]
    off += 4
    matched = np.frombuffer(buf, np.int32, n, off)
    off += 4 * n
    Scw = Tcw.astype(np.float32).copy()
    Scw[:3, :] *= np.float32(1.5)
    pts = {"bad": mp["bad"], "world_pos": wp, "normal": mp["normal"], "min_dist": mp["min_dist"],
           "max_dist": mp["max_dist"], "desc": mp["desc"]}
    log_sf = float(np.float32(np.log(np.float64(np.float32(sf[1])))))
    nso, kso = oracle.search_by_projection_sim3(of, Scw, float(st.fx), float(st.fy), float(st.cx), float(st.cy),
                                                log_sf, pts, 10, np.full(n, -1, np.int32))
    assert ns == nso and np.array_equal(matched, kso) and nso > 50
    best = np.frombuffer(buf, np.int32, 3, off)
    off += 12
    assert list(best) == [oracle.distinctive_descriptor(mp["desc"][0::2]), oracle.distinctive_descriptor(mp["desc"][1::2]), -1]
    # vocabulary transform + SearchByBoW through the shim (levelsup 2, ratio 0.7, no orientation check)
    ov = oracle.Vocabulary(voc["k"], voc["L"], voc["parent"], voc["is_leaf"], voc["desc"], voc["weight"])
    tk, tf = ov.transform(mp["desc"][:n_kf], 2), ov.transform(od, 2)
    valid = (mp["bad"][:n_kf] == 0).astype(np.uint8)
    nbo, mbo = oracle.search_by_bow(mp["desc"][:n_kf], np.zeros(n_kf, np.float32), valid, tk, od, ok["angle"], tf, 50, 0.7,
                                    False)
    nb = struct.unpack_from("<i", buf, off)[0]
    off += 4
    got = np.frombuffer(buf, np.int32, n, off)
    off += 4 * n
    assert nb == nbo and np.array_equal(got, mbo) and nbo > 20, (nb, nbo)
    nbow = struct.unpack_from("<i", buf, off)[0]
    off += 4
    assert nbow == len(tf["bow_ids"])
    for t in range(nbow):
        wid, val = struct.unpack_from("<id", buf, off)
        off += 12
        assert wid == tf["bow_ids"][t] and val == tf["bow_vals"][t]

    # background matchers through the shim: SearchForTriangulation of the frame against itself, then Fuse
    nt, npairs = struct.unpack_from("<2i", buf, off)
    off += 8
    pairs = np.frombuffer(buf, np.int32, 2 * npairs, off).reshape(npairs, 2)
    off += 8 * npairs
    h1 = (np.arange(n) % 3 == 0).astype(np.uint8)
    h2 = (np.arange(n) % 3 == 1).astype(np.uint8)
    F12 = np.array([[0, -0.0, 0.6], [0, 0, -0.8], [-0.6, 0.8, 0]], np.float32)
    f32 = np.float32
    C2 = [f32(0.5) + f32(0.25), f32(-0.25) + f32(0.5), f32(0.125) + f32(2.0)]
    invz = f32(1.0) / C2[2]
    ex = float(f32(f32(f32(st.fx) * C2[0]) * invz) + f32(st.cx))
    ey = float(f32(f32(f32(st.fy) * C2[1]) * invz) + f32(st.cy))
    sig2 = oe.sigma2()
    nto, mto = oracle.search_for_triangulation(of, h1, tf, of, h2, tf, F12, ex, ey, sig2, False, True)
    want = np.array([(i, mto[i]) for i in range(n) if mto[i] >= 0], np.int32).reshape(-1, 2)
    assert nt == nto == npairs and np.array_equal(pairs, want) and nto > 100, (nt, nto, npairs)
    nfused = struct.unpack_from("<i", buf, off)[0]
    off += 4
    kf_ids = np.frombuffer(buf, np.int32, n, off)
    off += 4 * n
    status = np.frombuffer(buf, np.int32, 3 * len(wp), off).reshape(len(wp), 3)
    off += 12 * len(wp)
    # the same edits on the oracle's candidates (ORBmatcher.cc:946-969 over the one-key-frame stand-in graph)
    m = len(wp)
    idx_in_kf = np.full(m, -1, np.int64)
    ids = ko.astype(np.int64).copy()  # SearchByProjection's associations
    for j in range(n):
        if ids[j] >= 0:
            idx_in_kf[ids[j]] = j
    bad = mp["bad"].astype(bool).copy()
    nobs = (mp["obs_pos"] != 0).astype(np.int64)
    repl = np.full(m, -1, np.int64)
    fpts = dict(pts)
    fpts["bad"] = (bad | (idx_in_kf >= 0)).astype(np.uint8)
    inv_s2 = oe.inv_sigma2()
    _, best = oracle.fuse(of, Tcw, float(st.fx), float(st.fy), float(st.cx), float(st.cy), float(st.bf), log_sf, fpts, th,
                          inv_s2)

    def replace(x, y):
        if x == y:
            return
        if idx_in_kf[x] >= 0:
            if idx_in_kf[y] < 0:
                ids[idx_in_kf[x]] = y
                idx_in_kf[y] = idx_in_kf[x]
                nobs[y] += 1
            else:
                ids[idx_in_kf[x]] = -1
        idx_in_kf[x] = -1
        nobs[x] = 0
        bad[x] = True
        repl[x] = y

    want_fused = 0
    for i in range(m):
        if best[i] < 0 or bad[i] or idx_in_kf[i] >= 0:
            continue
        j = best[i]
        pin = ids[j]
        if pin >= 0:
            if not bad[pin]:
                if nobs[pin] > nobs[i]:
                    replace(i, pin)
                else:
                    replace(pin, i)
        else:
            idx_in_kf[i] = j
            nobs[i] += 1
            ids[j] = i
        want_fused += 1
    assert nfused == want_fused and want_fused > 100, (nfused, want_fused)
    assert np.array_equal(kf_ids, ids)
    assert np.array_equal(status[:, 0], bad.astype(np.int32)) and np.array_equal(status[:, 1], nobs)
    assert np.array_equal(status[:, 2], repl)
    assert (repl >= 0).sum() > 0 and (ids != ko).sum() > 0  # both kinds of edit happened

    # point cloud thread: three key frames one per pass, a loop closure (poses moved, one key frame culled, key frames
    # taken in id order = reversed), the shutdown pass
    camv = (float(st.fx), float(st.fy), float(st.cx), float(st.cy))

    def kf_cloud(T):
        R, t = oracle.pose_inverse(T)
        return oracle.transform_points(oracle.backproject(depth, rgb, *camv), R, t)

    def read_cloud(off):
        nc = struct.unpack_from("<q", buf, off)[0]
        return np.frombuffer(buf, oracle.POINT_DTYPE, nc, off + 8), off + 8 + 16 * nc
    poses = []
    for i in range(3):
        T = Tcw.astype(np.float32).copy()
        T[0, 3] = np.float32(T[0, 3] + np.float32(0.25) * np.float32(i))
        poses.append(T)
    omap = np.zeros(0, oracle.POINT_DTYPE)
    for T in poses:
        omap, _ = oracle.voxel_filter(np.concatenate([omap, kf_cloud(T)]), 0.05)
    cloud_a, off = read_cloud(off)
    assert cloud_a.tobytes() == omap.tobytes(), "three inserts through the viewer thread"
    for T in poses:
        T[1, 3] = np.float32(T[1, 3] - np.float32(0.125))
    oreb, _ = oracle.voxel_filter(np.concatenate([kf_cloud(poses[2]), kf_cloud(poses[0])]), 0.05)  # ids 8, (9 bad), 10
    cloud_b, off = read_cloud(off)
    assert cloud_b.tobytes() == oreb.tobytes(), "loop-closure rebuild: non-bad key frames in id order"
    ocat = np.concatenate([oracle.voxel_filter(kf_cloud(T), 0.05)[0] for T in poses])
    ofin, _ = oracle.statistical_outlier_removal(ocat, 50, 1.0)
    assert 0 < len(ofin) < len(ocat)
    cloud_c, off = read_cloud(off)
    assert cloud_c.tobytes() == ofin.tobytes(), "shutdown pass: per-key-frame filtered clouds, concatenated, sor.filter"

    # the reference's insert semantics (default) and the corrected opt-in, composed from the oracle's pieces:
    # PointCloudMap.cc:244-262 inserts keyFrameCloud.back() with keyframes[lastKeyframeSize]'s pose and leaves
    # lastKeyframeSize alone in the loop branch (:217-243)
    def kf_cloud_of(i, T):
        R, t = oracle.pose_inverse(T)
        d_i = (depth + np.float32(0.125) * np.float32(i)).astype(np.float32)
        return oracle.transform_points(oracle.backproject(d_i, rgb, *camv), R, t)

    def add(omap, cloud):
        return oracle.voxel_filter(np.concatenate([omap, cloud]), 0.05)[0]

    def qposes(moved):
        out = []
        for i in range(5):
            T = Tcw.astype(np.float32).copy()
            T[0, 3] = np.float32(T[0, 3] + np.float32(0.25) * np.float32(i))
            T[2, 3] = np.float32(T[2, 3] - np.float32(0.0625) * np.float32(i))
            if moved:
                T[1, 3] = np.float32(T[1, 3] - np.float32(0.125))
            out.append(T)
        return out
    P0, P1 = qposes(False), qposes(True)
    empty = np.zeros(0, oracle.POINT_DTYPE)
    for mode, what in ((0, "reference semantics (default)"), (1, "corrected insert (setReferenceQuirks(false))")):
        m1 = add(empty, kf_cloud_of(0, P0[0]))
        if mode == 0:
            m1 = add(m1, kf_cloud_of(2, P0[1]))  # last cloud, first pose: ONE insert for the two key frames
        else:
            m1 = add(add(m1, kf_cloud_of(1, P0[1])), kf_cloud_of(2, P0[2]))
        got, off = read_cloud(off)
        assert got.tobytes() == m1.tobytes(), "two key frames in one wake-up, " + what
        m2 = add(empty, np.concatenate([kf_cloud_of(i, P1[i]) for i in range(4)]))  # rebuild: ids ascending
        got, off = read_cloud(off)
        assert got.tobytes() == m2.tobytes(), "loop closure at a wake-up with a new key frame, " + what
        m3 = add(m2, kf_cloud_of(4, P1[3] if mode == 0 else P1[4]))  # stale lastKeyframeSize = 3 in the reference
        got, off = read_cloud(off)
        assert got.tobytes() == m3.tobytes(), "insert after the loop closure, " + what
        if mode == 0:
            ref_maps = (m1, m3)
        else:
            assert m1.tobytes() != ref_maps[0].tobytes() and m3.tobytes() != ref_maps[1].tobytes(), \
                "the two modes must be told apart by this scenario"
    assert off == len(buf)
