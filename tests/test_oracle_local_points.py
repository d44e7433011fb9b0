"""oracle.search_local_points (the C composition bench.py's C3 CPU baseline times: Tracking::SearchLocalPoints,
Tracking.cc:1447-1497) against the point-by-point Python composition the GPU parity tests use (scenario.local_map =
Frame::isInFrustum per point, then ORBmatcher::SearchByProjection)."""
import numpy as np

import scenario


def test_search_local_points_equals_the_python_composition(oracle, stream640):
    st = stream640
    rng = np.random.default_rng(21)
    t_cur = 9
    g, _, depth = st.frame(t_cur)
    oe = oracle.Extractor(1000)
    ok, od = oe.extract(g)
    sf = oe.scale_factors()
    Tcw = scenario.rigid()
    ox, oy = st.offset(t_cur)
    wp, dsc, octv = [], [], []
    for t in (8, 7):
        gp, _, dp = st.frame(t)
        k, d = oracle.Extractor(1000).extract(gp)
        px, py = st.offset(t)
        P, _ = scenario.world_points_from_prev(k, dp, (ox - px, oy - py), st, Tcw, rng)
        wp.append(P), dsc.append(d), octv.append(k["octave"])
    wp, dsc, octv = np.concatenate(wp), np.concatenate(dsc), np.concatenate(octv)
    mp = scenario.local_map(oracle, st, Tcw, wp, dsc, octv, sf, rng, obs_zero_frac=0.1, vary=True)
    frame = scenario.make_frame(oracle, ok, od, depth, st, sf)
    n0, k0 = oracle.search_by_projection(frame, mp, 3.0, 0.8, np.full(frame.n, -1, np.int32))
    assert n0 > 100
    table = {"world_pos": wp, "normal": mp["normal"], "min_dist": mp["min_dist"], "max_dist": mp["max_dist"],
             "desc": dsc, "skip": mp["bad"], "obs_pos": mp["obs_pos"]}
    log_sf = float(np.log(np.float32(sf[1])))
    n1, k1, in_view, nlo = oracle.search_local_points(frame, Tcw, float(st.fx), float(st.fy), float(st.cx), float(st.cy),
                                                       float(st.bf), table, log_sf)
    # scenario.local_map evaluates isInFrustum for bad points too; SearchByProjection skips them either way
    assert np.array_equal(in_view[mp["bad"] == 0], mp["in_view"][mp["bad"] == 0])
    assert n1 == n0 and np.array_equal(k1, k0)
