"""pcl::StatisticalOutlierRemoval as the shutdown pass uses it (PointCloudMap.cc:46-47, 283-285).

CPU: the oracle restatement against an independent exact neighbour search (scipy cKDTree, double arithmetic) and
against hand-checkable cases.  GPU: the grid search through the C ABI against the oracle, bit for bit.
Third-party boundary (PCL 1.7 / FLANN) unpinned: the reference holds no fixture for this filter."""
import numpy as np
import pytest

from oracle import oracle_py as O


def cloud(xyz, rng=None):
    p = np.zeros(len(xyz), O.POINT_DTYPE)
    p["x"], p["y"], p["z"] = np.asarray(xyz, np.float32).T
    p["rgba"] = np.arange(len(xyz), dtype=np.uint32) if rng is None else rng.integers(0, 2 ** 32, len(xyz), dtype=np.uint32)
    return p


def surface(n, rng, noise=0.002):
    """a wavy sheet like a voxel-filtered depth map, plus a few floating outliers"""
    u, v = rng.uniform(-1, 1, n), rng.uniform(-0.7, 0.7, n)
    z = 2.0 + 0.1 * np.sin(3 * u) * np.cos(2 * v) + rng.normal(0, noise, n)
    xyz = np.stack([u, v, z], 1)
    k = max(n // 200, 3)
    xyz[:k] += rng.uniform(0.2, 1.0, (k, 3))
    return xyz.astype(np.float32)


def test_oracle_mean_distance_matches_kdtree():
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(1)
    xyz = surface(3000, rng)
    kept, md = O.statistical_outlier_removal(cloud(xyz), 50, 1.0)
    d, _ = cKDTree(xyz.astype(np.float64)).query(xyz.astype(np.float64), 51)
    ref = d[:, 1:].mean(1)
    assert np.abs(ref - md).max() < 2e-6  # float squared distances vs double
    thr = ref.mean() + ref.std(ddof=1)
    clear = np.abs(ref - thr) > 1e-5
    assert np.array_equal((md <= thr)[clear], (ref <= thr)[clear])
    assert 0 < len(xyz) - len(kept) < len(xyz) // 4
    # survivors keep their order and their colour word
    assert np.all(np.diff(kept["rgba"].astype(np.int64)) > 0)


def test_oracle_small_cases():
    # 60 points on a line 1 cm apart and one far away: only the far one goes
    xyz = np.zeros((61, 3), np.float32)
    xyz[:60, 0] = np.arange(60) * 0.01
    xyz[60] = (0.3, 5.0, 0.0)
    kept, md = O.statistical_outlier_removal(cloud(xyz), 50, 1.0)
    assert len(kept) == 60 and md[60] > 4.9
    # the middle point's 50 neighbours are 25 on each side: mean distance = 0.01 * (1 + ... + 25) * 2 / 50
    assert abs(md[30] - 0.01 * 13) < 1e-6
    # non-finite points: distance 0, not a neighbour of anyone, kept
    xyz2 = np.vstack([xyz, [[np.nan, 0, 0]]]).astype(np.float32)
    kept2, md2 = O.statistical_outlier_removal(cloud(xyz2), 50, 1.0)
    assert md2[61] == 0 and np.array_equal(md2[:61], md) and len(kept2) == 61
    with pytest.raises(ValueError):
        O.statistical_outlier_removal(cloud(xyz[:50]), 50, 1.0)


@pytest.mark.gpu
@pytest.mark.parametrize("n,kind", [(4000, "surface"), (3000, "ball"), (2500, "duplicates"), (600, "line"), (52, "tiny")])
def test_gpu_matches_oracle(gpu, n, kind):
    rng = np.random.default_rng(n)
    if kind == "surface":
        xyz = surface(n, rng)
    elif kind == "ball":
        xyz = rng.normal(0, 1, (n, 3)).astype(np.float32)  # volumetric, density falls off: several shells
    elif kind == "duplicates":
        xyz = surface(n // 2, rng)
        xyz = np.vstack([xyz, xyz[rng.integers(0, len(xyz), n - len(xyz))]])  # overlapping key-frame clouds
    elif kind == "line":
        xyz = np.zeros((n, 3), np.float32)
        xyz[:, 0] = rng.uniform(-5, 5, n)
    else:
        xyz = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    pts = cloud(xyz, rng)
    if kind == "surface":
        pts["x"][[7, 1234]] = np.nan
        pts["z"][99] = np.inf
    for mean_k, mul in ((50, 1.0), (8, 0.5)):
        ok, omd = O.statistical_outlier_removal(pts, mean_k, mul)
        gk, gmd = gpu.statistical_outlier_removal(pts, mean_k, mul)
        assert np.array_equal(gmd.view(np.uint32), omd.view(np.uint32)), "%s k=%d: mean distances differ at %d points" % (
            kind, mean_k, int((gmd.view(np.uint32) != omd.view(np.uint32)).sum()))
        assert gk.tobytes() == ok.tobytes(), "%s k=%d: kept sets differ (%d vs %d)" % (kind, mean_k, len(gk), len(ok))


@pytest.mark.gpu
def test_gpu_large_cloud_properties(gpu):
    """400 k points (the oracle is O(n^2)): against cKDTree in double, and idempotent bookkeeping."""
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(9)
    xyz = np.vstack([surface(200000, rng), surface(200000, rng) + np.float32([0.003, 0.002, 0.0])])
    pts = cloud(xyz)  # colour word = index: identifies the survivors
    gk, gmd = gpu.statistical_outlier_removal(pts, 50, 1.0)
    d, _ = cKDTree(xyz.astype(np.float64)).query(xyz.astype(np.float64), 51, workers=8)
    ref = d[:, 1:].mean(1)
    assert np.abs(ref - gmd).max() < 1e-5 * max(1.0, ref.max())
    thr = gmd.astype(np.float64).mean() + gmd.astype(np.float64).std(ddof=1)
    clear = np.abs(gmd - thr) > 1e-4 * thr
    kept_mask = np.isin(pts["rgba"], gk["rgba"])
    assert np.array_equal(kept_mask[clear], (gmd <= thr)[clear])
    assert np.array_equal(gk, pts[kept_mask])  # order kept


@pytest.mark.gpu
def test_gpu_shutdown_sequence(gpu, stream640, tmp_path):
    """clear -> per-key-frame filtered clouds appended -> outlier filter -> PCD (PointCloudMap.cc:270-287); a later
    insert still works on the filtered map."""
    import scenario
    st = stream640
    camv = (float(st.fx), float(st.fy), float(st.cx), float(st.cy))
    m = gpu.PointCloudMapping(0.02)
    m.clear()
    poses = [scenario.rigid(0.01 * i, 0.0, 0.0, (0.2 * i, 0.0, 0.0)) for i in range(3)]
    for i, T in enumerate(poses):
        _, rgb, depth = st.frame(8 * i)
        m.appendFiltered(depth, rgb, *camv, T)
    before = m.download()
    removed = m.remove_outliers(50, 1.0)
    after = m.download()
    assert removed == len(before) - len(after) and 0 < removed < len(before) // 3 and m.size() == len(after)
    want, _ = gpu.statistical_outlier_removal(before, 50, 1.0)
    assert after.tobytes() == want.tobytes()
    path = str(tmp_path / "optimized_pointcloud.pcd")
    m.save_pcd(path)
    raw = open(path, "rb").read()
    assert raw.endswith(after.tobytes()) and (b"POINTS %d\n" % len(after)) in raw
    _, rgb, depth = st.frame(30)
    m.insertKeyFrame(depth, rgb, *camv, poses[1])
    assert m.last_path() == 2 and m.size() > 0
    m.close()


@pytest.mark.gpu
def test_gpu_golden(gpu):
    import os
    import zlib
    G = os.path.join(os.path.dirname(__file__), "golden")
    g = np.load(os.path.join(G, "cloud_640x480_f0.npz"))
    s = np.load(os.path.join(G, "sor_cloud_640x480_f0.npz"))
    kept, md = gpu.statistical_outlier_removal(g["vox_005"], 50, 1.0)
    assert np.array_equal(md.view(np.uint32), s["mean_dist"].view(np.uint32))
    assert len(kept) == int(s["n_kept"][0]) and zlib.crc32(kept.tobytes()) == int(s["kept_crc"][0])
