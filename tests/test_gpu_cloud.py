"""GPU parity: RGB-D back-projection / pose transform / voxel grid vs the oracle.
Tolerance from BASELINE.json north_star: XYZ within 1e-4 (f32); with the stable sort used on both
sides the results are in fact bit-identical, which the tests also record."""
import os
import zlib

import numpy as np
import pytest

import scenario

pytestmark = pytest.mark.gpu
TOL = 1e-4


def cam(st):
    return float(st.fx), float(st.fy), float(st.cx), float(st.cy)


def assert_points_close(a, b, what):
    assert len(a) == len(b), "%s: %d vs %d points" % (what, len(a), len(b))
    for f in "xyz":
        d = np.abs(a[f].astype(np.float64) - b[f].astype(np.float64))
        assert np.nanmax(d) <= TOL if len(d) else True, "%s: %s differs by %g" % (what, f, np.nanmax(d))
    assert np.array_equal(a["rgba"], b["rgba"]), "%s: colours differ" % what


@pytest.mark.parametrize("w,h", [(640, 480), (1280, 960), (641, 479)])
def test_backproject_exact(gpu, oracle, w, h):
    from orb_slam2_map_amd.synth import Stream
    st = Stream(w, h, 1234)
    g, rgb, depth = st.frame(4)
    depth = depth.copy()
    depth[0, 0], depth[3, 3], depth[6, 9], depth[9, 6] = 0.009, 10.0, 10.5, 0.01  # threshold edges (:121)
    gp = gpu.backproject(depth, rgb, *cam(st))
    op = oracle.backproject(depth, rgb, *cam(st))
    assert gp.tobytes() == op.tobytes(), "camera-frame points must be bit-exact and in scan order"
    T = scenario.rigid()
    gw = gpu.backproject(depth, rgb, *cam(st), Tcw=T)
    R, t = oracle.pose_inverse(T)
    ow = oracle.transform_points(op, R, t)
    assert_points_close(gw, ow, "world points")
    assert gw.tobytes() == ow.tobytes()


def test_backproject_all_invalid_and_nan(gpu, oracle, stream640):
    g, rgb, depth = stream640.frame(0)
    z = np.zeros_like(depth)
    assert len(gpu.backproject(z, rgb, *cam(stream640))) == 0
    d2 = depth.copy()
    d2[30, 30] = np.nan  # NaN passes both comparisons of :121 and becomes a non-finite point
    gp = gpu.backproject(d2, rgb, *cam(stream640))
    op = oracle.backproject(d2, rgb, *cam(stream640))
    assert gp.tobytes() == op.tobytes()
    vg, _ = gpu.voxel_filter(gp, 0.05)
    vo, _ = oracle.voxel_filter(op, 0.05)
    assert_points_close(vg, vo, "voxel with NaN")


@pytest.mark.parametrize("leaf", [0.01, 0.03, 0.05, 0.2])
def test_voxel_filter(gpu, oracle, stream640, leaf):
    """0.01 m is the benchmark leaf (BASELINE.json configs[3]); 0.03-0.05 are the reference's YAML values."""
    g, rgb, depth = stream640.frame(0)
    T = scenario.rigid()
    pts = gpu.backproject(depth, rgb, *cam(stream640), Tcw=T)
    vg, ovg = gpu.voxel_filter(pts, leaf)
    vo, ovo = oracle.voxel_filter(pts, leaf)
    assert ovg == ovo == False
    assert_points_close(vg, vo, "leaf %g" % leaf)
    assert vg.tobytes() == vo.tobytes(), "stable sort on both sides: identical float sums"
    assert len(vg) < len(pts) or leaf == 0.01


def test_voxel_filter_overflow_returns_input(gpu, oracle):
    """PCL: if dx*dy*dz overflows int32 the input is returned unfiltered."""
    rng = np.random.default_rng(0)
    pts = np.zeros(5000, gpu.POINT_DTYPE)
    pts["x"], pts["y"], pts["z"] = rng.random(5000) * 40, rng.random(5000) * 40, rng.random(5000) * 40
    pts["rgba"] = rng.integers(0, 1 << 24, 5000)
    vg, ovg = gpu.voxel_filter(pts, 0.01)
    vo, ovo = oracle.voxel_filter(pts, 0.01)
    assert ovg and ovo and vg.tobytes() == pts.tobytes() == vo.tobytes()


def test_voxel_filter_random_clusters_and_negative_coords(gpu, oracle):
    rng = np.random.default_rng(7)
    centres = rng.normal(0, 1.5, (300, 3))
    p = (centres[rng.integers(0, 300, 200000)] + rng.normal(0, 0.02, (200000, 3))).astype(np.float32)
    pts = np.zeros(len(p), gpu.POINT_DTYPE)
    pts["x"], pts["y"], pts["z"] = p[:, 0], p[:, 1], p[:, 2]
    pts["rgba"] = rng.integers(0, 1 << 24, len(p))
    vg, _ = gpu.voxel_filter(pts, 0.01)
    vo, _ = oracle.voxel_filter(pts, 0.01)
    assert_points_close(vg, vo, "clusters")
    # size-independent properties: output is sorted by voxel index, one point per voxel, and the
    # count-weighted centroid of the output equals the centroid of the input
    key = np.floor(vg["x"] * np.float32(100)).astype(np.int64), np.floor(vg["y"] * np.float32(100)).astype(np.int64), \
        np.floor(vg["z"] * np.float32(100)).astype(np.int64)
    lin = (key[2] - key[2].min()) * (1 << 40) + (key[1] - key[1].min()) * (1 << 20) + (key[0] - key[0].min())
    assert len(np.unique(lin)) >= 0.999 * len(lin)  # centroids sit inside their voxel up to rounding
    idem, _ = gpu.voxel_filter(vg, 0.01)
    assert len(idem) <= len(vg)


def test_cloud_handle_insert_and_rebuild(gpu, oracle, stream640):
    """PointCloudMapping::viewer's no-loop branch (insert x3) and loop-closure branch (rebuild)."""
    poses = [scenario.rigid(0.01 * i, -0.02 * i, 0.005 * i, (0.05 * i, 0.0, 0.02 * i)) for i in range(3)]
    frames = [stream640.frame(10 * i) for i in range(3)]
    for leaf in (0.05, 0.01):
        cloud = gpu.PointCloudMapping(leaf)
        omap = np.zeros(0, oracle.POINT_DTYPE)
        for (g, rgb, depth), T in zip(frames, poses):
            cloud.insertKeyFrame(depth, rgb, *cam(stream640), T)
            R, t = oracle.pose_inverse(T)
            new = oracle.transform_points(oracle.backproject(depth, rgb, *cam(stream640)), R, t)
            omap, _ = oracle.voxel_filter(np.concatenate([omap, new]), leaf)
            assert cloud.size() == len(omap)
            assert_points_close(cloud.download(), omap, "insert leaf %g" % leaf)
        # loop closure: new poses for all key frames, single filter over the concatenation
        poses2 = [scenario.rigid(0.012 * i, -0.018 * i, 0.004 * i, (0.051 * i, 0.001, 0.019 * i)) for i in range(3)]
        cloud.rebuild([f[2] for f in frames], [f[1] for f in frames], *cam(stream640), poses2)
        allp = []
        for (g, rgb, depth), T in zip(frames, poses2):
            R, t = oracle.pose_inverse(T)
            allp.append(oracle.transform_points(oracle.backproject(depth, rgb, *cam(stream640)), R, t))
        oreb, _ = oracle.voxel_filter(np.concatenate(allp), leaf)
        assert_points_close(cloud.download(), oreb, "rebuild leaf %g" % leaf)
        cloud.close()


def test_cloud_golden(gpu, stream640):
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "cloud_640x480_f0.npz"))
    _, rgb, depth = stream640.frame(0)
    pts = gpu.backproject(depth, rgb, *cam(stream640), Tcw=g["Tcw"])
    assert len(pts) == int(g["n_points"][0]) and zlib.crc32(pts.tobytes()) == int(g["points_crc"][0])
    v5, _ = gpu.voxel_filter(pts, 0.05)
    gv = g["vox_005"]
    assert len(v5) == len(gv)
    for f in "xyz":
        assert np.max(np.abs(v5[f] - gv[f])) <= TOL
    v1, _ = gpu.voxel_filter(pts, 0.01)
    assert len(v1) == int(g["n_vox_001"][0])


def test_voxel_filter_properties_at_full_size(gpu):
    """Size-independent properties at a map size the CPU oracle would take minutes for (4 M points): the filter is
    idempotent bit for bit (one point per voxel is its own centroid), every output lies in a distinct voxel, the
    voxel indices ascend, and the point count is preserved through the per-voxel averages' weights."""
    rng = np.random.default_rng(11)
    n, leaf = 4_000_000, 0.02
    pts = np.zeros(n, dtype=gpu.POINT_DTYPE)
    centers = rng.uniform(-3, 3, (2000, 3))
    which = rng.integers(0, 2000, n)
    xyz = (centers[which] + rng.normal(0, 0.05, (n, 3))).astype(np.float32)
    pts["x"], pts["y"], pts["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    pts["rgba"] = rng.integers(0, 1 << 24, n, dtype=np.uint32)
    once, ov1 = gpu.voxel_filter(pts, leaf)
    twice, ov2 = gpu.voxel_filter(once, leaf)
    assert not ov1 and not ov2
    assert 1000 < len(once) < n
    assert np.array_equal(once.view(np.uint32), twice.view(np.uint32))
    inv = np.float32(1.0) / np.float32(leaf)
    ijk = np.stack([np.floor(once[c] * inv) for c in "xyz"], 1).astype(np.int64)
    ijk -= np.floor(np.array([pts[c].min() for c in "xyz"], np.float32) * inv).astype(np.int64)
    div = ijk.max(0) + 1  # not PCL's div_b (that uses the input's max), but monotone in the same lexicographic order
    dx = int(np.floor(pts["x"].max() * inv) - np.floor(pts["x"].min() * inv) + 1)
    dy = int(np.floor(pts["y"].max() * inv) - np.floor(pts["y"].min() * inv) + 1)
    idx = ijk[:, 0] + ijk[:, 1] * dx + ijk[:, 2] * dx * dy
    assert np.all(np.diff(idx) > 0), "outputs must lie in distinct voxels in ascending index order"
    assert div[0] <= dx and div[1] <= dy


def _oracle_step(oracle, omap, depth, rgb, camv, T, leaf):
    R, t = oracle.pose_inverse(T)
    new = oracle.transform_points(oracle.backproject(depth, rgb, *camv), R, t)
    return oracle.voxel_filter(np.concatenate([omap, new]), leaf)


@pytest.mark.parametrize("leaf", [0.01, 0.04])
def test_cloud_merge_path_sequence(gpu, oracle, stream640, leaf):
    """Eight key frames with a moving camera: from the second on the resident map is merged with the sorted new
    points (path 1) instead of re-sorted; every intermediate map must equal PCL's filter over (map ++ new)."""
    camv = cam(stream640)
    cloud = gpu.PointCloudMapping(leaf)
    omap = np.zeros(0, oracle.POINT_DTYPE)
    for i in range(8):
        _, rgb, depth = stream640.frame(9 * i)
        if i == 3:
            depth = depth.copy()
            depth[60, 90] = np.nan  # a non-finite new point: dropped by the filter, never part of the map
        T = scenario.rigid(0.006 * i, -0.009 * i, 0.003 * i, (0.06 * i, 0.01 * i, -0.02 * i))
        cloud.insertKeyFrame(depth, rgb, *camv, T)
        omap, ov = _oracle_step(oracle, omap, depth, rgb, camv, T, leaf)
        assert not ov and cloud.last_path() == 1, "key frame %d took path %d" % (i, cloud.last_path())
        assert cloud.size() == len(omap)
        got = cloud.download()
        assert_points_close(got, omap, "key frame %d leaf %g" % (i, leaf))
        assert got.tobytes() == omap.tobytes()
    cloud.close()


@pytest.mark.parametrize("cap", ["0", "300", "5000"])
def test_cloud_merge_path_with_long_resident_ranges(gpu, oracle, stream640, monkeypatch, cap):
    """k_merge_new ranks a tile of new keys against the resident indices of the range they can fall into: staged in
    LDS when the range has at most `cap` points (24576 in production), sampled + searched in global memory otherwise.
    ORBGPU_DEBUG_MERGE_CAP (read when the handle is created) makes ordinary key frames exercise the long-range path (0:
    every tile) and the mix of both (300, 5000); a dense map seen again through a sparse frame (3 of 4 depth samples
    invalid) stretches the ranges on top of that."""
    monkeypatch.setenv("ORBGPU_DEBUG_MERGE_CAP", cap)
    camv = cam(stream640)
    cloud = gpu.PointCloudMapping(0.02)
    omap = np.zeros(0, oracle.POINT_DTYPE)
    rng = np.random.default_rng(5)
    for i in range(6):
        _, rgb, depth = stream640.frame(7 * i)
        if i >= 4:
            depth = depth.copy()
            depth[rng.random(depth.shape) < 0.75] = 0.0
        T = scenario.rigid(0.004 * i, -0.006 * i, 0.002 * i, (0.05 * i, 0.01 * i, -0.02 * i))
        cloud.insertKeyFrame(depth, rgb, *camv, T)
        omap, ov = _oracle_step(oracle, omap, depth, rgb, camv, T, 0.02)
        assert not ov and cloud.last_path() == 1, "key frame %d took path %d" % (i, cloud.last_path())
        assert cloud.download().tobytes() == omap.tobytes(), "key frame %d cap %s" % (i, cap)
    cloud.close()


def test_cloud_same_view_twice_merges_into_existing_voxels(gpu, oracle, stream640):
    """The same key frame inserted twice: every new point falls into an occupied voxel (no new voxels, map size
    unchanged), then a third insert from elsewhere opens new ones before / between / after the resident ones."""
    camv = cam(stream640)
    _, rgb, depth = stream640.frame(3)
    T = scenario.rigid()
    cloud = gpu.PointCloudMapping(0.02)
    omap = np.zeros(0, oracle.POINT_DTYPE)
    sizes = []
    for Tk in (T, T, scenario.rigid(0.3, -0.2, 0.1, (-0.8, 0.4, 0.6))):
        cloud.insertKeyFrame(depth, rgb, *camv, Tk)
        omap, _ = _oracle_step(oracle, omap, depth, rgb, camv, Tk, 0.02)
        assert cloud.last_path() == 1
        assert cloud.download().tobytes() == omap.tobytes()
        sizes.append(cloud.size())
    assert sizes[0] == sizes[1] < sizes[2]
    cloud.close()


def test_cloud_overflow_then_recovery(gpu, oracle, stream640):
    """A key frame far away makes dx*dy*dz overflow int32: PCL returns map ++ new unfiltered (path 3: the merge
    notices and the general path redoes it).  The map is then unsorted, so the next key frame goes through the
    general path (2); once filtered again the merge path (1) resumes."""
    camv = cam(stream640)
    leaf = 0.01
    cloud = gpu.PointCloudMapping(leaf)
    omap = np.zeros(0, oracle.POINT_DTYPE)
    far = scenario.rigid(0.0, 0.0, 0.0, (400.0, -300.0, 250.0))
    poses = [scenario.rigid(), far, scenario.rigid(0.02, 0.01, 0.0, (0.1, 0.0, 0.0))]
    paths = []
    for i, T in enumerate(poses):
        _, rgb, depth = stream640.frame(5 * i)
        cloud.insertKeyFrame(depth, rgb, *camv, T)
        omap, ov = _oracle_step(oracle, omap, depth, rgb, camv, T, leaf)
        paths.append((cloud.last_path(), bool(cloud.last_overflow()), bool(ov)))
        assert cloud.size() == len(omap)
        assert cloud.download().tobytes() == omap.tobytes(), "key frame %d" % i
    assert paths[0] == (1, False, False) and paths[1] == (3, True, True) and paths[2][0] == 2
    # a coarser grid does not overflow: shrink the problem by rebuilding near the origin, then merge again
    cloud.rebuild([stream640.frame(0)[2]], [stream640.frame(0)[1]], *camv, [poses[0]])
    _, rgb, depth = stream640.frame(20)
    cloud.insertKeyFrame(depth, rgb, *camv, poses[2])
    assert cloud.last_path() == 1
    R, t = oracle.pose_inverse(poses[0])
    base, _ = oracle.voxel_filter(oracle.transform_points(oracle.backproject(stream640.frame(0)[2], stream640.frame(0)[1], *camv), R, t), leaf)
    want, _ = _oracle_step(oracle, base, depth, rgb, camv, poses[2], leaf)
    assert cloud.download().tobytes() == want.tobytes()
    cloud.close()


@pytest.mark.parametrize("which", [0, 1, 4])
def test_cloud_centroid_on_a_voxel_face_redoes_the_key_frame(gpu, oracle, stream640, which):
    """The merge path's OTHER precondition failure (no int32 overflow involved): a voxel whose float centroid the next
    filter call puts into the neighbouring voxel.  Key frame 1 plants `count` samples at the largest depth of voxel k-1
    whose mean already indexes voxel k, plus one sample in voxel k of the same column; the map that comes back holds two
    points with the same index under any later grid, so key frame 2 finds the resident map "not strictly increasing",
    redoes the insert through the general path (3) and must still equal PCL's result.  Afterwards the merge path resumes."""
    camv = cam(stream640)
    leaf = 0.05
    k, z, cnt = [v for v in scenario.face_depth_values(leaf) if v[2] == 3][which]  # (three samples 9 px apart share an x voxel)
    _, rgb, depth = stream640.frame(3)
    d1 = scenario.plant_face_voxel(depth, stream640, k, z, cnt)
    T = np.eye(4, dtype=np.float32)
    cloud = gpu.PointCloudMapping(leaf)
    omap = np.zeros(0, oracle.POINT_DTYPE)
    paths = []
    for dep in (d1, depth, stream640.frame(9)[2]):
        cloud.insertKeyFrame(dep, rgb, *camv, T)
        omap, ov = _oracle_step(oracle, omap, dep, rgb, camv, T, leaf)
        assert not ov
        paths.append(cloud.last_path())
        assert cloud.download().tobytes() == omap.tobytes(), "key frame %d (path %d)" % (len(paths), paths[-1])
    assert paths == [1, 3, 1], paths
    cloud.close()


def test_cloud_insert_device_resident(gpu, oracle, stream640):
    """orbgpu_cloud_insert_device: depth / rgb already in HBM (strided views of larger device buffers)."""
    import torch
    camv = cam(stream640)
    cloud = gpu.PointCloudMapping(0.01)
    omap = np.zeros(0, oracle.POINT_DTYPE)
    for i in range(3):
        _, rgb, depth = stream640.frame(4 * i)
        T = scenario.rigid(0.004 * i, 0.002 * i, 0.0, (0.05 * i, 0.0, 0.01 * i))
        h, w = depth.shape
        dd = torch.zeros((h, w + 16), dtype=torch.float32, device="cuda")
        dr = torch.zeros((h, (w + 5) * 3), dtype=torch.uint8, device="cuda")
        dd[:, :w] = torch.from_numpy(depth).cuda()
        dr[:, :w * 3] = torch.from_numpy(rgb.reshape(h, w * 3)).cuda()
        torch.cuda.synchronize()
        cloud.insertKeyFrameDevice(dd.data_ptr(), w + 16, dr.data_ptr(), (w + 5) * 3, w, h, *camv, T)
        omap, _ = _oracle_step(oracle, omap, depth, rgb, camv, T, 0.01)
        assert cloud.download().tobytes() == omap.tobytes()
    cloud.close()


def test_cloud_large_frame_and_tiny_frame(gpu, oracle, stream1280):
    """1280x960 key frames (136 k new points, 134 sort tiles) and a 7x5 image (one partial tile)."""
    camv = cam(stream1280)
    cloud = gpu.PointCloudMapping(0.02)
    omap = np.zeros(0, oracle.POINT_DTYPE)
    for i in range(2):
        _, rgb, depth = stream1280.frame(6 * i)
        T = scenario.rigid(0.01 * i, 0.0, 0.0, (0.1 * i, 0.0, 0.0))
        cloud.insertKeyFrame(depth, rgb, *camv, T)
        omap, _ = _oracle_step(oracle, omap, depth, rgb, camv, T, 0.02)
        assert cloud.last_path() == 1 and cloud.download().tobytes() == omap.tobytes()
    cloud.close()
    rng = np.random.default_rng(3)
    depth = (1.0 + rng.random((5, 7))).astype(np.float32)
    rgb = rng.integers(0, 255, (5, 7, 3), dtype=np.uint8)
    cloud = gpu.PointCloudMapping(0.01)
    T = scenario.rigid()
    cloud.insertKeyFrame(depth, rgb, 5.0, 5.0, 3.0, 2.0, T)
    cloud.insertKeyFrame(np.zeros_like(depth), rgb, 5.0, 5.0, 3.0, 2.0, T)  # no valid depth: map unchanged
    want, _ = _oracle_step(oracle, np.zeros(0, oracle.POINT_DTYPE), depth, rgb, (5.0, 5.0, 3.0, 2.0), T, 0.01)
    assert cloud.last_path() == 1 and cloud.download().tobytes() == want.tobytes()
    cloud.close()


def test_cloud_shutdown_pass(gpu, oracle, stream640):
    """PointCloudMapping::viewer after the loop (PointCloudMap.cc:270-282): clear, then per key frame the voxel
    filter of that key frame's cloud alone, concatenated; a later insert sees an unsorted map (general path)."""
    camv = cam(stream640)
    cloud = gpu.PointCloudMapping(0.03)
    poses = [scenario.rigid(0.01 * i, 0.0, 0.0, (0.2 * i, 0.0, 0.0)) for i in range(3)]
    frames = [stream640.frame(8 * i) for i in range(3)]
    for (g, rgb, depth), T in zip(frames, poses):
        cloud.insertKeyFrame(depth, rgb, *camv, T)
    cloud.clear()
    assert cloud.size() == 0
    want = []
    for (g, rgb, depth), T in zip(frames, poses):
        cloud.appendFiltered(depth, rgb, *camv, T)
        R, t = oracle.pose_inverse(T)
        v, _ = oracle.voxel_filter(oracle.transform_points(oracle.backproject(depth, rgb, *camv), R, t), 0.03)
        want.append(v)
        assert cloud.download().tobytes() == np.concatenate(want).tobytes()
    g, rgb, depth = stream640.frame(30)
    cloud.insertKeyFrame(depth, rgb, *camv, poses[1])
    omap, _ = _oracle_step(oracle, np.concatenate(want), depth, rgb, camv, poses[1], 0.03)
    assert cloud.last_path() == 2 and cloud.download().tobytes() == omap.tobytes()
    cloud.close()
