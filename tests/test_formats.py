"""On-disk formats: the Map::Save records (reference src/Map.cc:123-183) and the binary PCD
of PointCloudMapping's shutdown (src/PointCloudMap.cc:287), checked against an independent `struct` layout.  Host
serialisers: no device needed except for saving a handle's map."""
import struct

import numpy as np
import pytest

import scenario


@pytest.fixture(scope="module")
def glib():
    from orb_slam2_map_amd import lib
    lib.lib()
    return lib


def test_mappoint_record(glib):
    b = glib.mappoint_record(123456789012, [1.5, -2.25, 3.125])
    assert b == struct.pack("<Qfff", 123456789012, 1.5, -2.25, 3.125)


@pytest.mark.parametrize("rx,ry,rz", [(0.01, -0.02, 0.015), (3.0, 0.2, -0.1), (0.1, 3.1, 0.0), (-0.3, 0.2, 3.0)])
def test_keyframe_record_layout(glib, rx, ry, rz):
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(4)
    n = 37
    keys = np.zeros(n, glib.KEYPOINT_DTYPE)
    for f in ("x", "y", "size", "angle", "response"):
        keys[f] = rng.random(n).astype(np.float32) * 100
    keys["octave"] = rng.integers(0, 8, n)
    keys["class_id"] = -1
    desc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    idx = rng.integers(0, 5000, n).astype(np.uint64)
    idx[::5] = np.uint64(2 ** 64 - 1)  # ULONG_MAX: no map point
    T = scenario.rigid(rx, ry, rz, (0.3, -1.2, 2.5))
    b = glib.keyframe_record(42, 1305031102.175304, T, keys, desc, idx)
    assert len(b) == 8 + 8 + 12 + 16 + 4 + 64 * n
    kid, ts, px, py, pz, qx, qy, qz, qw, cnt = struct.unpack_from("<Qdfffffffi", b, 0)
    assert (kid, ts, cnt) == (42, 1305031102.175304, n)
    assert (px, py, pz) == tuple(float(v) for v in T[:3, 3])
    want = Rotation.from_matrix(T[:3, :3].astype(np.float64)).as_quat()  # x y z w, unit norm
    got = np.array([qx, qy, qz, qw])
    if np.dot(got, want) < 0:
        want = -want  # q and -q are the same rotation; Eigen keeps the branch's sign
    assert np.allclose(got, want, atol=2e-6), (got, want)
    off = 48
    for i in range(n):
        x, y, size, angle, resp, octave = struct.unpack_from("<fffffi", b, off)
        assert (x, y, size, angle, resp, octave) == tuple(keys[i][f].item() for f in ("x", "y", "size", "angle", "response", "octave"))
        assert b[off + 24:off + 56] == desc[i].tobytes()
        assert struct.unpack_from("<Q", b, off + 56)[0] == int(idx[i])
        off += 64
    assert off == len(b)
    assert glib.keyframe_record(7, 0.5, T, keys[:0], desc[:0], idx[:0])[-4:] == struct.pack("<i", 0)


def test_pcd_header_and_file(glib, tmp_path):
    h = glib.pcd_binary_header(12345)
    lines = h.decode().split("\n")
    assert lines[0] == "# .PCD v0.7 - Point Cloud Data file format" and lines[-1] == "" and lines[-2] == "DATA binary"
    assert "FIELDS x y z rgba" in lines and "SIZE 4 4 4 4" in lines and "TYPE F F F U" in lines
    assert "WIDTH 12345" in lines and "HEIGHT 1" in lines and "POINTS 12345" in lines
    pts = np.zeros(5, glib.POINT_DTYPE)
    pts["x"], pts["rgba"] = np.arange(5), np.arange(5) * 65793
    import ctypes as C
    L = glib.lib()
    L.orbgpu_write_pcd_binary.argtypes = [C.c_char_p, C.c_void_p, C.c_int64]
    p = str(tmp_path / "m.pcd")
    assert L.orbgpu_write_pcd_binary(p.encode(), pts.ctypes.data_as(C.c_void_p), 5) == 0
    raw = open(p, "rb").read()
    assert raw == glib.pcd_binary_header(5) + pts.tobytes()


@pytest.mark.gpu
def test_cloud_save_pcd(glib, tmp_path, stream640):
    _, rgb, depth = stream640.frame(0)
    cloud = glib.PointCloudMapping(0.05)
    cloud.insertKeyFrame(depth, rgb, float(stream640.fx), float(stream640.fy), float(stream640.cx), float(stream640.cy),
                         scenario.rigid())
    p = str(tmp_path / "optimized_pointcloud.pcd")
    cloud.save_pcd(p)
    raw = open(p, "rb").read()
    m = cloud.download()
    assert raw == glib.pcd_binary_header(len(m)) + m.tobytes()
    cloud.close()
