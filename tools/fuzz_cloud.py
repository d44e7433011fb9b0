"""Randomised parity sweep of the dense map (merge insert, general path, loop-closure rebuild, shutdown pass with the
outlier filter) against the oracle: random leaf sizes, camera paths that revisit or leave the mapped volume, frames of
two sizes.  Every intermediate map must be byte-identical.
    python tools/fuzz_cloud.py [seconds] [seed]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from orb_slam2_map_amd import lib as G
from orb_slam2_map_amd.synth import Stream
from oracle import oracle_py as O
import scenario

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0, n = time.time(), 0
paths = {1: 0, 2: 0, 3: 0}
while time.time() - t0 < budget:
    w, h = [(640, 480), (320, 240)][int(rng.integers(0, 2))]
    st = Stream(w, h, int(rng.integers(1, 1 << 30)))
    camv = (float(st.fx), float(st.fy), float(st.cx), float(st.cy))
    leaf = float(rng.choice([0.01, 0.02, 0.03, 0.05, 0.11]))
    cloud = G.PointCloudMapping(leaf)
    omap = np.zeros(0, O.POINT_DTYPE)
    kfs = []
    # scenarios that leave the merge path: "face" plants voxels whose centroid the next call indexes into the neighbouring
    # voxel (the resident map is then not strictly increasing: the key frame is redone, path 3); "far" moves the camera
    # tens of metres in all axes so that PCL's int32 overflow rule returns map ++ new unfiltered (path 3) and the key
    # frame after it starts from an unsorted map (general path, 2)
    kind = ["plain", "face", "far"][int(rng.choice([0, 0, 1, 2]))]
    faces = scenario.face_depth_values(leaf, 11, 120) if kind == "face" else []
    for i in range(int(rng.integers(2, 7))):
        _, rgb, depth = st.frame(int(rng.integers(0, 60)))
        step = float(rng.choice([0.0, 0.05, 0.4, 3.0]))  # same view, small motion, new area, far away
        if kind == "far" and i >= 1 and rng.random() < 0.5:
            step = 30.0 if leaf <= 0.02 else 120.0
        T = scenario.rigid(float(rng.normal(0, 0.02)), float(rng.normal(0, 0.02)), float(rng.normal(0, 0.02)),
                           tuple((rng.normal(0, 1, 3) * step).tolist()))
        if kind == "face":
            T = np.eye(4, dtype=np.float32)  # the planted depths must reach the map unrotated
            if faces and rng.random() < 0.7:
                k, z, cnt = faces[int(rng.integers(0, len(faces)))]
                depth = scenario.plant_face_voxel(depth, st, k, z, cnt, row=3 * int(rng.integers(10, h // 3 - 10)),
                                                  col0=3 * int(rng.integers(10, w // 3 - 20)))
        kfs.append((depth, rgb, T))
        cloud.insertKeyFrame(depth, rgb, *camv, T)
        paths[cloud.last_path()] += 1
        R, t = O.pose_inverse(T)
        new = O.transform_points(O.backproject(depth, rgb, *camv), R, t)
        omap, _ = O.voxel_filter(np.concatenate([omap, new]), leaf)
        if cloud.download().tobytes() != omap.tobytes():
            print("FAIL insert", dict(w=w, h=h, leaf=leaf, i=i, path=cloud.last_path()))
            sys.exit(1)
        n += 1
    # shutdown pass: per-key-frame clouds, concatenated, outlier filter
    cloud.clear()
    cat = []
    for depth, rgb, T in kfs:
        cloud.appendFiltered(depth, rgb, *camv, T)
        R, t = O.pose_inverse(T)
        cat.append(O.voxel_filter(O.transform_points(O.backproject(depth, rgb, *camv), R, t), leaf)[0])
    cat = np.concatenate(cat)
    if cloud.download().tobytes() != cat.tobytes():
        print("FAIL shutdown concat", dict(w=w, h=h, leaf=leaf))
        sys.exit(1)
    if len(cat) > 50 and len(cat) < 12000:  # the oracle's neighbour search is O(n^2)
        cloud.remove_outliers(50, 1.0)
        want, _ = O.statistical_outlier_removal(cat, 50, 1.0)
        if cloud.download().tobytes() != want.tobytes():
            print("FAIL outlier filter", dict(w=w, h=h, leaf=leaf, n=len(cat)))
            sys.exit(1)
    cloud.close()
print("fuzz ok: %d inserts (merge %d, general %d, redone %d) in %.0f s" % (n, paths[1], paths[2], paths[3], time.time() - t0))
