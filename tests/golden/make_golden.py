"""Generates the golden fixtures in this directory FROM THE ORACLE (oracle/liborb_oracle.so) on
seeded synthetic inputs.  They pin the oracle and the GPU path against regressions; they do NOT pin
the oracle to the reference (the reference cannot run here and ships no vectors: parity unpinned at
the OpenCV/PCL boundary, see oracle/orb_oracle.h).

    python tests/golden/make_golden.py
"""
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle_py as O  # noqa: E402
from orb_slam2_map_amd.synth import Stream  # noqa: E402

FIELDS = ("x", "y", "size", "angle", "response", "octave", "class_id")


def main():
    O.build()
    # (5) full extract, C2 and C3 frames
    for (w, h, nfeat) in ((640, 480, 1000), (1280, 960, 2000)):
        st = Stream(w, h, 1234)
        g, rgb, depth = st.frame(0)
        e = O.Extractor(nfeat)
        k, d = e.extract(g)
        arrs = {f: np.ascontiguousarray(k[f]) for f in FIELDS}
        arrs["desc"] = d
        arrs["image_crc"] = np.array([zlib.crc32(g.tobytes())], np.uint32)
        arrs["level_crc"] = np.array([zlib.crc32(e.pyramid_level(l).tobytes()) for l in range(8)], np.uint32)
        arrs["blur_crc"] = np.array([zlib.crc32(e.blurred_level(l).tobytes()) for l in range(8)], np.uint32)
        arrs["n_candidates"] = np.array([len(e.level_candidates(l)) for l in range(8)], np.int32)
        arrs["n_selected"] = np.array([len(e.level_selected(l)) for l in range(8)], np.int32)
        np.savez_compressed(os.path.join(HERE, "extract_%dx%d_seed1234_f0.npz" % (w, h)), **arrs)
    # (4) resize / blur on a small odd-sized image: full bytes
    rng = np.random.default_rng(42)
    img = rng.integers(0, 256, (61, 97), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "resize_blur_97x61.npz"), image=img,
                        resized=O.resize_linear(img, 81, 51), blurred=O.gauss7(img), bordered=O.border101(img, 19))
    # (3) FAST on a 64x64 texture
    tex = (rng.integers(0, 256, (64, 64)) // 64 * 64 + rng.integers(0, 20, (64, 64))).astype(np.uint8)
    np.savez_compressed(os.path.join(HERE, "fast_64x64.npz"), image=tex, th20=O.fast(tex, 20), th7=O.fast(tex, 7))
    # (6) BF matches of frames 0/1
    st = Stream(640, 480, 1234)
    e = O.Extractor(1000)
    k0, d0 = e.extract(st.frame(0)[0])
    k1, d1 = e.extract(st.frame(1)[0])
    n, mb = O.match_bf(d0, k0["angle"], d1, k1["angle"], nnratio=0.7)
    np.savez_compressed(os.path.join(HERE, "bf_640x480_f0_f1.npz"), nmatches=np.array([n], np.int32), match_b=mb)
    # (7) cloud: back-projection + voxel centroids of frame 0 with a non-trivial pose
    sys.path.insert(0, os.path.dirname(HERE))
    from scenario import rigid
    g, rgb, depth = st.frame(0)
    T = rigid()
    pts = O.backproject(depth, rgb, float(st.fx), float(st.fy), float(st.cx), float(st.cy))
    R, t = O.pose_inverse(T)
    wpts = O.transform_points(pts, R, t)
    vox, ov = O.voxel_filter(wpts, 0.01)
    vox5, _ = O.voxel_filter(wpts, 0.05)
    np.savez_compressed(os.path.join(HERE, "cloud_640x480_f0.npz"), Tcw=T, n_points=np.array([len(pts)], np.int32),
                        points_crc=np.array([zlib.crc32(wpts.tobytes())], np.uint32), first_points=wpts[:64],
                        n_vox_001=np.array([len(vox)], np.int32), vox_001_crc=np.array([zlib.crc32(vox.tobytes())], np.uint32),
                        vox_005=vox5)
    # (8) the outlier filter of the shutdown pass on the 5 cm cloud of (7): mean neighbour distances and the survivors
    kept, md = O.statistical_outlier_removal(vox5, 50, 1.0)
    np.savez_compressed(os.path.join(HERE, "sor_cloud_640x480_f0.npz"), mean_dist=md,
                        kept_crc=np.array([zlib.crc32(kept.tobytes())], np.uint32), n_kept=np.array([len(kept)], np.int32))
    # (2) Hamming known-answer test: 4096 descriptor pairs from a xorshift stream (tests/golden_scenarios.py), distances by
    #     the oracle, cross-checked here with Python integers
    import golden_scenarios as GS
    a, b = GS.hamming_kat_inputs()
    dist = np.array([O.descriptor_distance(x, y) for x, y in zip(a, b)], np.int32)
    assert np.array_equal(dist, GS.popcount_reference(a, b))
    np.savez_compressed(os.path.join(HERE, "hamming_kat_4096.npz"), dist=dist.astype(np.int16),
                        inputs_crc=np.array([zlib.crc32(a.tobytes() + b.tobytes())], np.uint32))
    # (6b) SearchByProjection assignments of C3's matcher problem (1280x960, 2000 features, ~10 k local map points)
    sc = GS.c3_projection_scenario(O)
    n, k2m = O.search_by_projection(sc["frame"], sc["mp"], sc["th"], sc["nnratio"], sc["k0"])
    np.savez_compressed(os.path.join(HERE, "projection_c3_seed5678.npz"), nmatches=np.array([n], np.int32), kp_to_mp=k2m,
                        n_map_points=np.array([len(sc["world_pos"])], np.int32),
                        inputs_crc=np.array([sc["inputs_crc"]], np.uint32))
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
