"""How far is (float)cos((double)x) -- the correctly rounded value, what ORBGPU_TRIG_ROUNDED_DOUBLE / rounds 1-3 computed
-- from this host's cosf / sinf, which is what ORBextractor.cc:112-113 really calls (`using namespace std`, :66-67)?
Counts, with the oracle in both modes, over the committed golden frames and a sweep of synthetic frames:
key points whose a or b differs, key points whose descriptor differs, descriptor bits that differ.
Used by tests/test_trig.py; run it directly for the DESIGN.md numbers:  python tools/trig_residual.py [keypoints]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def residual(frames, nfeatures=1000):
    """frames: iterable of gray images.  Returns the counts as a dict."""
    from oracle import oracle_py as O
    factor_pi = np.float32(np.float64(3.14159265358979323846) / np.float64(np.float32(180.0)))
    out = {"frames": 0, "keypoints": 0, "a_or_b_differs": 0, "a_differs": 0, "b_differs": 0, "descriptors_differ": 0,
           "descriptor_bits_differ": 0}
    ext = O.Extractor(nfeatures)
    try:
        for g in frames:
            O.set_trig_mode(O.TRIG_LIBM_FLOAT)
            k0, d0 = ext.extract(g)
            O.set_trig_mode(O.TRIG_ROUNDED_DOUBLE)
            k1, d1 = ext.extract(g)
            assert k0.tobytes() == k1.tobytes()  # the key points do not depend on the mode
            ang = (k0["angle"].astype(np.float32) * factor_pi).astype(np.float32)
            a1, b1 = np.cos(ang.astype(np.float64)).astype(np.float32), np.sin(ang.astype(np.float64)).astype(np.float32)
            O.set_trig_mode(O.TRIG_LIBM_FLOAT)
            a0, b0 = O.descriptor_trig_array(ang)
            da, db = a0 != a1, b0 != b1
            bits = np.unpackbits(d0 ^ d1, axis=1).sum(axis=1)
            out["frames"] += 1
            out["keypoints"] += len(k0)
            out["a_differs"] += int(da.sum())
            out["b_differs"] += int(db.sum())
            out["a_or_b_differs"] += int((da | db).sum())
            out["descriptors_differ"] += int((bits > 0).sum())
            out["descriptor_bits_differ"] += int(bits.sum())
            # a descriptor can only differ where a or b does
            assert not np.any((bits > 0) & ~(da | db))
    finally:
        O.set_trig_mode(O.TRIG_LIBM_FLOAT)
    return out


def golden_frames():
    from orb_slam2_map_amd.synth import Stream
    # the frames the committed goldens were made from (tests/golden/make_golden.py): seed 1234, frame 0, both sizes
    yield Stream(640, 480, 1234).frame(0)[0]
    yield Stream(1280, 960, 1234).frame(0)[0]


def sweep_frames(keypoints):
    from orb_slam2_map_amd.synth import Stream
    st = Stream(640, 480, 777)
    for t in range((keypoints + 999) // 1000):
        yield st.frame(t)[0]


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    g = residual(golden_frames(), 1000)
    s = residual(sweep_frames(n), 1000)
    print(json.dumps({"goldens": g, "sweep": s}))
